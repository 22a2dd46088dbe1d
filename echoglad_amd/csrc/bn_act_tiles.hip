// Train-mode BatchNorm affine + Dropout + ReLU + residual (reference src/core/models.py:333-335, :434-435) walked in the layer
// kernels' TILE order, so that the pass which writes a layer's output also leaves the CHILD SUMS of that output behind
// for the next layer's train forward (the side buffer eg_gcn_layer_fwd_chain's inference layers hand each other):
//     out[n]     = relu?(dropout(z[n] * scale + shift)) + residual[n]
//     kidsum[p]  = sum over the 4 children c of aux node p of (deg_c + 1)^-1/2 out[c]
// A streaming kernel: one wave per PAIR of 8-node segments (patch rows 2p, 2p+1 of an 8x8 patch: 16 whole 512-B rows of z
// and of the residual in flight per wave), the four children of a parent are columns 2j, 2j+1 of the pair's two rows, i.e.
// the two halves of the wave hold them in the same registers -- one cross-half add per parent row.  No MFMA beside it,
// so vector work is free here; bound by the 3 x [rows,128] it moves (+ the child-sum rows, 30 % of a pass at 224/7).
#include <stdlib.h>

#include "seg_wide.h"
#include "train_common.h"

namespace eg {

struct ActTileArgs {
    int n_per_frame, tiles_per_frame, batch, kid_rows;
    int relu;
    float p, inv_keep;
    unsigned long long seed;
    const unsigned long long* epoch;
};
__device__ inline ActTileArgs resolved(const ActTileArgs& a) {
    ActTileArgs r = a;
    r.seed = a.seed + epoch_now(a.epoch);
    return r;
}

__device__ inline f32x4 act4(const f32x4& z, const f32x4& sc, const f32x4& sh, const ActTileArgs& a, unsigned long long idx) {
    f32x4 v = z * sc + sh;                          // (the same expression as k_bn_act_fwd / the backward kernels: DESIGN 5.13)
    if (a.p > 0.f) v *= keep_scale4(a.seed, idx, a.p, a.inv_keep);
    if (a.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    return v;
}

#ifndef EG_ACT_AUX
#define EG_ACT_AUX 0            // aux bits of the row loads (2 = nt)
#endif
__device__ inline f32x4 ldp_s(const RowSrc& s, int row0, int k) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(s.r, s.pair + k * (2 * C * 4), row0 * (C * 4), EG_ACT_AUX));
}
#define ldp ldp_s

template <bool KOUT>
__device__ inline void act_pair(const float* __restrict__ z, const float* __restrict__ scale,
                                const float* __restrict__ shift, const float* __restrict__ residual,
                                float* __restrict__ out, float* __restrict__ kout, const float* __restrict__ patsq,
                                const ActTileArgs& a, int frame, const SegDesc& sd0, const SegDesc& sd1, int lane) {
    if (sd0.cnt == 0 && sd1.cnt == 0) return;
    const size_t fbase = (size_t)frame * a.n_per_frame * C;
    const int fbytes = a.n_per_frame * (C * 4);
    const PairLane pl{lane >> 5, lane & 31};
    if (sd0.pad0) {
        // ---- both segments whole runs of rows: 16 rows of every operand per wave, paired-row layout (row 2k + h in half h)
        const RowSrc zs = row_src(z + fbase, fbytes, lane);
        f32x4 Za[4], Zb[4], Ra[4], Rb[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { Za[k] = ldp(zs, sd0.n_first, k); Zb[k] = ldp(zs, sd1.n_first, k); }
        if (residual) {
            const RowSrc rs = row_src(residual + fbase, fbytes, lane);
#pragma unroll
            for (int k = 0; k < 4; ++k) { Ra[k] = ldp(rs, sd0.n_first, k); Rb[k] = ldp(rs, sd1.n_first, k); }
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) { Ra[k] = f32x4{0.f, 0.f, 0.f, 0.f}; Rb[k] = Ra[k]; }
        }
        const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + 4 * pl.q);
        const f32x4 sh = *reinterpret_cast<const f32x4*>(shift + 4 * pl.q);
        const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(out + fbase, 0, fbytes, 0x00020000);
        const int pair = pl.h * (C * 4) + pl.q * 16;
        const unsigned long long row0 = (unsigned long long)frame * a.n_per_frame;
        f32x4 oa[4], ob[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const unsigned long long ia = (row0 + sd0.n_first + 2 * k + pl.h) * C + 4 * pl.q;
            const unsigned long long ib = (row0 + sd1.n_first + 2 * k + pl.h) * C + 4 * pl.q;
            oa[k] = act4(Za[k], sc, sh, a, ia) + Ra[k];
            ob[k] = act4(Zb[k], sc, sh, a, ib) + Rb[k];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {                 // rows past the segment's count: offset -1, dropped by the bounds check
            const int va = (2 * k + pl.h) < sd0.cnt ? pair + k * (2 * C * 4) : -1;
            const int vb = (2 * k + pl.h) < sd1.cnt ? pair + k * (2 * C * 4) : -1;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4v, oa[k]), orsrc, va, sd0.n_first * (C * 4), 0);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4v, ob[k]), orsrc, vb, sd1.n_first * (C * 4), 0);
        }
        store_data_guard(oa);
        store_data_guard(ob);
        if (KOUT && sd0.pad1 > 0) {
            // parent k of the pair: children = columns 2k, 2k+1 (the two halves of register k) of both rows
            const f32x4 da = *reinterpret_cast<const f32x4*>(patsq + (size_t)sd0.pat * PATQ + 32 * pl.h + 4 * SLOT_SELF);
            const f32x4 db = *reinterpret_cast<const f32x4*>(patsq + (size_t)sd1.pat * PATQ + 32 * pl.h + 4 * SLOT_SELF);
            const __amdgpu_buffer_rsrc_t krsrc = __builtin_amdgcn_make_buffer_rsrc(
                kout + (size_t)frame * a.kid_rows * C, 0, a.kid_rows * (C * 4), 0x00020000);
            f32x4 ks[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const f32x4 t = da[k] * oa[k] + db[k] * ob[k];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(t[i]), __float_as_uint(t[i]), false, false);
                    ks[k][i] = __uint_as_float(r[0]) + __uint_as_float(r[1]);       // both halves hold the sum
                }
            }
            // lower half stores parents 0 and 1, upper half parents 2 and 3 (a whole 512-B row per half-wave and store)
            const f32x4 s0 = pl.h ? ks[2] : ks[0], s1 = pl.h ? ks[3] : ks[1];
            const int k0 = 2 * pl.h, k1 = 2 * pl.h + 1;
            const int v0 = k0 < sd0.pad1 ? k0 * (C * 4) + pl.q * 16 : -1;
            const int v1 = k1 < sd0.pad1 ? k1 * (C * 4) + pl.q * 16 : -1;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4v, s0), krsrc, v0, sd0.par0 * (C * 4), 0);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4v, s1), krsrc, v1, sd0.par0 * (C * 4), 0);
            store_data_guard(s0);
            store_data_guard(s1);
        }
        return;
    }
    // ---- ragged patches, coordinate nodes, the last rows of a frame (rare): row by row, two rows per wave instruction.
    // (No parent takes its children from such a segment: graph.hip switches the side buffer off otherwise.)
    const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + 4 * pl.q);
    const f32x4 sh = *reinterpret_cast<const f32x4*>(shift + 4 * pl.q);
#pragma unroll 1
    for (int e = 0; e < 2; ++e) {
        const int n0 = e ? sd1.n_first : sd0.n_first, cnt = e ? sd1.cnt : sd0.cnt;
#pragma unroll 1
        for (int u = pl.h; u < cnt; u += 2) {
            const size_t off = fbase + (size_t)(n0 + u) * C + 4 * pl.q;
            const f32x4 zz = *reinterpret_cast<const f32x4*>(z + off);
            const f32x4 rr = residual ? *reinterpret_cast<const f32x4*>(residual + off) : f32x4{0.f, 0.f, 0.f, 0.f};
            *reinterpret_cast<f32x4*>(out + off) = act4(zz, sc, sh, a, (unsigned long long)off) + rr;
        }
    }
}
#undef ldp

// One workgroup per tile (wave = segment pair).
template <bool KOUT>
__global__ __launch_bounds__(256) void k_bn_act_fwd_tiles(const float* __restrict__ z, const float* __restrict__ scale,
                                                          const float* __restrict__ shift, const float* __restrict__ residual,
                                                          float* __restrict__ out, float* __restrict__ kout,
                                                          const SegDesc* __restrict__ segs, const float* __restrict__ patsq,
                                                          const ActTileArgs a_) {
    const ActTileArgs a = resolved(a_);
    const int lane = threadIdx.x & 63, p = wave_id();
    const int tile = blockIdx.x;
    const int frame = tile / a.tiles_per_frame, t_in = tile - frame * a.tiles_per_frame;
    const SegDesc sd0 = segs[t_in * 8 + 2 * p];
    const SegDesc sd1 = segs[t_in * 8 + 2 * p + 1];
    act_pair<KOUT>(z, scale, shift, residual, out, kout, patsq, a, frame, sd0, sd1, lane);
}

}  // namespace eg

using namespace eg;

// BN affine + dropout + ReLU + residual over a topology handle's tiles; kout (nullable): child sums of the result.
// EG_ERR_UNSUPPORTED: not a topology handle / no child-sum side buffer for it (the caller uses the flat streaming kernel).
int eg_launch_bn_act_tiles(const eg_graph* g, int batch, const float* z, const float* scale, const float* shift, const float* residual,
                           int relu, float dropout_p, unsigned long long seed, float* out, float* kout, hipStream_t stream) {
    if (!g || g->kind != GRAPH_TOPO || (kout && g->kid_rows == 0)) return EG_ERR_UNSUPPORTED;
    if ((long long)g->n_nodes * (C * 4) >= (1ll << 31)) return EG_ERR_UNSUPPORTED;          // per-frame buffer descriptors
    if (int rc = eg_epoch_required(dropout_p)) return rc;
    ActTileArgs a{};
    a.n_per_frame = (int)g->n_nodes; a.tiles_per_frame = g->n_tiles; a.batch = batch; a.kid_rows = g->kid_rows;
    a.relu = relu; a.p = dropout_p; a.inv_keep = dropout_p > 0.f ? 1.0f / (1.0f - dropout_p) : 1.0f; a.seed = seed; a.epoch = eg_epoch_ptr();
    const long long n_tiles = (long long)g->n_tiles * batch;
    if (n_tiles <= 0) return EG_OK;
    if (n_tiles >= (1ll << 31)) return eg::set_error(EG_ERR_ARG, "too many tiles");
    // (one workgroup per tile; a persistent grid with descriptor prefetch measured no faster in round 4 and is gone)
    if (kout) hipLaunchKernelGGL(k_bn_act_fwd_tiles<true>, dim3((unsigned)n_tiles), dim3(256), 0, stream, z, scale, shift, residual, out,
                                 kout, g->segs_dev, g->patsq_dev, a);
    else hipLaunchKernelGGL(k_bn_act_fwd_tiles<false>, dim3((unsigned)n_tiles), dim3(256), 0, stream, z, scale, shift, residual, out,
                            kout, g->segs_dev, g->patsq_dev, a);
    EG_HIP_TRY(hipGetLastError());
    return EG_OK;
}

extern "C" int eg_bn_act_fwd_tiles(const eg_graph* g, int batch, const float* z, const float* scale, const float* shift,
                                   const float* residual, int relu, float dropout_p, uint64_t seed, float* out, float* kidsum_out,
                                   eg_stream_t stream) {
    if (!g || !z || !scale || !shift || !out || batch < 1) return set_error(EG_ERR_ARG, "bad argument");
    if (dropout_p < 0.f || dropout_p >= 1.f) return set_error(EG_ERR_ARG, "dropout_p must be in [0, 1)");
    if (out == z || out == residual || kidsum_out == out) return set_error(EG_ERR_ARG, "out must not alias an input, kidsum_out must not alias out");
    const int rc = eg_launch_bn_act_tiles(g, batch, z, scale, shift, residual, relu, dropout_p, seed, out, kidsum_out, (hipStream_t)stream);
    if (rc == EG_ERR_UNSUPPORTED)
        return set_error(EG_ERR_UNSUPPORTED, "the tile-order activation pass needs a topology handle (with eg_graph_kidsum_rows() > 0 for child sums)");
    return rc;
}
