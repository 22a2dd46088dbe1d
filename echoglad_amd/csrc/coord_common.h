// Device pieces of the coordinate-graph resampling (reference src/core/models.py:539-553) shared by coord.hip (the stand-alone
// launches) and coord_mlp.hip (the single-workgroup forms that run the landmark MLP and the resampling in ONE launch).
#pragma once
#include "train_common.h"


namespace eg {

struct BilArgs {
    int batch, points, frame;     // points per frame (4 in the reference)
    long long n_per_frame, main_base;
    long long row_stride;         // floats between two frames' sample rows in out / dout (points * 128: a packed [batch * points, 128] array)
};

struct Taps {
    int i[2];
    float w[2], dw[2];            // hat weight and its derivative wrt the coordinate
};

__device__ inline Taps taps_1d(float c, int F) {
    Taps t;
    const float f = floorf(c);
    const int i0 = (int)f;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int i = i0 + k;
        const float d = c - (float)i;
        const float w = 1.0f - fabsf(d);
        const bool ok = i >= 0 && i < F && w > 0.0f;
        t.i[k] = ok ? i : 0;
        t.w[k] = ok ? w : 0.0f;
        t.dw[k] = ok ? (d > 0.f ? -1.0f : (d < 0.f ? 1.0f : 0.0f)) : 0.0f;   // d/dc relu(1-|c-i|); 0 at the kink like torch.abs
    }
    return t;
}

// what the taps add to a dy whose BatchNorm-backward sums exist already (eg_gcn_layer_bwd_lower): the additions' own sums
struct TapSums {
    const float* z;           // NULL: not wanted
    const float* bn;          // mean, invstd, scale, shift
    int relu;
    float p, inv_keep;
    unsigned long long seed;
    const unsigned long long* epoch;
    float* out;               // [batch][2][128]
};

// One point sampled by one wave (lane = channel pair): `main` = the frame's main-grid rows + 2 * lane.
__device__ inline f32x2 bilinear_sample(const float* __restrict__ main, float ch, float cw, int F) {
    const Taps th = taps_1d(ch, F), tw = taps_1d(cw, F);
    f32x2 v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = *reinterpret_cast<const f32x2*>(main + ((size_t)th.i[k >> 1] * F + tw.i[k & 1]) * C);
    f32x2 acc = {0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 4; ++k) acc += (th.w[k >> 1] * tw.w[k & 1]) * v[k];
    return acc;
}

// Backward of the FOUR points of one frame by one wave (lane = channel pair).  The sequential walk (landmark after landmark, tap after
// tap: load a row of dh, add, store, because two taps may hit the same pixel) is 16 dependent memory round trips -- 15 us for a
// kernel with 8 KB of work.  Here the frame's rows are loaded in three rounds (h; dh; z) and the read-modify-writes are resolved
// in registers: a tap that hits the row of an earlier tap continues from that tap's value; stores go out in tap order.  Same values,
// same order of additions as the walk.
//   dout_f = the frame's 4 gradient rows;  cf = the frame's 8 coordinates (global or LDS);  fbase = element offset of the frame's
//   main grid in h / dh / ts.z (both wave-uniform: the lane's channel pair is added here);  gh / gw: d coords (h, w) per point, reduced over the wave;  ts1 / ts2: the lane's
//   tap sums (TS).
template <bool TS>
__device__ inline void bilinear_bwd_frame4(const float* __restrict__ dout_f, const float* __restrict__ h, const float* cf, float* dh,
                                           size_t fbase, int F, const TapSums& ts, unsigned long long tseed, int lane,
                                           float (&gh)[4], float (&gw)[4], f32x2& ts1, f32x2& ts2) {
    // buffer accesses: one descriptor per array over the frame's main grid (scalar registers), the tap's row as the scalar byte offset,
    // the lane's channel pair as the only vector offset -- no address registers per tap
    typedef unsigned u32x2v __attribute__((ext_vector_type(2)));
    const int lane_b = 8 * lane, grid_bytes = F * F * C * 4;
    const __amdgpu_buffer_rsrc_t rs_h = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(h + fbase), 0, grid_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_d = __builtin_amdgcn_make_buffer_rsrc(dh + fbase, 0, dh ? grid_bytes : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_z = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ts.z + fbase), 0, TS ? grid_bytes : 0, 0x00020000);
    // the taps depend on the coordinates only: row offset and weight of each go to scalar registers (one copy per wave, and the tests
    // on them below become scalar branches); the two d-coordinate coefficients per tap live until the dot products
    f32x2 g[4];
    int rowb[16];
    float w[16], cgh[16], cgw[16];
    f32x2 v[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const Taps th = taps_1d(cf[2 * q], F), tw = taps_1d(cf[2 * q + 1], F);
        g[q] = *reinterpret_cast<const f32x2*>(dout_f + q * C + 2 * lane);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int k = 4 * q + kk, ka = kk >> 1, kb = kk & 1;
            rowb[k] = __builtin_amdgcn_readfirstlane((th.i[ka] * F + tw.i[kb]) * (C * 4));
            w[k] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, th.w[ka] * tw.w[kb])));
            cgh[k] = th.dw[ka] * tw.w[kb];
            cgw[k] = th.w[ka] * tw.dw[kb];
            v[k] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rs_h, lane_b, rowb[k], 0));
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float a = 0.f, b = 0.f;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int k = 4 * q + kk;
            const float dot = g[q].x * v[k].x + g[q].y * v[k].y;
            a += cgh[k] * dot;
            b += cgw[k] * dot;
        }
        gh[q] = a;
        gw[q] = b;
    }
    // second round of loads, in the registers the rows of h have left: the rows of dh; third round (TS): the rows of z.  (The barriers
    // keep the scheduler from hoisting all three into the first: 160 live registers, and the single-workgroup callers run 16 waves =
    // 128 registers per lane.)
    __builtin_amdgcn_sched_barrier(0);
    if (dh) {
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rs_d, lane_b, rowb[k], 0));
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            if (w[k] == 0.f) continue;              // (wave-uniform: the weights depend on the coordinates only)
            const f32x2 val = v[k] + w[k] * g[k >> 2];
#pragma unroll
            for (int j = k + 1; j < 16; ++j)
                if (rowb[j] == rowb[k]) v[j] = val;
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2v, val), rs_d, lane_b, rowb[k], 0);
        }
    }
    if (TS) {
        __builtin_amdgcn_sched_barrier(0);
        const f32x2 tmn = *reinterpret_cast<const f32x2*>(ts.bn + 2 * lane), tis = *reinterpret_cast<const f32x2*>(ts.bn + C + 2 * lane);
        const f32x2 tsc = *reinterpret_cast<const f32x2*>(ts.bn + 2 * C + 2 * lane), tsh = *reinterpret_cast<const f32x2*>(ts.bn + 3 * C + 2 * lane);
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rs_z, lane_b, rowb[k], 0));
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            if (w[k] == 0.f) continue;
            f32x2 add = w[k] * g[k >> 2];
            if (ts.p > 0.f) add *= keep_scale2(tseed, (unsigned long long)fbase + (unsigned)((rowb[k] + lane_b) >> 2), ts.p, ts.inv_keep);
            const f32x2 y = v[k] * tsc + tsh, xh = (v[k] - tmn) * tis;
            if (ts.relu) { add.x = y.x > 0.f ? add.x : 0.f; add.y = y.y > 0.f ? add.y : 0.f; }
            ts1 += add;
            ts2 += add * xh;
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
        for (int o = 32; o > 0; o >>= 1) { gh[q] += __shfl_xor(gh[q], o); gw[q] += __shfl_xor(gw[q], o); }
}

}  // namespace eg
