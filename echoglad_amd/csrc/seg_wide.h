// Paired-segment aggregation for the producer waves of the producer/consumer layer kernel (gfx950).
// Same math and paired-row register layout as stencil_row8 (tile.h), written for a 256-VGPR budget: one wave
// handles patch rows 2p and 2p+1 of an 8x8 patch and has ALL their row loads in flight at once, so it pays one
// memory round trip per tile.  Everything that depends only on the topology comes precomputed from the host
// (graph.hip): run bases in a SegDesc, neighbour weights as an index into a pattern table that the kernel keeps
// in LDS in "quad" layout -- [pattern][row parity h][slot][k] = weight of (node 2k + h, slot) -- so that one
// ds_read_b128 yields the four weights a lane needs for one neighbour class.
#pragma once
#include "tile.h"

namespace eg {

constexpr int PATQ = 64;       // floats per pattern in quad layout
enum { SLOT_SELF = 0, SLOT_UP = 1, SLOT_DOWN = 2, SLOT_LEFT = 3, SLOT_RIGHT = 4, SLOT_PARENT = 5, SLOT_HASKIDS = 6, SLOT_EDGE = 7 };

// Row loads of the producers: buffer loads off a frame descriptor in SGPRs.  The lane's part of the address is one VGPR
// for the whole kernel, the run's first row goes into the scalar offset and the pair index into the immediate, so a load
// costs no VALU instruction (a flat 64-bit address costs three per load, and fp32 VALU work takes MFMA issue slots).
// Rows past the frame read as zero.
typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
struct RowSrc {
    __amdgpu_buffer_rsrc_t r;
    int pair;        // byte offset of this lane inside a row pair: h * 512 + 16 q
    int q16;         // 16 q
    int h;
};
__device__ inline RowSrc row_src(const float* base, int bytes, int lane) {
    return RowSrc{__builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, bytes, 0x00020000),
                  (lane >> 5) * (C * 4) + (lane & 31) * 16, (lane & 31) * 16, lane >> 5};
}
// rows row0 + 2k + h (row0 wave-uniform)
__device__ inline f32x4 ldp(const RowSrc& s, int row0, int k) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(s.r, s.pair + k * (2 * C * 4), row0 * (C * 4), 0));
}
// row `lower` in the lower half of the wave, row `upper` in the upper half (both wave-uniform)
__device__ inline f32x4 ld_two_rows(const RowSrc& s, int lower, int upper) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(s.r, s.q16 + s.h * ((upper - lower) * (C * 4)), lower * (C * 4), 0));
}

// Two wait states behind 16-byte buffer stores whose scalar offset is a register, with the stores' data registers as inputs: the
// registers stay allocated (nothing can write them) until the wait states have passed (see segw_store_agg).
__device__ inline void store_data_guard(const f32x4& a) {
    asm volatile("s_nop 1" :: "v"(a.x), "v"(a.y), "v"(a.z), "v"(a.w) : "memory");
}
__device__ inline void store_data_guard(const f32x4 (&o)[4]) {
    asm volatile("s_nop 1" :: "v"(o[0].x), "v"(o[0].y), "v"(o[0].z), "v"(o[0].w), "v"(o[1].x), "v"(o[1].y), "v"(o[1].z), "v"(o[1].w),
                 "v"(o[2].x), "v"(o[2].y), "v"(o[2].z), "v"(o[2].w), "v"(o[3].x), "v"(o[3].y), "v"(o[3].z), "v"(o[3].w) : "memory");
}

__device__ inline f32x4 quad_w(const float* wq, int slot) { return *reinterpret_cast<const f32x4*>(wq + 4 * slot); }

// acc = sum over self / left / right / above / below / parent rows of one segment (children are added afterwards).
// wq: this lane's quad weights (LDS): pattern base + 32 * (lane >> 5).
// The segment's own rows go to LDS first (s_t: the residual stash of the tile, or the wave's rows of the A tile, which
// segw_store overwrites afterwards) and the left / right neighbours come back as the rows one above / below, shifted by
// the LDS address: no cross-lane VALU work (fp32 VALU and the MFMA share one issue slot per SIMD).  The
// segment's first / last node take their outer neighbour from LR (lower half: the row left of the segment, upper half:
// the row right of it) with the weights of SLOT_EDGE; SLOT_LEFT / SLOT_RIGHT are zero there (graph.hip).
__device__ inline void segw_rows(int lane, const float* wq, const f32x4 (&S)[4], const f32x4& LR,
                                 const f32x4 (&U)[4], const f32x4 (&D)[4], const f32x4 (&P)[4], f32x4 (&acc)[4],
                                 float* s_t, int rl) {
    const PairLane pl{lane >> 5, lane & 31};
    {
        float* row0 = s_t + rl * LDA + 4 * pl.q;           // row 0 of the segment, this lane's channels
        float* rowh = row0 + pl.h * LDA;                   // row h
#pragma unroll
        for (int k = 0; k < 4; ++k) *reinterpret_cast<f32x4*>(rowh + 2 * k * LDA) = S[k];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");      // LDS operations of one wave complete in order
        const f32x4 w0 = quad_w(wq, SLOT_SELF), w3 = quad_w(wq, SLOT_LEFT), w4 = quad_w(wq, SLOT_RIGHT), we = quad_w(wq, SLOT_EDGE);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            // node 2k + h: left neighbour = row 2k + h - 1, right neighbour = row 2k + h + 1, clamped into the segment
            const f32x4 L = *reinterpret_cast<const f32x4*>(k == 0 ? row0 : rowh + (2 * k - 1) * LDA);
            const f32x4 R = *reinterpret_cast<const f32x4*>(k == 3 ? row0 + 7 * LDA : rowh + (2 * k + 1) * LDA);
            f32x4 a = w4[k] * R;
            a += w3[k] * L;
            a += w0[k] * S[k];
            acc[k] = a;
        }
        acc[0] += we[0] * LR;
        acc[3] += we[3] * LR;
    }
    {
        const f32x4 w1 = quad_w(wq, SLOT_UP);
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[k] += w1[k] * U[k];
    }
    {
        const f32x4 w2 = quad_w(wq, SLOT_DOWN);
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[k] += w2[k] * D[k];
    }
    {   // parents: nodes 2k and 2k+1 share parent k, whose row both halves of P[k] hold
        const f32x4 w5 = quad_w(wq, SLOT_PARENT);
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[k] += w5[k] * P[k];
    }
}

// ---- two vertically adjacent segments (patch rows 2p, 2p+1) handled by one wave ----------------------------------
// The rows below segment a ARE segment b and the rows above b are a; both share their 4 parents, and the two edge
// rows of a segment come in one load: 22 wave loads instead of 32.  SegDesc::pad0 of the upper segment says when this
// holds (host, graph.hip).
struct SegPair {
    f32x4 Sa[4], Sb[4], LRa, LRb, U[4], D[4], P[4];   // LR: lower half = row left of the segment, upper half = row right of it
};

__device__ inline void segp_issue(const SegDesc& sa, const SegDesc& sb, const RowSrc& xs, SegPair& A) {
#pragma unroll
    for (int k = 0; k < 4; ++k) A.Sa[k] = ldp(xs, sa.n_first, k);
#pragma unroll
    for (int k = 0; k < 4; ++k) A.Sb[k] = ldp(xs, sb.n_first, k);
    // one 1-KB wave load fetches both edge rows of a segment, each in the half whose lanes use it (segw_rows)
    A.LRa = ld_two_rows(xs, sa.left, sa.right);
    A.LRb = ld_two_rows(xs, sb.left, sb.right);
#pragma unroll
    for (int k = 0; k < 4; ++k) A.U[k] = ldp(xs, sa.up0, k);
#pragma unroll
    for (int k = 0; k < 4; ++k) A.D[k] = ldp(xs, sb.down0, k);
#pragma unroll
    for (int k = 0; k < 4; ++k) A.P[k] = ld_two_rows(xs, sa.par0 + k, sa.par0 + k);     // the same row in both halves
}

// ---- 'grid-diagonal' levels (8-neighbour grids: reference src/core/datasets.py:1469-1475, :1494-1500) --------------------------
// A neighbour's weight d_j belongs to the SOURCE node, not to the (target, source) pair: a row scaled once by its own nodes'
// d_j serves its three targets (the node below / above it and the two diagonal ones) unweighted.  So every row set -- the
// pair's own two rows, the row above, the row below -- goes through LDS scaled, and a target adds the entries one to the
// left, at and one to the right of its column ("sum3").  What lies outside the segment comes from the edge rows (one load per
// row set: left edge in the lower half-wave, right edge in the upper), whose weights are zero where the grid ends.  Pattern
// slots of a diagonal segment (graph.hip): SELF / UP / DOWN = d of the node itself / above / below, EDGE = own row's edge
// nodes, slots 3 / 4 (LEFT / RIGHT on plain levels) = the edge nodes of the row above / below.
constexpr int SLOT_EDGE_U = SLOT_LEFT, SLOT_EDGE_D = SLOT_RIGHT;
constexpr int SEG_AUX_DIAG = 2;                       // SegDesc::aux bit 1: the segment's level is 'grid-diagonal'

struct SegPairDiag { f32x4 LRu, LRd; };               // edge rows of the row above segment a / below segment b

__device__ inline void segp_diag_issue(const SegDesc& sa, const SegDesc& sb, const RowSrc& xs, int last_row, SegPairDiag& E) {
    const int ul = sa.up0 > 0 ? sa.up0 - 1 : 0, ur = sa.up0 + 8 < last_row ? sa.up0 + 8 : last_row;
    const int dl = sb.down0 > 0 ? sb.down0 - 1 : 0, dr = sb.down0 + 8 < last_row ? sb.down0 + 8 : last_row;
    E.LRu = ld_two_rows(xs, ul, ur);
    E.LRd = ld_two_rows(xs, dl, dr);
}

// out[k] (+)= R'[c-1] + R'[c] + R'[c+1] for node c = 2k + h of an 8-node row set R scaled by w (R' = w R); the entries left of
// node 0 / right of node 7 are the edge rows LR scaled by we (non-zero only in the half that holds that edge's target).
// s_r: 8 LDS rows this wave may use for the set (LDS operations of one wave complete in order).
template <bool ADD>
__device__ inline void row_sum3(int lane, const f32x4 (&R)[4], const f32x4& w, const f32x4& LR, const f32x4& we, float* s_r, f32x4 (&out)[4]) {
    const PairLane pl{lane >> 5, lane & 31};
    float* row0 = s_r + 4 * pl.q;
    float* rowh = row0 + pl.h * LDA;
    f32x4 Rs[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        Rs[k] = w[k] * R[k];
        *reinterpret_cast<f32x4*>(rowh + 2 * k * LDA) = Rs[k];
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    const float in_l = (float)pl.h, in_r = (float)(1 - pl.h);      // node 0 has no left, node 7 no right neighbour inside the segment
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const f32x4 L = *reinterpret_cast<const f32x4*>(k == 0 ? row0 : rowh + (2 * k - 1) * LDA);
        const f32x4 Rr = *reinterpret_cast<const f32x4*>(k == 3 ? row0 + 7 * LDA : rowh + (2 * k + 1) * LDA);
        f32x4 t = Rs[k];
        if (k == 0) t += in_l * L; else t += L;
        if (k == 3) t += in_r * Rr; else t += Rr;
        if (ADD) out[k] += t; else out[k] = t;
    }
    out[0] += we[0] * LR;
    out[3] += we[3] * LR;
}

// both segments of a pair on a diagonal level: acc = sum over the 8 grid neighbours + self + parents (children follow)
__device__ inline void segp_diag_rows(int lane, const float* wqa, const float* wqb, const SegPair& A, const SegPairDiag& E,
                                      f32x4 (&acc0)[4], f32x4 (&acc1)[4], float* s_rows) {
    // the pair's own rows serve both segments (own row + the row below for a, above for b): their sum T goes into acc0, the row
    // below the pair is added to a copy (acc1), the row above to acc0 itself
    row_sum3<false>(lane, A.Sa, quad_w(wqa, SLOT_SELF), A.LRa, quad_w(wqa, SLOT_EDGE), s_rows, acc0);
    row_sum3<true>(lane, A.Sb, quad_w(wqb, SLOT_SELF), A.LRb, quad_w(wqb, SLOT_EDGE), s_rows + 8 * LDA, acc0);
    const f32x4 pa = quad_w(wqa, SLOT_PARENT), pb = quad_w(wqb, SLOT_PARENT);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        acc1[k] = acc0[k] + pb[k] * A.P[k];
        acc0[k] += pa[k] * A.P[k];
    }
    pin_acc4(acc0);
    pin_acc4(acc1);
    row_sum3<true>(lane, A.D, quad_w(wqb, SLOT_DOWN), E.LRd, quad_w(wqb, SLOT_EDGE_D), s_rows + 8 * LDA, acc1);
    row_sum3<true>(lane, A.U, quad_w(wqa, SLOT_UP), E.LRu, quad_w(wqa, SLOT_EDGE_U), s_rows, acc0);
}

// Aux levels, chained layers: the children's contribution of node n is ONE row of the side buffer (row n; the previous
// layer's epilogue summed dis[c] * h[c] over the four children), so the children are one more run of 8 rows per
// segment, loaded once the main stage has freed its registers.
struct SegKidsum { f32x4 Ka[4], Kb[4]; };

__device__ inline void segp_kidsum_issue(const SegDesc& sa, const SegDesc& sb, const RowSrc& ks, SegKidsum& K) {
#pragma unroll
    for (int k = 0; k < 4; ++k) { K.Ka[k] = ldp(ks, sa.n_first, k); K.Kb[k] = ldp(ks, sb.n_first, k); }
}

__device__ inline void segp_kidsum_add(const SegKidsum& K, const float* wqa, const float* wqb, f32x4 (&acc0)[4], f32x4 (&acc1)[4]) {
    const f32x4 fa = quad_w(wqa, SLOT_HASKIDS), fb = quad_w(wqb, SLOT_HASKIDS);       // 1.0 / 0.0
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        acc0[k] += fa[k] * K.Ka[k];
        acc1[k] += fb[k] * K.Kb[k];
    }
}

// ---- children pulled as rows (first layer of a stack, or unchained calls) ------------------------------------------
struct SegKids { f32x4 Ca[8], Cb[8]; float wa, wb; };      // child rows 2r (Ca) and 2r+1 (Cb): 8 pairs = 16 node rows each

// pats: the lane-layout pattern table in global memory (lane (u, s): weight of slot s / slot 8 + (s & 1) of node u)
__device__ inline void segw_kids_issue(const SegDesc& sd, const float* __restrict__ pats, const RowSrc& xs,
                                       int lane, SegKids& K) {
    const float* pw = pats + (size_t)sd.pat * 128;
    K.wa = pw[lane];
    K.wb = pw[64 + lane];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        K.Ca[k] = ldp(xs, sd.c0, k); K.Ca[4 + k] = ldp(xs, sd.c1, k);
        K.Cb[k] = ldp(xs, sd.c2, k); K.Cb[4 + k] = ldp(xs, sd.c3, k);
    }
}

__device__ inline void segw_kids_add(int lane, const SegKids& K, f32x4 (&acc)[4]) {
    const PairLane pl{lane >> 5, lane & 31};
    const int hs4 = pl.h * 4;                                  // slot 6 -> 7 / 8 -> 9 for the upper half
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        f32x4 t[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int u = 2 * k + e;                            // child pair u holds node u's two children of a child row
            const float w67 = __int_as_float(__builtin_amdgcn_ds_bpermute(hs4 + (u * 8 + 6) * 4, __float_as_int(K.wa)));
            const float w89 = __int_as_float(__builtin_amdgcn_ds_bpermute(hs4 + (u * 8 + 0) * 4, __float_as_int(K.wb)));
            t[e] = w67 * K.Ca[u] + w89 * K.Cb[u];
        }
        // node 2k (lower half of the wave) wants both halves of t[0], node 2k+1 (upper half) both halves of t[1]:
        // one swap gives [t0.lo | t1.lo] and [t0.up | t1.up], whose sum is exactly that
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(t[0][i]), __float_as_uint(t[1][i]), false, false);
            acc[k][i] += __uint_as_float(r[0]) + __uint_as_float(r[1]);
        }
    }
}

// final scaling by the node's own (deg+1)^-1/2 and hand-over to the consumers' A tile
__device__ inline void segw_store(int lane, const float* wq, const f32x4 (&acc)[4], float* s_a, int rl) {
    const PairLane pl{lane >> 5, lane & 31};
    const f32x4 w0 = quad_w(wq, SLOT_SELF);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const f32x4 o = acc[k] * w0[k];
        *reinterpret_cast<f32x4*>(&s_a[(rl + 2 * k + pl.h) * LDA + 4 * pl.q]) = o;       // rows >= cnt are padding rows
    }
}

// TRAIN: the same rows to global memory as well (A_hat x, kept for the weight gradient): two rows per store, rows that hold
// no node get offset -1 (out of range: dropped)
__device__ inline void segw_store_agg(int lane, const float* wq, const f32x4 (&acc)[4], __amdgpu_buffer_rsrc_t agg, int n_first, int cnt) {
    const PairLane pl{lane >> 5, lane & 31};
    const f32x4 w0 = quad_w(wq, SLOT_SELF);
    const int pair = pl.h * (C * 4) + pl.q * 16;
    f32x4 o[4];
    int voff[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        o[k] = acc[k] * w0[k];
        voff[k] = (2 * k + pl.h) < cnt ? pair + k * (2 * C * 4) : -1;
    }
    // All four values exist before the first store and nothing writes their registers until two wait states after the last
    // one: a 16-byte buffer store with a REGISTER scalar offset still reads its data when the next instruction issues -- a
    // VALU write of those registers right behind it corrupted the stored rows (LLVM's hazard table only covers the
    // immediate-offset form; measured on gfx950).
#pragma unroll
    for (int k = 0; k < 4; ++k) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4v, o[k]), agg, voff[k], n_first * (C * 4), 0);
    store_data_guard(o);
}

}  // namespace eg
