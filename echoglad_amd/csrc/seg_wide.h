// Wide segment aggregation for the producer waves of the producer/consumer layer kernel (gfx950).
// Same math and paired-row layout as stencil_row8 (tile.h), but written for a 256-VGPR budget: ALL row loads
// of a segment (self, edges, above, below, parents = 16 wave loads) are issued at once, for two segments back
// to back, so a producer wave pays one memory round trip per tile (two on aux levels, for the children).
#pragma once
#include "tile.h"

namespace eg {

struct SegW {
    f32x4 S[4], Le, Re, U[4], D[4], P[2];
    float wa, wb;              // lane (u, s) holds the weight of slot s (wa: 0..7, wb: 8 + (s & 1)) of node u
    int mode;                  // 0: no rows, 1: fast path (loads in flight), 2: per-node slow path
};

// Everything that depends only on the topology (run bases, weight pattern) was worked out on the host when the
// handle was made (graph.hip: SegDesc): the producer issues one scalar descriptor load, one coalesced pattern load
// and the 16 row loads -- no decode arithmetic in front of the memory pipeline.
__device__ inline void segw_issue(const SegDesc& sd, const float* __restrict__ pats, const float* __restrict__ xf,
                                  int lane, SegW& A) {
    A.mode = sd.mode;
    A.wa = A.wb = 0.f;
    if (A.mode != 1) return;
    const float* pw = pats + (size_t)sd.pat * 128;
    A.wa = pw[lane];
    A.wb = pw[64 + lane];
    const PairLane pl{lane >> 5, lane & 31};
    const unsigned os = pair_off(sd.n_first, pl);
    const unsigned ou = pair_off(sd.up0, pl);
    const unsigned od = pair_off(sd.down0, pl);
    const unsigned op = pair_off(sd.par0, pl);
    const unsigned oL = bcast_off(sd.left, pl);
    const unsigned oR = bcast_off(sd.right, pl);
#pragma unroll
    for (int k = 0; k < 4; ++k) A.S[k] = ld4(xf, os, k);
    A.Le = *reinterpret_cast<const f32x4*>(xf + oL);
    A.Re = *reinterpret_cast<const f32x4*>(xf + oR);
#pragma unroll
    for (int k = 0; k < 4; ++k) A.U[k] = ld4(xf, ou, k);
#pragma unroll
    for (int k = 0; k < 4; ++k) A.D[k] = ld4(xf, od, k);
    A.P[0] = ld4(xf, op, 0);
    A.P[1] = ld4(xf, op, 1);
}

// self / left / right / above / below / parents -> acc (children are added by segw_kids)
__device__ inline void segw_rows(int lane, float wa, const f32x4 (&S)[4], const f32x4& Le, const f32x4& Re, const f32x4 (&U)[4],
                                 const f32x4 (&D)[4], const f32x4 (&P)[2], f32x4 (&acc)[4], float* s_x, int rl);

__device__ inline void segw_main(const Topo* __restrict__ T, const SegDesc& sd,
                                 const float* __restrict__ xf, const float* __restrict__ dis, int lane, SegW& A,
                                 f32x4 (&acc)[4], float* s_a, float* s_x, int rl) {
    if (A.mode == 2) {
        for (int u = 0; u < sd.cnt; ++u) {
            *reinterpret_cast<f32x2*>(&s_a[(rl + u) * LDA + 2 * lane]) = agg_stencil(T, xf, dis, sd.n_first + u, lane);
            if (s_x) *reinterpret_cast<f32x2*>(&s_x[(rl + u) * LDA + 2 * lane]) = load_row2(xf, sd.n_first + u, lane);
        }
        return;
    }
    if (A.mode != 1) return;
    segw_rows(lane, A.wa, A.S, A.Le, A.Re, A.U, A.D, A.P, acc, s_x, rl);
}

// acc = sum over self / left / right / above / below / parent rows of one segment (children: segw_kids_add)
__device__ inline void segw_rows(int lane, float wa, const f32x4 (&S)[4], const f32x4& Le, const f32x4& Re, const f32x4 (&U)[4],
                                 const f32x4 (&D)[4], const f32x4 (&P)[2], f32x4 (&acc)[4], float* s_x, int rl) {
    const PairLane pl{lane >> 5, lane & 31};
    const bool up_half = pl.h != 0;
    const int hb4 = pl.h * 32;
    {
        f32x4 Mk = seam(Le, S[0], up_half);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const f32x4 Mn = seam(S[k], k < 3 ? S[k < 3 ? k + 1 : 3] : Re, up_half);
            f32x4 a = pair_w(wa, hb4, k, 4) * Mn;
            a += pair_w(wa, hb4, k, 3) * Mk;
            a += pair_w(wa, hb4, k, 0) * S[k];
            acc[k] = a;
            Mk = Mn;
        }
        if (s_x) {
#pragma unroll
            for (int k = 0; k < 4; ++k) *reinterpret_cast<f32x4*>(&s_x[(rl + 2 * k + pl.h) * LDA + 4 * pl.q]) = S[k];
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[k] += pair_w(wa, hb4, k, 1) * U[k];
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[k] += pair_w(wa, hb4, k, 2) * D[k];
    {
        f32x4 plo[2], pup[2];
        halves(P[0], plo[0], pup[0]);
        halves(P[1], plo[1], pup[1]);
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[k] += pair_w(wa, hb4, k, 5) * ((k & 1) ? pup[k >> 1] : plo[k >> 1]);
    }
}

// ---- two vertically adjacent segments (patch rows 2p, 2p+1) handled by one wave ----------------------------------
// The rows below segment a ARE segment b and the rows above b are a; both share their 4 parents: 22 wave loads
// instead of 32.  SegDesc::pad0 of the upper segment says when this holds (host, graph.hip).
struct SegPair {
    f32x4 Sa[4], Sb[4], La, Ra, Lb, Rb, U[4], D[4], P[2];
    float wa_a, wb_a, wa_b, wb_b;
};

__device__ inline void segp_issue(const SegDesc& sa, const SegDesc& sb, const float* __restrict__ pats,
                                  const float* __restrict__ xf, int lane, SegPair& A) {
    const float* pwa = pats + (size_t)sa.pat * 128;
    const float* pwb = pats + (size_t)sb.pat * 128;
    A.wa_a = pwa[lane]; A.wb_a = pwa[64 + lane];
    A.wa_b = pwb[lane]; A.wb_b = pwb[64 + lane];
    const PairLane pl{lane >> 5, lane & 31};
    const unsigned osa = pair_off(sa.n_first, pl), osb = pair_off(sb.n_first, pl);
    const unsigned ou = pair_off(sa.up0, pl), od = pair_off(sb.down0, pl), op = pair_off(sa.par0, pl);
#pragma unroll
    for (int k = 0; k < 4; ++k) A.Sa[k] = ld4(xf, osa, k);
#pragma unroll
    for (int k = 0; k < 4; ++k) A.Sb[k] = ld4(xf, osb, k);
    A.La = *reinterpret_cast<const f32x4*>(xf + bcast_off(sa.left, pl));
    A.Ra = *reinterpret_cast<const f32x4*>(xf + bcast_off(sa.right, pl));
    A.Lb = *reinterpret_cast<const f32x4*>(xf + bcast_off(sb.left, pl));
    A.Rb = *reinterpret_cast<const f32x4*>(xf + bcast_off(sb.right, pl));
#pragma unroll
    for (int k = 0; k < 4; ++k) A.U[k] = ld4(xf, ou, k);
#pragma unroll
    for (int k = 0; k < 4; ++k) A.D[k] = ld4(xf, od, k);
    A.P[0] = ld4(xf, op, 0);
    A.P[1] = ld4(xf, op, 1);
}

struct SegKids { f32x4 Ca[8], Cb[8]; };      // child rows 2r (Ca) and 2r+1 (Cb): 8 pairs = 16 node rows each

__device__ inline void segw_kids_issue(const SegDesc& sd, const float* __restrict__ xf, int lane, SegKids& K) {
    const PairLane pl{lane >> 5, lane & 31};
    const unsigned oc0 = pair_off(sd.c0, pl);
    const unsigned oc1 = pair_off(sd.c1, pl);
    const unsigned oc2 = pair_off(sd.c2, pl);
    const unsigned oc3 = pair_off(sd.c3, pl);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        K.Ca[k] = ld4(xf, oc0, k); K.Ca[4 + k] = ld4(xf, oc1, k);
        K.Cb[k] = ld4(xf, oc2, k); K.Cb[4 + k] = ld4(xf, oc3, k);
    }
}

__device__ inline void segw_kids_add(int lane, float wa, float wb, const SegKids& K, f32x4 (&acc)[4]) {
    const PairLane pl{lane >> 5, lane & 31};
    const bool up_half = pl.h != 0;
    const int hs4 = pl.h * 4;                                  // slot 6 -> 7 / 8 -> 9 for the upper half
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        f32x4 red[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int u = 2 * k + e;                            // child pair u holds node u's two children of a child row
            const float w67 = __int_as_float(__builtin_amdgcn_ds_bpermute(hs4 + (u * 8 + 6) * 4, __float_as_int(wa)));
            const float w89 = __int_as_float(__builtin_amdgcn_ds_bpermute(hs4 + (u * 8 + 0) * 4, __float_as_int(wb)));
            const f32x4 t = w67 * K.Ca[u] + w89 * K.Cb[u];
            f32x4 tl, tu;
            halves(t, tl, tu);
            red[e] = tl + tu;
        }
        acc[k] += sel(up_half, red[0], red[1]);
    }
}

__device__ inline void segw_store(int lane, float wa, const f32x4 (&acc)[4], float* s_a, int rl) {
    const PairLane pl{lane >> 5, lane & 31};
    const int hb4 = pl.h * 32;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const f32x4 o = acc[k] * pair_w(wa, hb4, k, 0);
        *reinterpret_cast<f32x4*>(&s_a[(rl + 2 * k + pl.h) * LDA + 4 * pl.q]) = o;       // rows >= cnt are padding rows
    }
}

}  // namespace eg
