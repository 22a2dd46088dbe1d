// Device building blocks shared by the layer and classifier kernels (gfx950 only).
//
// Work decomposition of one workgroup (256 threads = 4 waves) on a TILE=64-row tile:
//   phase 1  every wave produces 16 rows of the A tile [64][128] fp32 in LDS
//            (plain copy, CSR pull-aggregate or implicit-stencil pull-aggregate);
//            one wave handles one node row at a time, lane l owns channels 2l, 2l+1,
//            so node ids / neighbour ids / normalisation weights are wave-uniform and
//            live on the scalar unit (s_load), and every neighbour row is one
//            coalesced 512-B wave load.
//   phase 2  wave w owns OUTPUT channels [32w, 32w+32): its 32x128 slice of W stays in
//            64 VGPRs for the lifetime of the (persistent) workgroup and is the MFMA
//            A operand; the LDS tile is the B operand.  D^T[32 ch x 32 rows] +=
//            W_slice[32 x 2] * A^T[2 x 32] with v_mfma_f32_32x32x2_f32 (exact fp32),
//            64 chained MFMAs per 32-row block.  The k index is permuted
//            (k = 64*(lane>>5) + s) so a lane's 64 A values are contiguous in LDS
//            (16 ds_read_b128 per row block) and its 64 W values contiguous in memory.
#pragma once
#include "common.h"

namespace eg {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

enum Agg { AGG_NONE = 0, AGG_CSR = 1, AGG_STENCIL = 2, AGG_CSRT = 3 };     // CSRT: CSR in clustered 64-node tiles (graph.hip csr_tiles)

// ---- wave-uniform helpers -------------------------------------------------------
__device__ inline int wave_id() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }

// ---- phase 1: one aggregated row (channels 2*lane, 2*lane+1) ---------------------
// 32-bit element offset from a wave-uniform base: lets the compiler use the SGPR-base + VGPR-offset
// addressing form (one VGPR per load instead of a 64-bit address pair).  rows * 128 < 2^31 is checked on the host.
__device__ inline f32x2 load_row2(const float* __restrict__ base, int row, int lane) {
    const unsigned off = (unsigned)row * (unsigned)C + 2u * (unsigned)lane;
    return *reinterpret_cast<const f32x2*>(base + off);
}

// out_i = d_i * sum_{j in N(i) u {i}} d_j x_j   — implicit topology, no index loads
__device__ inline f32x2 agg_stencil(const Topo* __restrict__ Tp, const float* __restrict__ xf, const float* __restrict__ dis,
                                    int n, int lane) {
    Nbrs nb;
    neighbours(*Tp, n, nb);
    float w[MAX_SLOTS];
#pragma unroll
    for (int s = 0; s < MAX_SLOTS; ++s) {
        const float d = dis[nb.id[s]];
        w[s] = nb.valid[s] ? d : 0.0f;
    }
    f32x2 v[MAX_SLOTS];
#pragma unroll
    for (int s = 0; s < 6; ++s) v[s] = load_row2(xf, nb.id[s], lane);
    f32x2 acc = {0.f, 0.f};
    if (nb.count > 6) {                     // aux-level node: 4 children as well (uniform branch)
#pragma unroll
        for (int s = 6; s < MAX_SLOTS; ++s) v[s] = load_row2(xf, nb.id[s], lane);
#pragma unroll
        for (int s = 6; s < MAX_SLOTS; ++s) acc += w[s] * v[s];
    }
#pragma unroll
    for (int s = 5; s >= 0; --s) acc += w[s] * v[s];
    return acc * w[0];
}

// ---- run-based implicit stencil -------------------------------------------------------
// A wave aggregates a run of consecutive node ids.  The level lookup and the (row, col)
// decode (one s_load_dwordx16 of the level descriptor, one magic division) happen once per
// run or level change; after that the position advances incrementally on the scalar unit.
// Loads of several nodes are issued together so one wave keeps ~20 row loads (512 B each)
// in flight.
#ifndef EG_MAIN_U
#define EG_MAIN_U 4      // main-grid nodes per burst of row loads (24 loads in flight per wave)
#endif

struct RunPos { int n, r, c; };

__device__ inline LevelDesc run_decode(const Topo* __restrict__ T, int n, RunPos& p) {
    const int nd = T->n_desc;
    int l = 0;
#pragma unroll 1
    for (int q = 1; q < nd; ++q) l += (n >= T->desc[q].base) ? 1 : 0;
    const LevelDesc d = T->desc[l];
    const int idx = n - d.base;
    const int r_main = (int)(((unsigned long long)(unsigned)idx * T->magic) >> 40);
    const int r_aux = idx >> d.lg;
    const int r = d.kind == KIND_MAIN ? r_main : (d.kind == KIND_AUX ? r_aux : 0);
    p.n = n; p.r = r; p.c = idx - r * d.side;
    return d;
}

__device__ inline void run_advance(const LevelDesc& d, RunPos& p) {
    ++p.n; ++p.c;
    const bool wrap = (p.c == d.side);
    p.c = wrap ? 0 : p.c;
    p.r = wrap ? p.r + 1 : p.r;
}

// slots: 0 self, 1 up, 2 down, 3 left, 4 right, 5 parent, 6..9 children.  Invalid slots point at the node itself.
template <int NS>
__device__ inline void run_slots(const LevelDesc& d, const RunPos& p, int (&id)[NS], int (&valid)[NS]) {
    const int n = p.n, r = p.r, c = p.c;
    const bool grid = d.kind != KIND_COORD;
    id[0] = n; valid[0] = 1;
    // coordinate pseudo-level: slots 1..4 are the four coordinate nodes (self masked), no parent / children
    valid[1] = grid ? (r > 0) : (d.base + 0 != n);            id[1] = grid ? (valid[1] ? n - d.side : n) : d.base + 0;
    valid[2] = grid ? (r < d.side - 1) : (d.base + 1 != n);   id[2] = grid ? (valid[2] ? n + d.side : n) : d.base + 1;
    valid[3] = grid ? (c > 0) : (d.base + 2 != n);            id[3] = grid ? (valid[3] ? n - 1 : n) : d.base + 2;
    valid[4] = grid ? (c < d.side - 1) : (d.base + 3 != n);   id[4] = grid ? (valid[4] ? n + 1 : n) : d.base + 3;
    valid[5] = (r < d.plim) && (c < d.plim);
    id[5] = valid[5] ? d.pbase + (d.poff + (r >> 1)) * d.pside + d.poff + (c >> 1) : n;
    if constexpr (NS > 6) {
        const int has = (r >= d.clo) && (r < d.chi) && (c >= d.clo) && (c < d.chi);
        const int b = has ? d.cbase + 2 * (r - d.clo) * d.cside + 2 * (c - d.clo) : n;
        const int cs = has ? d.cside : 0;
        id[6] = b; id[7] = b + has; id[8] = b + cs; id[9] = b + cs + has;
        valid[6] = valid[7] = valid[8] = valid[9] = has;
    }
}

// U consecutive nodes of one level, NS slots each; results to LDS rows rl .. rl+U-1
template <int U, int NS>
__device__ inline void run_group(const LevelDesc& d, RunPos& p, const float* __restrict__ xf,
                                 const float* __restrict__ dis, int lane, float* s_a, int rl) {
    float w[U][NS];
    f32x2 v[U][NS];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        int id[NS], valid[NS];
        run_slots<NS>(d, p, id, valid);
        run_advance(d, p);
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const float dd = dis[id[s]];
            w[u][s] = valid[s] ? dd : 0.0f;
#ifdef EG_ABL_NO_NBR          // timing-only ablation: neighbour rows are not loaded (results wrong)
            v[u][s] = s == 0 ? load_row2(xf, id[0], lane) : v[u][0];
#else
            v[u][s] = load_row2(xf, id[s], lane);
#endif
        }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        f32x2 acc = {0.f, 0.f};
#pragma unroll
        for (int s = NS - 1; s >= 0; --s) acc += w[u][s] * v[u][s];
        *reinterpret_cast<f32x2*>(&s_a[(rl + u) * LDA + 2 * lane]) = acc * w[u][0];
    }
}

// ---- lane-parallel stencil decode ------------------------------------------------------------
// The neighbourhood of several nodes is decoded at once on the VALU: lane L handles
// (node u = L / LPN, slot s = L % LPN) of a batch whose nodes all lie in one level (descriptor d
// is wave-uniform).  One vector gather fetches every normalisation weight of the batch; ids and
// weights are then broadcast with v_readlane as the row loads / FMAs are issued.  The scalar unit
// only walks levels.  slots: 0 self, 1 up, 2 down, 3 left, 4 right, 5 parent, 6..9 children.
__device__ inline void lane_slot(const LevelDesc& d, unsigned long long magic, int n, int s, int& id, int& valid) {
    const int idx = n - d.base;
    const int r_main = (int)(((unsigned long long)(unsigned)idx * magic) >> 40);
    const int r = d.kind == KIND_MAIN ? r_main : (d.kind == KIND_AUX ? (idx >> d.lg) : 0);
    const int c = idx - r * d.side;
    const bool par_ok = (r < d.plim) && (c < d.plim);
    const int par_id = d.pbase + (d.poff + (r >> 1)) * d.pside + d.poff + (c >> 1);
    const bool ch_ok = (r >= d.clo) && (r < d.chi) && (c >= d.clo) && (c < d.chi);
    const int ch_b = d.cbase + 2 * (r - d.clo) * d.cside + 2 * (c - d.clo);
    int cand = n;
    bool ok = (s == 0);
    cand = s == 1 ? n - d.side : cand;  ok = s == 1 ? (r > 0) : ok;
    cand = s == 2 ? n + d.side : cand;  ok = s == 2 ? (r < d.side - 1) : ok;
    cand = s == 3 ? n - 1 : cand;       ok = s == 3 ? (c > 0) : ok;
    cand = s == 4 ? n + 1 : cand;       ok = s == 4 ? (c < d.side - 1) : ok;
    cand = s == 5 ? par_id : cand;      ok = s == 5 ? par_ok : ok;
    const int q = s - 6;                                    // children 6..9
    const int ch_id = ch_b + (q & 1) + ((q >> 1) & 1) * d.cside;
    cand = (s >= 6 && s <= 9) ? ch_id : cand;
    ok = (s >= 6 && s <= 9) ? ch_ok : ok;
    if (d.kind == KIND_COORD) {                             // isolated K4: slots 1..4 = the four coordinate nodes
        cand = (s >= 1 && s <= 4) ? d.base + s - 1 : n;
        ok = (s == 0) || (s >= 1 && s <= 4 && cand != n);
    }
    valid = ok ? 1 : 0;
    id = ok ? cand : n;
}

__device__ inline float readlane_f(float v, int l) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}
// An SGPR zero the optimiser cannot see through.  Added to a lane index it ties a v_readlane to this program
// point: without it LLVM hoists all weight broadcasts of a batch to where the weights are produced and then
// spills dozens of SGPRs around the loads.
__device__ inline int opaque_zero() {
    int z = 0;
    asm volatile("" : "+s"(z));
    return z;
}

// ---- one grid-row segment of up to 8 consecutive nodes ("row-run stencil") -----------------------
// For 8 consecutive nodes of one grid row every neighbour class is a run of consecutive node rows:
//   S  the 8 nodes themselves (also each other's left / right neighbours), L / R the two edge rows,
//   U / D the 8 rows above / below, P the 4 parents (nodes 2k, 2k+1 share one),
//   C0 / C1 the 16 + 16 children (aux levels only).
// 30 row loads per 8 main-grid nodes (62 on an aux level) instead of 8 x 6 (8 x 10).  Run bases are
// scalar; weights come from the lane-parallel decode above (one vector gather per batch).
// Out-of-grid neighbours have weight 0 and a clamped (in-frame) address.
__device__ inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// Forces the accumulator updates of a stage to be complete at this program point and keeps later loads
// below it ("memory"): ISel otherwise sinks every FMA of a batch to the end, which keeps all 30-62 row
// loads of the batch live in registers at once.
__device__ inline void pin_acc(f32x2 (&acc)[8]) {
    asm volatile("" : "+v"(acc[0].x), "+v"(acc[0].y), "+v"(acc[1].x), "+v"(acc[1].y), "+v"(acc[2].x), "+v"(acc[2].y),
                      "+v"(acc[3].x), "+v"(acc[3].y), "+v"(acc[4].x), "+v"(acc[4].y), "+v"(acc[5].x), "+v"(acc[5].y),
                      "+v"(acc[6].x), "+v"(acc[6].y), "+v"(acc[7].x), "+v"(acc[7].y) :: "memory");
}

__device__ inline const float* run_ptr(const float* __restrict__ xf, int row0, int lane) {
    return xf + ((unsigned)row0 * (unsigned)C + 2u * (unsigned)lane);     // rows row0+k are at imm offsets k*512 B
}
__device__ inline f32x2 ld2(const float* p, int k) { return *reinterpret_cast<const f32x2*>(p + k * C); }

// ---- paired-row layout ---------------------------------------------------------------------------
// One wave access moves TWO consecutive node rows (1 KB): lane l -> row parity h = l >> 5, channels
// 4q .. 4q+3 with q = l & 31 (16 B per lane).  The L1/TA pipeline handles 16-B lanes at twice the byte rate
// of 8-B lanes, and a segment of 8 nodes needs 18 such loads instead of 30 (46 instead of 62 on aux levels).
// Node u of the segment lives in half h = u & 1 of pair k = u >> 1.
struct PairLane { int h, q; };

// 32-bit element offsets from the wave-uniform frame base (SGPR base + one VGPR offset + immediate)
__device__ inline unsigned pair_off(int row0, const PairLane& pl) {
    return (unsigned)row0 * (unsigned)C + (unsigned)(pl.h * C + 4 * pl.q);              // rows row0+2k+h at imm offsets k*1 KB
}
__device__ inline unsigned bcast_off(int row, const PairLane& pl) {
    return (unsigned)row * (unsigned)C + (unsigned)(4 * pl.q);                          // the same row in both halves
}
__device__ inline f32x4 ld4(const float* __restrict__ xf, unsigned off, int k) {
    return *reinterpret_cast<const f32x4*>(xf + (off + (unsigned)(k * 2 * C)));
}
// [a.upper | b.lower]: the node before / after a pair boundary, in the half that needs it
__device__ inline f32x4 seam(const f32x4& a, const f32x4& b, bool upper_half) {
    f32x4 m;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a[i]), __float_as_uint(b[i]), false, false);
        // r[0] = [a.lo | b.lo], r[1] = [a.up | b.up]
        m[i] = upper_half ? __uint_as_float(r[0]) : __uint_as_float(r[1]);
    }
    return m;
}

// lower / upper half of v broadcast to both halves (one v_permlane32_swap per register, no LDS)
__device__ inline void halves(const f32x4& v, f32x4& lo, f32x4& up) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const unsigned x = __float_as_uint(v[i]);
        const auto r = __builtin_amdgcn_permlane32_swap(x, x, false, false);
        lo[i] = __uint_as_float(r[0]);
        up[i] = __uint_as_float(r[1]);
    }
}
__device__ inline f32x4 sel(bool upper_half, const f32x4& if_lower, const f32x4& if_upper) {
    return upper_half ? if_upper : if_lower;
}
// weight of (node 2k + h, slot) for this lane: lane index (2k+h)*8 + slot of the decode vector
__device__ inline float pair_w(float w, int hb4, int k, int slot) {
    return __int_as_float(__builtin_amdgcn_ds_bpermute(hb4 + ((2 * k) * 8 + slot) * 4, __float_as_int(w)));
}
__device__ inline void pin_acc4(f32x4 (&acc)[4]) {
    asm volatile("" : "+v"(acc[0].x), "+v"(acc[0].y), "+v"(acc[0].z), "+v"(acc[0].w), "+v"(acc[1].x), "+v"(acc[1].y),
                      "+v"(acc[1].z), "+v"(acc[1].w), "+v"(acc[2].x), "+v"(acc[2].y), "+v"(acc[2].z), "+v"(acc[2].w),
                      "+v"(acc[3].x), "+v"(acc[3].y), "+v"(acc[3].z), "+v"(acc[3].w) :: "memory");
}

template <bool AUX>
__device__ inline void stencil_row8(const Topo* __restrict__ T, const LevelDesc& d, unsigned long long magic,
                                    const float* __restrict__ xf, const float* __restrict__ dis, int n_first, int cnt,
                                    int n_frame, int lane, float* s_a, float* s_x, int rl) {
    // run bases from the position of the first node (scalar)
    const int idx = n_first - d.base;
    const int r_main = (int)(((unsigned long long)(unsigned)idx * magic) >> 40);
    const int r = d.kind == KIND_MAIN ? r_main : (idx >> d.lg);
    const int c0 = idx - r * d.side;
    const int cb = d.cbase + 2 * (r - d.clo) * d.cside + 2 * (c0 - d.clo);
    const bool kids = AUX && r >= d.clo && r < d.chi;
    // Rare per-node scalar path: coordinate K4, and segments so close to the end of the frame that a run of 8
    // rows (self / below / children) would have to be clamped while some of its rows are real neighbours.
    if (d.kind == KIND_COORD || n_first + d.side + 8 > n_frame || (kids && cb + d.cside + 16 > n_frame)) {
        for (int u = 0; u < cnt; ++u) {
            *reinterpret_cast<f32x2*>(&s_a[(rl + u) * LDA + 2 * lane]) = agg_stencil(T, xf, dis, n_first + u, lane);
            if (s_x) *reinterpret_cast<f32x2*>(&s_x[(rl + u) * C + 2 * lane]) = load_row2(xf, n_first + u, lane);
        }
        return;
    }
    // weights: lane (u = lane>>3, s = lane&7) -> slot s of node u; aux levels also need slots 8, 9
    int u_l = lane >> 3, s_l = lane & 7;
    // opaque to the optimiser: otherwise the dozen (slot == k) lane masks of lane_slot are hoisted out of the
    // tile loop as loop invariants and pin ~30 SGPRs (2 per mask) for the whole kernel
    asm volatile("" : "+v"(u_l), "+v"(s_l));
    const int n_l = n_first + (u_l < cnt ? u_l : cnt - 1);
    int id, valid;
    // raw gathered weights; they are masked (valid ? w : 0) only after the row loads below have been issued,
    // otherwise the wave would sit out the gather's round trip before issuing them
    lane_slot(d, magic, n_l, s_l, id, valid);
    const float dwa = dis[id];
    const int va = valid;
    float dwb = 0.0f;
    int vb = 0;
    if constexpr (AUX) {
        lane_slot(d, magic, n_l, 8 + (s_l & 1), id, valid);
        dwb = dis[id];
        vb = valid;
    }
    const PairLane pl{lane >> 5, lane & 31};
    const bool up_half = pl.h != 0;
    const int hb4 = pl.h * 32;                                   // byte offset of "+1 node" in the decode vector (8 lanes * 4 B)
    const int hi8 = n_frame - 8, last = n_frame - 1;
    const unsigned os = pair_off(n_first, pl);
    const unsigned ou = pair_off(clampi(n_first - d.side, 0, hi8), pl);
    const unsigned od = pair_off(clampi(n_first + d.side, 0, hi8), pl);
    const unsigned op = pair_off(clampi(d.pbase + (d.poff + (r >> 1)) * d.pside + d.poff + (c0 >> 1), 0, hi8), pl);
    const unsigned oL = bcast_off(clampi(n_first - 1, 0, last), pl);
    const unsigned oR = bcast_off(clampi(n_first + 8, 0, last), pl);

    // stage A loads: self pairs and the two edge rows (6 wave loads = 5 KB in flight)
    f32x4 S[4], Le, Re;
#pragma unroll
    for (int k = 0; k < 4; ++k) S[k] = ld4(xf, os, k);
    Le = *reinterpret_cast<const f32x4*>(xf + oL);
    Re = *reinterpret_cast<const f32x4*>(xf + oR);
    __builtin_amdgcn_sched_barrier(0);
    const float wa = va ? dwa : 0.0f;
    const float wb = vb ? dwb : 0.0f;

    f32x4 acc[4];
    {   // self + left + right.  M[j] = [node 2j-1 | node 2j] is the left neighbour vector of pair j and the right
        // neighbour vector of pair j-1 (edge rows Le / Re are already in both halves)
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
            for (int k = 0; k < 4; ++k) *reinterpret_cast<f32x4*>(&s_x[(rl + 2 * k + pl.h) * C + 4 * pl.q]) = S[k];
        }
    }
    pin_acc4(acc);
    // stage B loads: rows above, rows below and the two parent pairs (10 wave loads = 10 KB in flight)
    f32x4 U[4], D[4], P[2];
#pragma unroll
    for (int k = 0; k < 4; ++k) U[k] = ld4(xf, ou, k);
#pragma unroll
    for (int k = 0; k < 4; ++k) D[k] = ld4(xf, od, k);
    P[0] = ld4(xf, op, 0);
    P[1] = ld4(xf, op, 1);
    pin_acc4(acc);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[k] += pair_w(wa, hb4, k, 1) * U[k];
    pin_acc4(acc);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[k] += pair_w(wa, hb4, k, 2) * D[k];
    pin_acc4(acc);
    __builtin_amdgcn_sched_barrier(0);
    {   // parents: nodes 2k and 2k+1 share parent k = half (k & 1) of parent pair (k >> 1)
        f32x4 plo[2], pup[2];
        halves(P[0], plo[0], pup[0]);
        halves(P[1], plo[1], pup[1]);
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[k] += pair_w(wa, hb4, k, 5) * ((k & 1) ? pup[k >> 1] : plo[k >> 1]);
    }
    pin_acc4(acc);
    if constexpr (AUX) {
        // children of node u: the two halves of child pair u in child row 2r (slots 6, 7) and in child row 2r+1
        // (slots 8, 9).  Weighted lane-wise, then the two halves are added and routed to node u's half of acc[u>>1].
        __builtin_amdgcn_sched_barrier(0);
        const unsigned oc0 = pair_off(clampi(cb, 0, hi8), pl);
        const unsigned oc1 = pair_off(clampi(cb + 8, 0, hi8), pl);
        const unsigned oc2 = pair_off(clampi(cb + d.cside, 0, hi8), pl);
        const unsigned oc3 = pair_off(clampi(cb + d.cside + 8, 0, hi8), pl);
        const int hs4 = pl.h * 4;                                  // slot 6 -> 7 / 8 -> 9 for the upper half
#pragma unroll
        for (int half = 0; half < 2; ++half) {                     // nodes 4*half .. 4*half+3
            f32x4 Ca[4], Cb[4];
            const unsigned qa = half ? oc1 : oc0;
            const unsigned qb = half ? oc3 : oc2;
#pragma unroll
            for (int k = 0; k < 4; ++k) { Ca[k] = ld4(xf, qa, k); Cb[k] = ld4(xf, qb, k); }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                f32x4 red[2];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int u = 4 * half + 2 * kk + e;
                    const float w67 = __int_as_float(__builtin_amdgcn_ds_bpermute(hs4 + (u * 8 + 6) * 4, __float_as_int(wa)));
                    const float w89 = __int_as_float(__builtin_amdgcn_ds_bpermute(hs4 + (u * 8 + 0) * 4, __float_as_int(wb)));
                    const f32x4 t = w67 * Ca[2 * kk + e] + w89 * Cb[2 * kk + e];
                    f32x4 tl, tu;
                    halves(t, tl, tu);
                    red[e] = tl + tu;
                }
                acc[2 * half + kk] += sel(up_half, red[0], red[1]);
            }
            pin_acc4(acc);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const f32x4 o = acc[k] * pair_w(wa, hb4, k, 0);
        *reinterpret_cast<f32x4*>(&s_a[(rl + 2 * k + pl.h) * LDA + 4 * pl.q]) = o;       // rows >= cnt are padding rows of the tile
    }
}

// patch rows tr0 .. tr0+ntr-1 of patch td -> LDS rows 8*tr .. 8*tr+7
__device__ inline void stencil_patch_rows(const Topo* __restrict__ T, const LevelDesc& d, const TileDesc& td, unsigned long long magic,
                                          const float* __restrict__ xf, const float* __restrict__ dis, int n_frame,
                                          int tr0, int ntr, int lane, float* s_a, float* s_x) {
#pragma unroll 1
    for (int tr = tr0; tr < tr0 + ntr; ++tr) {
        if (tr >= td.nrows) break;
        const int n_first = d.base + (td.r0 + tr) * d.side + td.c0;
        if (d.kind == KIND_AUX) stencil_row8<true>(T, d, magic, xf, dis, n_first, td.ncols, n_frame, lane, s_a, s_x, tr * 8);
        else stencil_row8<false>(T, d, magic, xf, dis, n_first, td.ncols, n_frame, lane, s_a, s_x, tr * 8);
    }
}

// generic CSR (by target).  rowptr/colidx/dis are wave-uniform reads.
__device__ inline f32x2 agg_csr(const float* __restrict__ xf, const float* __restrict__ dis,
                                const int* __restrict__ rowptr, const int* __restrict__ colidx, int n, int lane) {
    const int e0 = rowptr[n], e1 = rowptr[n + 1];
    const float dn = dis[n];
    f32x2 acc = {0.f, 0.f};
    for (int e = e0; e < e1; e += 4) {
        int j[4];
        float w[4];
        f32x2 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int ee = (e + k < e1) ? e + k : e1 - 1;
            j[k] = colidx[ee];
            w[k] = (e + k < e1) ? dis[j[k]] : 0.0f;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = load_row2(xf, j[k], lane);
#pragma unroll
        for (int k = 0; k < 4; ++k) acc += w[k] * v[k];
    }
    acc += dn * load_row2(xf, n, lane);      // the implied self loop comes last, as in gcn_norm
    return acc * dn;
}

// ---- phase 2: W slice in registers + MFMA over one 32-row block -------------------
// lane (i = l&31, h = l>>5) of wave w holds W[32w+i][64h+s] (or W[64h+s][32w+i] when transposed), s = 0..63
__device__ inline void load_w_slice(const float* __restrict__ W, int wave, int lane, int transpose, float (&wreg)[64]) {
    const int i = lane & 31, h = lane >> 5;
    if (!transpose) {
        const f32x4* p = reinterpret_cast<const f32x4*>(W + (size_t)(32 * wave + i) * C + 64 * h);
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const f32x4 q = p[t];
            wreg[4 * t + 0] = q.x; wreg[4 * t + 1] = q.y; wreg[4 * t + 2] = q.z; wreg[4 * t + 3] = q.w;
        }
    } else {
        const float* p = W + (size_t)(64 * h) * C + 32 * wave + i;
#pragma unroll
        for (int s = 0; s < 64; ++s) wreg[s] = p[(size_t)s * C];
    }
}

// acc[m][n]: n = lane&31 = row (row0 + n), m = (reg&3) + 8*(reg>>2) + 4*(lane>>5) = channel within the wave's 32.
// The 16 ds_read_b128 are software-pipelined in chunks of 4 (two named fragment sets) so that at most
// 32 VGPRs hold A fragments while the 64-cycle MFMAs of the previous chunk cover the LDS latency.
// EG_YIELD (experiment): a vector-class instruction of the MFMA wave's own between two MFMAs is what hands the
// SIMD's other wave its issue slots (tools/micro_coissue5.hip: v_nop behind an MFMA: +16 cycles for the chain, ~10 instructions
// for the partner instead of ~1).  EG_YIELD = n puts a v_nop behind every n-th MFMA of a chunk (n in 1, 2, 4).
#ifdef EG_YIELD
#define EG_YIELD_POINT(i) do { if ((i) % EG_YIELD == EG_YIELD - 1) { __builtin_amdgcn_sched_barrier(0); asm volatile("v_nop"); __builtin_amdgcn_sched_barrier(0); } } while (0)
#else
#define EG_YIELD_POINT(i) do {} while (0)
#endif
template <bool YIELD = false>
__device__ inline void mfma_chunk(const f32x4 (&av)[4], const float (&wreg)[64], int t0, f32x16& acc) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wreg[4 * (t0 + t) + 0], av[t].x, acc, 0, 0, 0);
        if (YIELD) EG_YIELD_POINT(0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wreg[4 * (t0 + t) + 1], av[t].y, acc, 0, 0, 0);
        if (YIELD) EG_YIELD_POINT(1);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wreg[4 * (t0 + t) + 2], av[t].z, acc, 0, 0, 0);
        if (YIELD) EG_YIELD_POINT(2);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wreg[4 * (t0 + t) + 3], av[t].w, acc, 0, 0, 0);
        if (YIELD) EG_YIELD_POINT(3);
    }
}

__device__ inline void mfma_rowblock(const float* s_a, int row0, int lane, const float (&wreg)[64], f32x16& acc) {
    const int j = lane & 31, h = lane >> 5;
    const f32x4* ap = reinterpret_cast<const f32x4*>(s_a + (row0 + j) * LDA + 64 * h);
    f32x4 a0[4], a1[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) a0[t] = ap[t];
#pragma unroll
    for (int t = 0; t < 4; ++t) a1[t] = ap[4 + t];
    __builtin_amdgcn_sched_barrier(0);
    mfma_chunk<true>(a0, wreg, 0, acc);
#pragma unroll
    for (int t = 0; t < 4; ++t) a0[t] = ap[8 + t];
    __builtin_amdgcn_sched_barrier(0);
    mfma_chunk<true>(a1, wreg, 4, acc);
#pragma unroll
    for (int t = 0; t < 4; ++t) a1[t] = ap[12 + t];
    __builtin_amdgcn_sched_barrier(0);
    mfma_chunk<true>(a0, wreg, 8, acc);
    __builtin_amdgcn_sched_barrier(0);
    mfma_chunk<true>(a1, wreg, 12, acc);
}

// Same chain; between(c) is emitted in the scheduling region of chunk c (c = 0..3) so that independent VALU / LDS /
// store work of the caller fills the issue slots the dependent 64-cycle MFMAs leave free.
// EG_CLUMP (experiment): keep each between(c) piece in ONE place behind its chunk of the chain instead of letting the
// scheduler spread it between the MFMAs -- an interruption of a chain costs 16 - 24 cycles however short it is (micro_coissue5)
#ifdef EG_CLUMP
#define EG_CLUMP_FENCE do { asm volatile("" : "+v"(acc)); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define EG_CLUMP_FENCE do {} while (0)
#endif
template <typename F>
__device__ inline void mfma_rowblock_with(const float* s_a, int row0, int lane, const float (&wreg)[64], f32x16& acc, F between) {
    const int j = lane & 31, h = lane >> 5;
    const f32x4* ap = reinterpret_cast<const f32x4*>(s_a + (row0 + j) * LDA + 64 * h);
    f32x4 a0[4], a1[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) a0[t] = ap[t];
#pragma unroll
    for (int t = 0; t < 4; ++t) a1[t] = ap[4 + t];
    __builtin_amdgcn_sched_barrier(0);
    mfma_chunk(a0, wreg, 0, acc);
    EG_CLUMP_FENCE; between(0); EG_CLUMP_FENCE;
#pragma unroll
    for (int t = 0; t < 4; ++t) a0[t] = ap[8 + t];
    __builtin_amdgcn_sched_barrier(0);
    mfma_chunk(a1, wreg, 4, acc);
    EG_CLUMP_FENCE; between(1); EG_CLUMP_FENCE;
#pragma unroll
    for (int t = 0; t < 4; ++t) a1[t] = ap[12 + t];
    __builtin_amdgcn_sched_barrier(0);
    mfma_chunk(a0, wreg, 8, acc);
    EG_CLUMP_FENCE; between(2); EG_CLUMP_FENCE;
    __builtin_amdgcn_sched_barrier(0);
    mfma_chunk(a1, wreg, 12, acc);
    EG_CLUMP_FENCE; between(3); EG_CLUMP_FENCE;
}

// ---- 8-wave layout: wave w owns 16 output channels, v_mfma_f32_16x16x4_f32 ------------------------
// A operand = W slice (32 VGPRs for the lifetime of the workgroup): lane (i = l&15, kq = l>>4) holds
// W[16w+i][koff(kq) + s], s = 0..31.  B operand = LDS tile: lane (j = l&15, kq) reads a[row0+j][koff(kq)+s]
// as 8 ds_read_b128.  koff = {0, 64, 32, 96}: the two kq values that share a ds_read_b128 lane group
// are 64 floats apart, which keeps the 16 lanes of a group on disjoint banks at LDS row stride 132.
// D: lane (j, q = l>>4), reg i  ->  out[row0 + j][16w + 4q + i].
typedef float f32x4v __attribute__((ext_vector_type(4)));

__device__ inline int koff16(int kq) { return ((kq & 1) << 6) | ((kq & 2) << 4); }

__device__ inline void load_w_slice16(const float* __restrict__ W, int wave, int lane, int transpose, float (&wreg)[32]) {
    const int i = lane & 15, k0 = koff16(lane >> 4);
    if (!transpose) {
        const f32x4* p = reinterpret_cast<const f32x4*>(W + (size_t)(16 * wave + i) * C + k0);
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const f32x4 q = p[t];
            wreg[4 * t + 0] = q.x; wreg[4 * t + 1] = q.y; wreg[4 * t + 2] = q.z; wreg[4 * t + 3] = q.w;
        }
    } else {
        const float* p = W + (size_t)k0 * C + 16 * wave + i;
#pragma unroll
        for (int s = 0; s < 32; ++s) wreg[s] = p[(size_t)s * C];
    }
}

// two 16-row blocks (row0, row0+16) with independent accumulators: the 40-cycle dependent latency of one
// chain is covered by the other chain's 32-cycle issue slot.
template <bool PREFETCH = true>
__device__ inline void mfma16_pair(const float* s_a, int row0, int lane, const float (&wreg)[32], f32x4v& accA, f32x4v& accB) {
    const int j = lane & 15, k0 = koff16(lane >> 4);
    const f32x4* pa = reinterpret_cast<const f32x4*>(s_a + (row0 + j) * LDA + k0);
    const f32x4* pb = reinterpret_cast<const f32x4*>(s_a + (row0 + 16 + j) * LDA + k0);
    f32x4 fa[4], fb[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) { fa[t] = pa[t]; fb[t] = pb[t]; }
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        f32x4 ga[4], gb[4];
        if (PREFETCH && c == 0) {
#pragma unroll
            for (int t = 0; t < 4; ++t) { ga[t] = pa[4 + t]; gb[t] = pb[4 + t]; }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int w0 = 16 * c + 4 * t;
            accA = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[w0 + 0], fa[t].x, accA, 0, 0, 0);
            accB = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[w0 + 0], fb[t].x, accB, 0, 0, 0);
            accA = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[w0 + 1], fa[t].y, accA, 0, 0, 0);
            accB = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[w0 + 1], fb[t].y, accB, 0, 0, 0);
            accA = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[w0 + 2], fa[t].z, accA, 0, 0, 0);
            accB = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[w0 + 2], fb[t].z, accB, 0, 0, 0);
            accA = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[w0 + 3], fa[t].w, accA, 0, 0, 0);
            accB = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[w0 + 3], fb[t].w, accB, 0, 0, 0);
        }
        if (c == 0) {
            if (!PREFETCH) {                      // register-lean form: second half is read after the first is consumed
#pragma unroll
                for (int t = 0; t < 4; ++t) { ga[t] = pa[4 + t]; gb[t] = pb[4 + t]; }
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) { fa[t] = ga[t]; fb[t] = gb[t]; }
        }
    }
}

// ---- persistent tile walk ------------------------------------------------------------------
// Tiles that are neighbours in the node order share halo rows (grid rows +-1, parents), so they
// should be processed on the SAME XCD (same L2) at about the same time.  Three modes:
//   WALK_QUEUE   (default) each XCD owns one contiguous chunk of tiles; a workgroup reads the XCD it
//                really runs on (HW_REG_XCC_ID) and claims the next tile of that chunk from a
//                per-XCD device counter (stealing from other chunks once its own is drained).
//                Placement-independent for correctness, load-balanced, no assumption on dispatch order.
//   WALK_MOD8    static: blockIdx % 8 labels the chunk (round-robin dispatch heuristic)
//   WALK_STRIDE  static: tile = blockIdx + k * gridDim
enum { WALK_QUEUE = 0, WALK_MOD8 = 1, WALK_STRIDE = 2 };
constexpr int WALK_GROUPS = 8;
constexpr int WALK_CTR_STRIDE = 32;     // ints between counters: one 128-B line each

__device__ inline int xcc_id() {
    // s_getreg_b32 hwreg(HW_REG_XCC_ID, 0, 4): id 20, offset 0, size 4
    return __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 7;
}

struct TileWalk {
    int mode, n_tiles, chunk;
    int base, local, stride;        // static modes
    int group;                      // queue mode: home chunk
    int* counters;
    int* s_slot;                    // one int of LDS, written by thread 0

    __device__ TileWalk(int mode_, int n_tiles_, int* counters_, int* s_slot_)
        : mode(mode_), n_tiles(n_tiles_), counters(counters_), s_slot(s_slot_) {
        chunk = (n_tiles + WALK_GROUPS - 1) / WALK_GROUPS;
        if (mode == WALK_MOD8) {
            const int g = blockIdx.x % WALK_GROUPS;
            stride = (gridDim.x + WALK_GROUPS - 1 - g) / WALK_GROUPS;
            base = g * chunk; local = blockIdx.x / WALK_GROUPS;
        } else if (mode == WALK_STRIDE) {
            stride = gridDim.x; base = 0; local = blockIdx.x; chunk = n_tiles;
        } else {
            group = xcc_id(); base = local = stride = 0;
        }
    }

    // Split form: claim_issue() starts thread 0's atomic on its home queue, claim_commit() (placed after this
    // wave's phase-1 work, so the atomic's round trip is hidden) turns the ticket into a tile id — stealing from
    // the other queues once the home chunk is drained — and publishes it in LDS; fetch() must be separated from
    // the commit by a workgroup barrier.
    int pend;
    __device__ void claim_issue() {
        pend = 0;
        if (mode == WALK_QUEUE && threadIdx.x == 0) pend = atomicAdd(&counters[group * WALK_CTR_STRIDE], 1);
    }
    __device__ void claim_commit() {
        if (mode != WALK_QUEUE) {
            if (threadIdx.x == 0) *s_slot = (local < chunk && base + local < n_tiles) ? base + local : -1;
            local += stride;
            return;
        }
        if (threadIdx.x == 0) {
            const int lo0 = group * chunk;
            const int size0 = (n_tiles - lo0) < chunk ? (n_tiles - lo0) : chunk;
            int t = pend < size0 ? lo0 + pend : -1;
            for (int k = 1; k < WALK_GROUPS && t < 0; ++k) {
                const int q = (group + k) % WALK_GROUPS;
                const int lo = q * chunk;
                const int size = (n_tiles - lo) < chunk ? (n_tiles - lo) : chunk;
                if (size <= 0) continue;
                const int got = atomicAdd(&counters[q * WALK_CTR_STRIDE], 1);
                if (got < size) t = lo + got;
            }
            *s_slot = t;
        }
    }
    __device__ void claim() { claim_issue(); claim_commit(); }
    __device__ int fetch() const { return __builtin_amdgcn_readfirstlane(*s_slot); }

    // All threads of the workgroup must call this together (it contains a barrier in queue mode).
    __device__ bool next(int& tile) {
        if (mode != WALK_QUEUE) {
            if (local >= chunk) return false;
            tile = base + local;
            local += stride;
            return tile < n_tiles;
        }
        if (threadIdx.x == 0) {
            int t = -1;
            for (int k = 0; k < WALK_GROUPS && t < 0; ++k) {
                const int q = (group + k) % WALK_GROUPS;
                const int lo = q * chunk;
                const int size = (n_tiles - lo) < chunk ? (n_tiles - lo) : chunk;
                if (size <= 0) continue;
                const int got = atomicAdd(&counters[q * WALK_CTR_STRIDE], 1);
                if (got < size) t = lo + got;
            }
            *s_slot = t;
        }
        __syncthreads();
        tile = *s_slot;
        tile = __builtin_amdgcn_readfirstlane(tile);
        return tile >= 0;
    }
};

}  // namespace eg
