// Fused GCN layer for gfx950:  aggregate (pull, atomic-free) -> 128x128 node update on
// fp32 MFMA -> bias/BatchNorm(eval)/ReLU/residual epilogue, one kernel.
// Replaces torch_geometric GCNConv + BatchNorm1d + ReLU + residual as used at
// reference src/core/models.py:328-335, :431-435.  A_hat (X W^T) == (A_hat X) W^T, so the
// aggregation runs first on the layer input and its result never leaves LDS.
#include <stdlib.h>

#include "tile.h"

namespace eg {

// Read-only pointers are separate __restrict__ kernel parameters (not struct members) so the
// compiler can prove the wave-uniform table reads (dis / rowptr / colidx / Topo) are never
// clobbered by the kernel's stores and emit them as scalar loads.
struct GraphPtrs {
    const float* dis;
    const int* rowptr;
    const int* colidx;
    const Topo* topo;
    const TileDesc* tiles;
    const int *t_rows, *t_rowptr, *t_code, *t_tgt;      // AGG_CSRT: the CSR in clustered tiles (common.h eg_graph::t_*)
    const float *t_w, *t_dis;
};

struct LayerDims {
    int n_per_frame;       // rows per frame
    int batch;
    int tiles_per_frame;
    int relu;
    int transpose_w;
    int walk_mode;
    int stagger;           // experiment knob: the second half of the grid starts this many s_sleep(127) late
    int self_reset;        // queue walk: the last workgroup out zeroes the launch's slice of the queue ring (common.h: device invariant)
};

struct LayerArgs {         // host-side bundle only
    const float* x;
    const float* W;
    const float* scale;
    const float* shift;
    const float* residual;
    float* out;
    float* agg_out;        // nullable: the aggregated rows A_hat x (what the weight gradient of a train-mode layer needs)
    float* stats_partial;  // nullable (train forward): per-workgroup column sums of out and out^2, [grid][2][128]
    GraphPtrs gp;
    LayerDims d;
    int* walk_counters;    // this launch's slice of the handle's queue ring (NULL: static walk)
    const eg_graph* graph; // the handle whose ring the slice comes from (NULL: handle-less entry points)
    Knobs knobs;
};

// Diagnostic build only (-DEG_STAMP): per-wave cycle sums per phase, added to a stats area behind the
// queue counters.  Never enabled in the shipped library; its run time is not representative.
#ifdef EG_STAMP
#define STAMP(i) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long _t = __builtin_amdgcn_s_memtime(); \
                      __builtin_amdgcn_s_waitcnt(0xC07F); st[i] += _t - t_prev; t_prev = _t; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define STAMP(i) do {} while (0)
#endif

template <int AGG>
__device__ inline f32x2 produce_row(const float* __restrict__ dis, const int* __restrict__ rowptr,
                                    const int* __restrict__ colidx, const Topo* __restrict__ T,
                                    const float* __restrict__ xf, int n, int lane) {
    if constexpr (AGG == AGG_STENCIL) return agg_stencil(T, xf, dis, n, lane);
    else if constexpr (AGG == AGG_CSR) return agg_csr(xf, dis, rowptr, colidx, n, lane);
    else return load_row2(xf, n, lane);
}

// One workgroup = 8 waves walks 64-row tiles (persistent, XCD-aware dynamic queue).  A tile is an
// 8x8 patch of one level (implicit topology: neighbour reuse stays inside the workgroup / its L2) or
// 64 consecutive rows (CSR / plain linear).  Wave w owns LDS rows 8w..8w+7 (one patch row) in the
// memory phases and output channels 16w..16w+15 in the MFMA phase.  Per tile:
//   phase 1  each wave aggregates its 8 rows (natural layout: one row = one 512-B wave access) -> LDS
//   phase 2  each wave: its 16 output channels x 64 rows on the fp32 MFMA (W slice in 32 registers)
//   phase 3  accumulators go back through the same LDS tile so the epilogue runs in the natural
//            layout again: per-lane scale/shift, ReLU, residual, and full-row (512 B) coalesced
//            residual loads / output stores.  A wave touches only its own 8 LDS rows in phases 3 and 1,
//            so no barrier separates a tile's phase 3 from the next tile's phase 1.
constexpr int LAYER_THREADS = 512;

// RES_GLOBAL: the residual rows are fetched from global memory in the epilogue (any pointer).  The hot
// instantiation <AGG_STENCIL, false> instead reuses the self rows that phase 1 stashed in LDS (residual == x)
// or has no residual at all, and carries no registers for the residual.
// WAGG = train-mode forward (its own instantiation so that the inference instantiations keep their register allocation):
// the aggregated rows are also written to agg_out (when given) and the column sums of the output and its square —
// the BatchNorm batch statistics — are accumulated in the epilogue (stats_partial, when given), which saves the separate
// statistics pass over z.
template <int AGG, bool RES_GLOBAL, bool WAGG = false>
__global__ __launch_bounds__(LAYER_THREADS, 4) void k_gcn_layer(const float* __restrict__ x, const float* __restrict__ W,
                                                   const float* __restrict__ scale, const float* __restrict__ shift,
                                                   const float* __restrict__ residual, float* __restrict__ out,
                                                   float* __restrict__ agg_out, float* __restrict__ stats_partial,
                                                   const float* __restrict__ dis, const int* __restrict__ rowptr,
                                                   const int* __restrict__ colidx, const Topo* __restrict__ T,
                                                   const TileDesc* __restrict__ tiles,
                                                   const int* __restrict__ t_rows, const int* __restrict__ t_rowptr,
                                                   const int* __restrict__ t_code, const int* __restrict__ t_tgt,
                                                   const float* __restrict__ t_w, const float* __restrict__ t_dis,
                                                   int* __restrict__ walk_counters, const LayerDims a) {
    __shared__ __attribute__((aligned(16))) float s_a[TILE * LDA + 4];
    // raw self rows of the tile (implicit-topology path with residual == x): the epilogue's residual source
    // (AGG_CSRT: the raw rows of the tile's 64 nodes, whatever the residual is: most sources of a tile's edges are its own rows)
    __shared__ __attribute__((aligned(16))) float s_xbuf[((AGG == AGG_STENCIL && !RES_GLOBAL) || AGG == AGG_CSRT) ? TILE * C : 4];
    float* const s_x = (AGG == AGG_STENCIL && !RES_GLOBAL && residual) ? s_xbuf : nullptr;
    __shared__ int s_ids[AGG == AGG_CSRT ? TILE : 1];            // AGG_CSRT: node id of every tile slot (-1: padding)

    const int tid = threadIdx.x;
    const int lane_k = tid & 63;
    const int wave = wave_id();

    float wreg[32];
    load_w_slice16(W, wave, lane_k, a.transpose_w, wreg);
    // epilogue constants live in LDS (not in 8 VGPRs held across the whole tile loop): read per tile in phase 3
    __shared__ __attribute__((aligned(16))) float s_scale[C];
    __shared__ __attribute__((aligned(16))) float s_shift[C];
    if (tid < C) {
        s_scale[tid] = scale ? scale[tid] : 1.0f;
        s_shift[tid] = shift ? shift[tid] : 0.0f;
    }

    // train forward: per-thread column sums live in LDS (lane (h, q): channels 4q..4q+3 -> [sum x4 | sum of squares x4])
    __shared__ __attribute__((aligned(16))) float s_stat[WAGG ? LAYER_THREADS * 8 : 4];
    if constexpr (WAGG) {
        *reinterpret_cast<f32x4*>(&s_stat[tid * 8]) = f32x4{0.f, 0.f, 0.f, 0.f};
        *reinterpret_cast<f32x4*>(&s_stat[tid * 8 + 4]) = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    TileWalk walk(a.walk_mode, a.tiles_per_frame * a.batch, walk_counters, reinterpret_cast<int*>(&s_a[TILE * LDA]));
#ifdef EG_STAMP
    unsigned long long st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long t_prev = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
#endif
    if (a.stagger > 0 && blockIdx.x >= gridDim.x / 2)
        for (int i = 0; i < a.stagger; ++i) __builtin_amdgcn_s_sleep(127);
    // the next tile is claimed by thread 0 at the top of an iteration and read back after barrier 1, so the
    // queue's atomic round trip never sits on the critical path
    walk.claim();
    __syncthreads();
    int tile = walk.fetch();
    __syncthreads();                       // everyone has read the slot before thread 0 overwrites it
    // AGG_CSRT: the edge words of the wave's 8 slots (source code, weight, target slot: lane l holds edge l of the list) are
    // fetched ONE TILE AHEAD -- right after the next tile is known, under this tile's matrix phase -- so that a tile's outside
    // sources can be requested at its very top, together with its own rows (one memory round trip per tile, like the stencil)
    int m_code = 0, m_tg = 0, m_elo = 0, m_cnt = 0;
    float m_w = 0.f;
    auto load_meta = [&](int tl) {
        if constexpr (AGG == AGG_CSRT) {
            const int fr = tl / a.tiles_per_frame;
            const int slot0 = (tl - fr * a.tiles_per_frame) * TILE + 8 * wave;
            m_elo = t_rowptr[slot0];
            m_cnt = t_rowptr[slot0 + 8] - m_elo;
            const int l = lane_k;
            m_code = 0; m_tg = 0; m_w = 0.f;
            if (l < m_cnt) { m_code = t_code[m_elo + l]; m_w = t_w[m_elo + l]; m_tg = t_tgt[m_elo + l]; }
        }
    };
    if (tile >= 0) load_meta(tile);
    while (tile >= 0) {
        STAMP(0);
        walk.claim_issue();
        // lane index made opaque per tile: keeps LLVM from hoisting dozens of lane-derived loop invariants
        // (slot masks = 2 SGPRs each, address pieces) out of the tile loop and pinning registers kernel-wide
        int lane = lane_k;
        asm volatile("" : "+v"(lane));
        const int frame = tile / a.tiles_per_frame;
        const int t_in = tile - frame * a.tiles_per_frame;
        const size_t frame_row0 = (size_t)frame * a.n_per_frame;
        const float* __restrict__ xf = x + frame_row0 * C;
        const int rl0 = 8 * wave;
        int seg_first, seg_rows;          // this wave's 8 LDS rows = seg_rows consecutive node rows from seg_first
        int blocks;                       // 16-row blocks of the tile that hold nodes

        // ---- phase 1
        if constexpr (AGG == AGG_STENCIL) {
            const TileDesc td = tiles[t_in];
            const LevelDesc d = T->desc[td.level];
            seg_first = d.base + (td.r0 + wave) * d.side + td.c0;
            seg_rows = wave < td.nrows ? td.ncols : 0;
            blocks = (td.nrows + 1) >> 1;
            stencil_patch_rows(T, d, td, T->magic, xf, dis, a.n_per_frame, wave, 1, lane, s_a, s_x);
        } else {
            const int n0 = t_in * TILE;
            const int rows_here = (a.n_per_frame - n0) < TILE ? (a.n_per_frame - n0) : TILE;
            seg_first = n0 + rl0;
            const int left = rows_here - rl0;
            seg_rows = left < 0 ? 0 : (left > 8 ? 8 : left);
            blocks = (rows_here + 15) >> 4;
            const int last = a.n_per_frame - 1;
            if constexpr (AGG == AGG_CSRT) {
                // The wave's 8 slots as ONE edge list.  A row-by-row walk is a chain of dependent round trips (row pointers ->
                // edge words -> source rows, a dozen per tile and wave): 0.52 ms per launch at configs[1] whether the sources come
                // from memory or from LDS (tools/tools_csr.py).  Here the list's words (source code, weight d_src d_tgt, target
                // slot) are fetched lane-parallel in one vector load, every source row outside the tile is requested at once
                // (up to 16 in flight per wave), the sources inside the tile come out of the stash meanwhile, and a row's sum is
                // kept in a running accumulator that is added to the row's LDS line when the target changes (edges are sorted by
                // target).  Order of a row's sum: self, in-tile sources, outside sources, each group in edge_index order (fixed).
                const int slot0 = t_in * TILE + rl0;
                const int e_lo = m_elo, cnt = m_cnt;                                    // (fetched a tile ahead: load_meta)
                f32x2 raw[8];
                int id[8];
                float ws[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    id[u] = t_rows[slot0 + u];                                   // scalar
                    ws[u] = t_dis[slot0 + u];                                    // (deg + 1)^-1 of the slot's node: the self loop's weight
                    raw[u] = load_row2(xf, id[u] < 0 ? 0 : id[u], lane);
                }
                int code = m_code, tg = m_tg;
                float wv = m_w;
#ifndef EG_CSRT_GMAX
#define EG_CSRT_GMAX 16
#endif
#ifndef EG_CSRT_LUNROLL
#define EG_CSRT_LUNROLL 4
#endif
                constexpr int GMAX = EG_CSRT_GMAX;
                f32x2 gv[GMAX];
                auto issue = [&](unsigned long long mm) -> unsigned long long {          // up to GMAX outside sources, all in flight
#pragma unroll
                    for (int k = 0; k < GMAX; ++k) {
                        gv[k] = f32x2{0.f, 0.f};
                        if (mm) {
                            const int bit = __builtin_ctzll(mm);
                            gv[k] = load_row2(xf, __builtin_amdgcn_readlane(code, bit), lane);
                            mm &= mm - 1;
                        }
                    }
                    return mm;
                };
                // the first 64 edges' outside sources are requested BEFORE the barrier (they do not depend on the stash)
                unsigned long long mg = __builtin_amdgcn_ballot_w64(lane < cnt && code >= 0);
                unsigned long long ml = __builtin_amdgcn_ballot_w64(lane < cnt && code < 0);
                unsigned long long rest = issue(mg);
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    *reinterpret_cast<f32x2*>(&s_xbuf[(rl0 + u) * C + 2 * lane]) = raw[u];
                    *reinterpret_cast<f32x2*>(&s_a[(rl0 + u) * LDA + 2 * lane]) = ws[u] * raw[u];
                }
                if (lane < 8) s_ids[rl0 + lane] = t_rows[slot0 + lane];
                __syncthreads();
                int cur = -1;
                f32x2 acc = {0.f, 0.f};
                auto flush = [&]() {
                    if (cur >= 0) {
                        f32x2* p = reinterpret_cast<f32x2*>(&s_a[(rl0 + cur) * LDA + 2 * lane]);
                        *p = *p + acc;
                    }
                };
                auto add = [&](int u, const f32x2& val) {
                    if (u != cur) { flush(); cur = u; acc = f32x2{0.f, 0.f}; }
                    acc += val;
                };
                auto inside = [&](unsigned long long m) {                                // sources inside the tile: four stash reads in flight
                    while (m) {
                        constexpr int LU = EG_CSRT_LUNROLL;
                        f32x2 lv[LU];
                        float lw[LU];
                        int lu[LU];
                        bool on[LU];
#pragma unroll
                        for (int q = 0; q < LU; ++q) {
                            on[q] = m != 0;
                            lv[q] = f32x2{0.f, 0.f}; lw[q] = 0.f; lu[q] = 0;
                            if (on[q]) {
                                const int bit = __builtin_ctzll(m);
                                m &= m - 1;
                                const int src = -__builtin_amdgcn_readlane(code, bit) - 1;
                                lw[q] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wv), bit));
                                lu[q] = __builtin_amdgcn_readlane(tg, bit);
                                lv[q] = *reinterpret_cast<const f32x2*>(&s_xbuf[src * C + 2 * lane]);
                            }
                        }
#pragma unroll
                        for (int q = 0; q < LU; ++q)
                            if (on[q]) add(lu[q], lw[q] * lv[q]);
                    }
                };
                auto consume = [&](unsigned long long mm) {                              // the rows issue(mm) requested, in the same order
#pragma unroll
                    for (int k = 0; k < GMAX; ++k) {
                        if (mm) {
                            const int bit = __builtin_ctzll(mm);
                            const float wgt = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wv), bit));
                            add(__builtin_amdgcn_readlane(tg, bit), wgt * gv[k]);
                            mm &= mm - 1;
                        }
                    }
                };
                inside(ml);
                consume(mg);
                while (rest) {                                                       // (more than GMAX outside sources among 64 edges)
                    const unsigned long long m = rest;
                    rest = issue(m);
                    consume(m);
                }
                for (int base = 64; base < cnt; base += 64) {                            // (more than 64 edges into the wave's 8 rows)
                    code = 0; tg = 0; wv = 0.f;
                    if (base + lane < cnt) { code = t_code[e_lo + base + lane]; wv = t_w[e_lo + base + lane]; tg = t_tgt[e_lo + base + lane]; }
                    const bool valid = base + lane < cnt;
                    mg = __builtin_amdgcn_ballot_w64(valid && code >= 0);
                    ml = __builtin_amdgcn_ballot_w64(valid && code < 0);
                    rest = issue(mg);
                    inside(ml);
                    consume(mg);
                    while (rest) {
                        const unsigned long long m = rest;
                        rest = issue(m);
                        consume(m);
                    }
                }
                flush();
            } else if constexpr (AGG == AGG_NONE) {
                const PairLane pl{lane >> 5, lane & 31};
                f32x4 v[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int n = seg_first + 2 * k + pl.h;
                    v[k] = *reinterpret_cast<const f32x4*>(xf + ((unsigned)(n < last ? n : last) * (unsigned)C + 4u * pl.q));
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) *reinterpret_cast<f32x4*>(&s_a[(rl0 + 2 * k + pl.h) * LDA + 4 * pl.q]) = v[k];
            } else {
#pragma unroll 2
                for (int u = 0; u < 8; ++u) {
                    const int n = seg_first + u < last ? seg_first + u : last;
                    *reinterpret_cast<f32x2*>(&s_a[(rl0 + u) * LDA + 2 * lane]) =
                        produce_row<AGG>(dis, rowptr, colidx, T, xf, n, lane);
                }
            }
        }
        if (WAGG && agg_out && seg_rows > 0) {
            // the wave's own 8 aggregated rows (it wrote them itself), whole 512-B rows; a missing row repeats a real one
            const PairLane pa{lane >> 5, lane & 31};
            const int fix = pa.h < seg_rows ? pa.h : 0;
            float* ap = agg_out + (frame_row0 + seg_first) * C + 4 * pa.q;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int row = (2 * k + pa.h) < seg_rows ? 2 * k + pa.h : fix;
                *reinterpret_cast<f32x4*>(ap + (size_t)row * C) = *reinterpret_cast<const f32x4*>(&s_a[(rl0 + row) * LDA + 4 * pa.q]);
            }
        }
        walk.claim_commit();
        STAMP(1);
        __syncthreads();
        STAMP(2);
        const int next_tile = walk.fetch();
        if (next_tile >= 0) load_meta(next_tile);

        // ---- phase 2 (LDS rows without a node hold stale data; their accumulator columns are never read back)
        f32x4v acc[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[b] = f32x4v{0.f, 0.f, 0.f, 0.f};
#ifndef EG_ABL_NO_MFMA         // timing-only ablation
        mfma16_pair(s_a, 0, lane, wreg, acc[0], acc[1]);
        if (blocks > 2) mfma16_pair(s_a, 32, lane, wreg, acc[2], acc[3]);
#else
        acc[0][0] = wreg[lane & 63 ? 1 : 0] + s_a[lane]; acc[1][0] = wreg[31];
#endif
        // residual rows of this wave's segment (when they are not already in LDS): issued now so their latency
        // hides behind the barriers and the accumulator round trip through LDS.  Paired-row layout: this lane
        // handles rows 2k + h of the segment.  Everything in phase 3 is branch-free: a conditional store becomes
        // its own basic block that starts with s_waitcnt vmcnt(0) (serialised store round trips).  A lane whose
        // row holds no node re-loads / re-stores a row that does (row h if it exists, else row 0; same value).
        const PairLane pl{lane >> 5, lane & 31};
        const int fix_row = pl.h < seg_rows ? pl.h : 0;
        f32x4 res[RES_GLOBAL ? 4 : 1];
        const size_t seg_off = (frame_row0 + seg_first) * C + 4 * pl.q;
        // AGG_CSRT: the wave's 8 slots hold arbitrary nodes -- row offsets (in rows, relative to seg_first) of this lane's 4 rows
        int rowoff[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int row = (2 * k + pl.h) < seg_rows ? 2 * k + pl.h : fix_row;
            if constexpr (AGG == AGG_CSRT) rowoff[k] = (seg_rows > 0 ? s_ids[rl0 + row] : 0) - seg_first;
            else rowoff[k] = row;
        }
        // (AGG_CSRT with residual == x: the rows are the wave's own slots of the stash, read in phase 3)
        const bool res_stash = AGG == AGG_CSRT && residual == x;
        if constexpr (RES_GLOBAL) {
            const float* rp = (residual ? residual : x) + seg_off;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
#ifndef EG_ABL_NO_P3
                if (res_stash) res[k] = f32x4{0.f, 0.f, 0.f, 0.f};
                else res[k] = *reinterpret_cast<const f32x4*>(rp + (long long)rowoff[k] * C);
#else
                res[k] = f32x4{0.f, 0.f, 0.f, 0.f};
#endif
            }
        }
        STAMP(3);
        __syncthreads();
        STAMP(4);
        {
            const int j = lane & 15, q4 = lane >> 4;
#pragma unroll
            for (int b = 0; b < 4; ++b)
                *reinterpret_cast<f32x4v*>(&s_a[(16 * b + j) * LDA + 16 * wave + 4 * q4]) = acc[b];
        }
        STAMP(5);
        __syncthreads();
        STAMP(6);

        // ---- phase 3
        if (seg_rows > 0) {                                                    // uniform; false only on ragged tiles
            float* op = out + seg_off;
            const f32x4 sc = *reinterpret_cast<const f32x4*>(&s_scale[4 * pl.q]);
            const f32x4 sh = *reinterpret_cast<const f32x4*>(&s_shift[4 * pl.q]);
            f32x4 v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                // rows that hold no node compute on row fix_row's data so the duplicate store writes the same value
                const int row = (2 * k + pl.h) < seg_rows ? 2 * k + pl.h : fix_row;
                f32x4 t = *reinterpret_cast<const f32x4*>(&s_a[(rl0 + row) * LDA + 4 * pl.q]);
                t = t * sc + sh;
                if (a.relu) { t.x = fmaxf(t.x, 0.f); t.y = fmaxf(t.y, 0.f); t.z = fmaxf(t.z, 0.f); t.w = fmaxf(t.w, 0.f); }
                if constexpr (RES_GLOBAL) {
                    if (res_stash) t += *reinterpret_cast<const f32x4*>(&s_xbuf[(rl0 + row) * C + 4 * pl.q]);
                    else if (residual) t += res[k];
                }
                else { if (s_x) t += *reinterpret_cast<const f32x4*>(&s_x[(rl0 + row) * C + 4 * pl.q]); }
                v[k] = t;
            }
            if constexpr (WAGG) {
                if (stats_partial) {                                             // (uniform) rows that hold no node do not count
                    f32x4 cs = *reinterpret_cast<const f32x4*>(&s_stat[tid * 8]);
                    f32x4 cq = *reinterpret_cast<const f32x4*>(&s_stat[tid * 8 + 4]);
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float m = (2 * k + pl.h) < seg_rows ? 1.0f : 0.0f;
                        const f32x4 t = v[k] * m;
                        cs += t; cq += t * v[k];
                    }
                    *reinterpret_cast<f32x4*>(&s_stat[tid * 8]) = cs;
                    *reinterpret_cast<f32x4*>(&s_stat[tid * 8 + 4]) = cq;
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
#ifndef EG_ABL_NO_P3
                *reinterpret_cast<f32x4*>(op + (long long)rowoff[k] * C) = v[k];
#else
                if (v[k].x == 1234.5678f) *reinterpret_cast<f32x4*>(op + (long long)rowoff[k] * C) = v[k];
#endif
            }
        }
        STAMP(7);
        tile = next_tile;
    }
    if constexpr (WAGG) {
        if (stats_partial) {
            __syncthreads();
            if (tid < 2 * C) {                                                   // fixed order over the 8 waves x 2 row parities
                const int quantity = tid >> 7, c = tid & 127, q = c >> 2, e = c & 3;
                float s = 0.f;
#pragma unroll
                for (int w = 0; w < 8; ++w)
#pragma unroll
                    for (int h = 0; h < 2; ++h) s += s_stat[(w * 64 + h * 32 + q) * 8 + quantity * 4 + e];
                stats_partial[(size_t)blockIdx.x * 2 * C + tid] = s;
            }
        }
    }
    // queue walk: thread 0 -- the claiming thread, all of whose claims have returned -- counts its workgroup out; the last one
    // puts the eight queue heads and the exit counter back to 0 (atomics only: no XCD's L2 holds the line dirty), so the slice is
    // clean for whoever takes it next, eagerly or in a HIP-graph replay
    if (a.self_reset && tid == 0) {
        const int done = atomicAdd(&walk_counters[QUEUE_DONE_IDX], 1);
        if (done == (int)gridDim.x - 1) {
#pragma unroll
            for (int q = 0; q < WALK_GROUPS; ++q) atomicExch(&walk_counters[q * WALK_CTR_STRIDE], 0);
            atomicExch(&walk_counters[QUEUE_DONE_IDX], 0);
        }
    }
#ifdef EG_STAMP
    if (lane_k == 0) {
        unsigned long long* stats = reinterpret_cast<unsigned long long*>(walk_counters + WALK_GROUPS * WALK_CTR_STRIDE);
        for (int i = 0; i < 8; ++i) atomicAdd(&stats[i], st[i]);
        atomicAdd(&stats[8], 1ull);
    }
#endif
}

// aggregation only: out = A_hat x, one wave per node row, no LDS
template <int AGG>
__global__ __launch_bounds__(256) void k_aggregate(const float* __restrict__ x, float* __restrict__ out,
                                                   const float* __restrict__ dis, const int* __restrict__ rowptr,
                                                   const int* __restrict__ colidx, const Topo* __restrict__ T,
                                                   const LayerDims a) {
    const int lane = threadIdx.x & 63;
    const int wave = wave_id();
    const long long total = (long long)a.n_per_frame * a.batch;
    for (long long row = (long long)blockIdx.x * 4 + wave; row < total; row += (long long)gridDim.x * 4) {
        const int frame = (int)(row / a.n_per_frame);
        const int n = (int)(row - (long long)frame * a.n_per_frame);
        const float* __restrict__ xf = x + (size_t)frame * a.n_per_frame * C;
        const f32x2 v = produce_row<AGG>(dis, rowptr, colidx, T, xf, n, lane);
        *reinterpret_cast<f32x2*>(out + (size_t)row * C + 2 * lane) = v;
    }
}

__global__ void k_debug_xcc(int* __restrict__ out) {
    if (threadIdx.x == 0) {
        out[blockIdx.x] = xcc_id();
        out[gridDim.x + blockIdx.x] = (int)(__builtin_amdgcn_s_memtime() & 0x7fffffff);
    }
}

// Persistent grid: what is resident at once (256 CUs x 2 workgroups of 8 waves at this kernel's register
// budget).  A larger grid would run a second, under-occupied round.
static int grid_for_tiles(long long n_tiles, const Knobs& kn) {
    const long long cap = kn.grid_cap;
    long long g = n_tiles < cap ? n_tiles : cap;
    g = (g + 7) / 8 * 8;                                     // static walk modes use groups of 8
    return (int)g;
}

#define LAYER_KARGS a.x, a.W, a.scale, a.shift, a.residual, a.out, a.agg_out, a.stats_partial, a.gp.dis, a.gp.rowptr, a.gp.colidx, a.gp.topo, a.gp.tiles, a.gp.t_rows, a.gp.t_rowptr, a.gp.t_code, a.gp.t_tgt, a.gp.t_w, a.gp.t_dis, a.walk_counters, a.d

static int launch_layer(int agg, LayerArgs& a, hipStream_t stream, int* grid_out = nullptr) {
    const long long n_tiles = (long long)a.d.tiles_per_frame * a.d.batch;
    if (grid_out) *grid_out = 0;
    if (n_tiles <= 0) return EG_OK;
    if (n_tiles >= (1ll << 31)) return set_error(EG_ERR_ARG, "too many tiles");
    const dim3 grid(grid_for_tiles(n_tiles, a.knobs)), block(LAYER_THREADS);
    if (grid_out) *grid_out = (int)grid.x;
    const bool train = a.agg_out || a.stats_partial;
    int slot = -1;
    if (a.graph) {
        const int rc = a.graph->acquire_queue_slice(stream, &a.walk_counters, &slot);
        if (rc != EG_OK) return rc == EG_ERR_UNSUPPORTED ? EG_ERR_RING : rc;
    }
    a.d.walk_mode = a.walk_counters ? a.knobs.walk_mode : WALK_MOD8;
    // Batch statistics are summed per WORKGROUP over the tiles it walks: with the dynamic queue the set of tiles behind each
    // partial sum, hence the rounding of the totals, would change from launch to launch.  A static walk keeps a training step
    // bit-reproducible (tests/test_gpu_train.py::test_cfg4_train_full_batch_32_properties).
    if (a.stats_partial && a.d.walk_mode == WALK_QUEUE) a.d.walk_mode = WALK_MOD8;
    a.d.stagger = a.knobs.stagger;
    // queue walk: the kernel leaves its slice zeroed (last workgroup out), like the producer / consumer kernel -- the slice it is
    // handed is clean by the same invariant (common.h); EG_QUEUE_SELF_RESET=0 / stamp builds: a memset in front instead
#ifdef EG_STAMP
    a.d.self_reset = 0;
    if (a.d.walk_mode == WALK_QUEUE) EG_HIP_TRY(hipMemsetAsync(a.walk_counters, 0, sizeof(int) * QUEUE_SLICE_INTS, stream));
#else
    a.d.self_reset = a.d.walk_mode == WALK_QUEUE ? 1 : 0;
    if (a.d.walk_mode == WALK_QUEUE && a.knobs.queue_self_reset == 0) {      // (diagnostic: an extra memset in front of EAGER launches; gcn_layer_ps.hip)
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(stream, &cs) != hipSuccess) { (void)hipGetLastError(); cs = hipStreamCaptureStatusNone; }
        if (cs == hipStreamCaptureStatusNone) EG_HIP_TRY(hipMemsetAsync(a.walk_counters, 0, sizeof(int) * QUEUE_SLICE_INTS, stream));
    }
#endif
    const bool res_lds = (a.residual == nullptr) || (a.residual == a.x);     // epilogue residual from the LDS stash / none
    eg::LaunchTimer timer(EG_LAUNCH_SYMMETRIC, stream);
    switch (agg) {
        case AGG_NONE: hipLaunchKernelGGL((k_gcn_layer<AGG_NONE, true>), grid, block, 0, stream, LAYER_KARGS); break;
        case AGG_CSR:
        case AGG_CSRT:
            if (train) hipLaunchKernelGGL((k_gcn_layer<AGG_CSR, true, true>), grid, block, 0, stream, LAYER_KARGS);
            else if (agg == AGG_CSRT) hipLaunchKernelGGL((k_gcn_layer<AGG_CSRT, true>), grid, block, 0, stream, LAYER_KARGS);
            else hipLaunchKernelGGL((k_gcn_layer<AGG_CSR, true>), grid, block, 0, stream, LAYER_KARGS);
            break;
        default:
            if (train) hipLaunchKernelGGL((k_gcn_layer<AGG_STENCIL, true, true>), grid, block, 0, stream, LAYER_KARGS);
            else if (res_lds) hipLaunchKernelGGL((k_gcn_layer<AGG_STENCIL, false>), grid, block, 0, stream, LAYER_KARGS);
            else hipLaunchKernelGGL((k_gcn_layer<AGG_STENCIL, true>), grid, block, 0, stream, LAYER_KARGS);
            break;
    }
    if (a.graph) {
        a.graph->commit_queue_slice(slot, stream);
        a.graph->layer_launches.fetch_add(1u, std::memory_order_relaxed);
    }
    EG_HIP_TRY(hipGetLastError());
    return EG_OK;
}

static int fill_graph_args(const eg_graph* g, int batch, LayerArgs& a, int& agg) {
    if (!g) return set_error(EG_ERR_ARG, "graph is NULL");
    if (batch < 1) return set_error(EG_ERR_ARG, "batch must be >= 1");
    if (g->n_nodes * (int64_t)batch >= (1ll << 31)) return set_error(EG_ERR_ARG, "batch * nodes exceeds int32");
    a.gp.dis = g->dis;
    a.gp.rowptr = g->rowptr;
    a.gp.colidx = g->colidx;
    a.gp.topo = g->topo_dev;
    a.walk_counters = nullptr;                   // taken per launch (launch_layer): the slice is tied to the launch's stream
    a.graph = g;
    a.knobs = g->knobs;
    a.d.n_per_frame = (int)g->n_nodes;
    a.d.batch = batch;
    a.gp.tiles = g->tiles_dev;
    // (a 'grid-diagonal' topology handle carries the CSR of one frame for this kernel: its stencil lives in the producer/consumer kernel)
    agg = (g->kind == GRAPH_TOPO && !g->hybrid) ? AGG_STENCIL : AGG_CSR;
    // a CSR handle's inference launches take the clustered tiles (the train forward keeps the row-by-row aggregator: launch_layer)
    if (g->kind == GRAPH_CSR && g->t_rows) {
        agg = AGG_CSRT;
        a.gp.t_rows = g->t_rows; a.gp.t_rowptr = g->t_rowptr; a.gp.t_code = g->t_code; a.gp.t_tgt = g->t_tgt; a.gp.t_w = g->t_w; a.gp.t_dis = g->t_dis;
    }
    // implicit topology: one tile per 8x8 patch of a level; CSR: 64 consecutive rows
    a.d.tiles_per_frame = agg == AGG_STENCIL ? g->n_tiles : (int)((g->n_nodes + TILE - 1) / TILE);
    return EG_OK;
}

}  // namespace eg

using namespace eg;

int eg_launch_layer_sym(const eg_graph* g, int batch, const float* x, const float* W, const float* scale, const float* shift,
                        const float* residual, int relu, int transpose_w, float* out, float* agg_out, float* stats_partial,
                        int* grid_out, hipStream_t stream) {
    if (!x || !W || !out) return set_error(EG_ERR_ARG, "x, W and out must not be NULL");
    if (out == x || out == residual || (agg_out && (agg_out == x || agg_out == out)))
        return set_error(EG_ERR_ARG, "out / agg_out must not alias the inputs or each other");
    LayerArgs a{};
    int agg;
    const int rc = fill_graph_args(g, batch, a, agg);
    if (rc != EG_OK) return rc;
    a.x = x; a.W = W; a.scale = scale; a.shift = shift; a.residual = residual; a.out = out; a.agg_out = agg_out;
    a.stats_partial = stats_partial;
    a.d.relu = relu; a.d.transpose_w = transpose_w;
    return launch_layer(agg, a, stream, grid_out);
}

extern "C" {

int eg_debug_xcc(int* out_dev, int nblocks, eg_stream_t stream) {
    if (!out_dev || nblocks < 1) return set_error(EG_ERR_ARG, "bad argument");
    hipLaunchKernelGGL(k_debug_xcc, dim3(nblocks), dim3(256), 0, (hipStream_t)stream, out_dev);
    EG_HIP_TRY(hipGetLastError());
    return EG_OK;
}

int eg_gcn_layer_fwd(const eg_graph* g, int batch, const float* x, const float* W, const float* scale,
                     const float* shift, const float* residual, int relu, int transpose_w, float* out,
                     eg_stream_t stream) {
    if (!x || !W || !out) return set_error(EG_ERR_ARG, "x, W and out must not be NULL");
    if (out == x || out == residual) return set_error(EG_ERR_ARG, "out must not alias x or residual");
    LayerArgs a{};
    int agg;
    int rc = fill_graph_args(g, batch, a, agg);
    if (rc != EG_OK) return rc;
    // implicit topology with the residual in {none, x}: producer/consumer kernel
    rc = eg_launch_layer_ps(g, batch, x, W, scale, shift, residual, relu, transpose_w, out, nullptr, nullptr, nullptr, (hipStream_t)stream);
    if (rc != EG_ERR_UNSUPPORTED) return public_rc(rc);
    a.x = x; a.W = W; a.scale = scale; a.shift = shift; a.residual = residual; a.out = out;
    a.d.relu = relu; a.d.transpose_w = transpose_w;
    return public_rc(launch_layer(agg, a, (hipStream_t)stream));
}

int eg_gcn_layer_fwd_chain(const eg_graph* g, int batch, const float* x, const float* W, const float* scale,
                           const float* shift, const float* residual, int relu, int transpose_w, float* out,
                           const float* kidsum_in, float* kidsum_out, eg_stream_t stream) {
    if (!x || !W || !out) return set_error(EG_ERR_ARG, "x, W and out must not be NULL");
    if (out == x || out == residual) return set_error(EG_ERR_ARG, "out must not alias x or residual");
    if (kidsum_in && kidsum_in == kidsum_out) return set_error(EG_ERR_ARG, "kidsum_out must not alias kidsum_in");
    if (!kidsum_in && !kidsum_out) return eg_gcn_layer_fwd(g, batch, x, W, scale, shift, residual, relu, transpose_w, out, stream);
    if (!g || batch <= 0) return set_error(EG_ERR_ARG, "bad graph handle or batch");
    const int rc = eg_launch_layer_ps(g, batch, x, W, scale, shift, residual, relu, transpose_w, out, kidsum_in, kidsum_out,
                                      nullptr, (hipStream_t)stream);
    if (rc == EG_ERR_UNSUPPORTED)
        return set_error(EG_ERR_UNSUPPORTED, "chained layers need a topology handle with eg_graph_kidsum_rows() > 0 and residual in {NULL, x}");
    return public_rc(rc);
}

int eg_graph_fused_classifier_ok(const eg_graph* g) {
    return g && g->kind == GRAPH_TOPO && (g->kid_rows > 0 || g->flat) && g->topo.coord_base >= g->n_nodes;
}

int eg_gcn_layer_fwd_jk(const eg_graph* g, int batch, const float* x, const float* W, const float* scale, const float* shift,
                        const float* residual, int relu, float* out, const float* kidsum_in, float* kidsum_out, const float* jk_in,
                        float* jk_out, eg_stream_t stream) {
    if (!x || !W || !out || !jk_in || !jk_out) return set_error(EG_ERR_ARG, "x, W, out, jk_in and jk_out must not be NULL");
    if (out == x || out == residual || out == jk_in || jk_out == x || jk_out == out || jk_out == jk_in || jk_out == residual)
        return set_error(EG_ERR_ARG, "out / jk_out must not alias the inputs or each other");
    // the child-sum side buffers are [kidsum_rows, 128] arrays of their own: an overlap with any row array is a race
    for (const float* k : {kidsum_in, (const float*)kidsum_out})
        if (k && (k == x || k == out || k == jk_in || k == jk_out || k == residual))
            return set_error(EG_ERR_ARG, "kidsum_in / kidsum_out must not alias x, out, residual, jk_in or jk_out");
    if (kidsum_in && kidsum_in == kidsum_out) return set_error(EG_ERR_ARG, "kidsum_out must not alias kidsum_in");
    if (!g || batch <= 0) return set_error(EG_ERR_ARG, "bad graph handle or batch");
    const int rc = eg_launch_layer_ps(g, batch, x, W, scale, shift, residual, relu, 0, out, kidsum_in, kidsum_out, nullptr,
                                      (hipStream_t)stream, jk_in, jk_out);
    if (rc == EG_ERR_UNSUPPORTED)
        return set_error(EG_ERR_UNSUPPORTED, "the running JumpingKnowledge maximum needs a topology handle on the producer/consumer kernel "
                                             "(child sums available, or a single-level grid) and residual in {NULL, x}");
    return public_rc(rc);
}

int eg_gcn_layer_cls_fwd(const eg_graph* g, int batch, const float* x, const float* W, const float* scale,
                         const float* shift, const float* residual, int relu, const float* kidsum_in, const float* jk_in,
                         const float* w1, const float* s1, const float* t1, const float* w2, const float* s2, const float* t2,
                         const float* w3, const float* b3, int sigmoid, float* logits, eg_stream_t stream) {
    if (!x || !W || !logits || !w1 || !s1 || !t1 || !w2 || !s2 || !t2 || !w3 || !b3) return set_error(EG_ERR_ARG, "NULL argument");
    if (!g || batch <= 0) return set_error(EG_ERR_ARG, "bad graph handle or batch");
    if (g->kind != GRAPH_TOPO || g->topo.coord_base < g->n_nodes)
        return set_error(EG_ERR_UNSUPPORTED, "the fused classifier needs a topology handle without coordinate nodes");
    // connection nodes (the first n_conn rows of a frame) are dropped by the heads' node-type filter: logits is [batch * (n - n_conn), 4]
    eg::ClsArgs c{w1, s1, t1, w2, s2, t2, w3, b3, logits, sigmoid, g->topo.n_conn, (int)g->n_nodes - g->topo.n_conn};
    const int rc = eg_launch_layer_ps(g, batch, x, W, scale, shift, residual, relu, 0, nullptr, kidsum_in, nullptr, &c,
                                      (hipStream_t)stream, jk_in, nullptr);
    if (rc == EG_ERR_UNSUPPORTED)
        return set_error(EG_ERR_UNSUPPORTED, "the fused classifier needs eg_graph_fused_classifier_ok() and residual in {NULL, x}");
    return public_rc(rc);
}

unsigned eg_graph_ps_launches(const eg_graph* g) { return g ? g->ps_launches.load(std::memory_order_relaxed) : 0u; }
unsigned eg_graph_layer_launches(const eg_graph* g) { return g ? g->layer_launches.load(std::memory_order_relaxed) : 0u; }


int eg_gcn_aggregate(const eg_graph* g, int batch, const float* x, float* out, eg_stream_t stream) {
    if (!x || !out || x == out) return set_error(EG_ERR_ARG, "x/out NULL or aliased");
    LayerArgs a{};
    int agg;
    int rc = fill_graph_args(g, batch, a, agg);
    if (rc != EG_OK) return rc;
    a.x = x; a.out = out;
    const long long rows = (long long)a.d.n_per_frame * batch;
    long long blocks = (rows + 3) / 4;
    if (blocks > 4096) blocks = 4096;
    const dim3 grid((unsigned)blocks), block(256);
#define AGG_KARGS a.x, a.out, a.gp.dis, a.gp.rowptr, a.gp.colidx, a.gp.topo, a.d
    if (agg == AGG_STENCIL) hipLaunchKernelGGL(k_aggregate<AGG_STENCIL>, grid, block, 0, (hipStream_t)stream, AGG_KARGS);
    else hipLaunchKernelGGL(k_aggregate<AGG_CSR>, grid, block, 0, (hipStream_t)stream, AGG_KARGS);
    EG_HIP_TRY(hipGetLastError());
    return EG_OK;
}

int eg_linear128_fwd(const float* x, int64_t rows, const float* W, const float* scale, const float* shift,
                     const float* residual, int relu, int transpose_w, float* out, eg_stream_t stream) {
    if (!x || !W || !out) return set_error(EG_ERR_ARG, "x, W and out must not be NULL");
    if (out == x || out == residual) return set_error(EG_ERR_ARG, "out must not alias x or residual");
    if (rows < 0 || rows >= (1ll << 31)) return set_error(EG_ERR_ARG, "rows out of range");
    if (rows == 0) return EG_OK;
    LayerArgs a{};
    a.x = x; a.W = W; a.scale = scale; a.shift = shift; a.residual = residual; a.out = out;
    a.d.relu = relu; a.d.transpose_w = transpose_w;
    a.d.n_per_frame = (int)rows;
    a.d.batch = 1;
    a.d.tiles_per_frame = (int)((rows + TILE - 1) / TILE);
    a.knobs = process_knobs();
    return launch_layer(AGG_NONE, a, (hipStream_t)stream);
}

}  // extern "C"
