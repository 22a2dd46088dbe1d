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

enum Agg { AGG_NONE = 0, AGG_CSR = 1, AGG_STENCIL = 2 };

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
enum { KIND_AUX = 0, KIND_MAIN = 1, KIND_COORD = 2 };
#ifndef EG_MAIN_U
#define EG_MAIN_U 2      // main-grid nodes aggregated per batch of loads (2 keeps the kernel at 3 waves/SIMD)
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
            v[u][s] = load_row2(xf, id[s], lane);
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

// rows [rl, rl_end) of the LDS tile <- aggregated rows of nodes n_first, n_first+1, ...
__device__ inline void stencil_run_to_lds(const Topo* __restrict__ T, const float* __restrict__ xf,
                                          const float* __restrict__ dis, int n_first, int rl, int rl_end, int lane,
                                          float* s_a) {
    if (rl >= rl_end) return;
    RunPos p;
    LevelDesc d = run_decode(T, n_first, p);
    while (rl < rl_end) {
        if (p.n >= d.end) d = run_decode(T, p.n, p);
        const int left = rl_end - rl, in_level = d.end - p.n;
        const int avail = left < in_level ? left : in_level;
        if (d.kind == KIND_MAIN && avail >= EG_MAIN_U) { run_group<EG_MAIN_U, 6>(d, p, xf, dis, lane, s_a, rl); rl += EG_MAIN_U; }
#ifndef EG_NO_GEN2
        else if (avail >= 2) { run_group<2, MAX_SLOTS>(d, p, xf, dis, lane, s_a, rl); rl += 2; }
#endif
        else { run_group<1, MAX_SLOTS>(d, p, xf, dis, lane, s_a, rl); rl += 1; }
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
__device__ inline void mfma_chunk(const f32x4 (&av)[4], const float (&wreg)[64], int t0, f32x16& acc) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wreg[4 * (t0 + t) + 0], av[t].x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wreg[4 * (t0 + t) + 1], av[t].y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wreg[4 * (t0 + t) + 2], av[t].z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wreg[4 * (t0 + t) + 3], av[t].w, acc, 0, 0, 0);
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
    mfma_chunk(a0, wreg, 0, acc);
#pragma unroll
    for (int t = 0; t < 4; ++t) a0[t] = ap[8 + t];
    __builtin_amdgcn_sched_barrier(0);
    mfma_chunk(a1, wreg, 4, acc);
#pragma unroll
    for (int t = 0; t < 4; ++t) a1[t] = ap[12 + t];
    __builtin_amdgcn_sched_barrier(0);
    mfma_chunk(a0, wreg, 8, acc);
    __builtin_amdgcn_sched_barrier(0);
    mfma_chunk(a1, wreg, 12, acc);
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
