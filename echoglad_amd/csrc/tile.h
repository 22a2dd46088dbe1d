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
__device__ inline f32x2 load_row2(const float* __restrict__ base, int row, int lane) {
    return *reinterpret_cast<const f32x2*>(base + (size_t)row * C + 2 * lane);
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

// acc[m][n]: n = lane&31 = row (row0 + n), m = (reg&3) + 8*(reg>>2) + 4*(lane>>5) = channel within the wave's 32
__device__ inline void mfma_rowblock(const float* s_a, int row0, int lane, const float (&wreg)[64], f32x16& acc) {
    const int j = lane & 31, h = lane >> 5;
    const f32x4* ap = reinterpret_cast<const f32x4*>(s_a + (row0 + j) * LDA + 64 * h);
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        const f32x4 av = ap[t];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wreg[4 * t + 0], av.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wreg[4 * t + 1], av.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wreg[4 * t + 2], av.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wreg[4 * t + 3], av.w, acc, 0, 0, 0);
    }
}

// XCD-aware persistent tile walk: workgroups with equal blockIdx % 8 share an XCD (and its
// L2) under round-robin dispatch, so each such group walks one contiguous chunk of tiles —
// vertically adjacent grid rows are then served from the same L2.  Pure speed choice.
struct TileWalk {
    int chunk, base, local, stride, n_tiles;
    __device__ TileWalk(int n_tiles_) : n_tiles(n_tiles_) {
        const int groups = 8;
        const int g = blockIdx.x % groups, slot = blockIdx.x / groups;
        stride = (gridDim.x + groups - 1 - g) / groups;      // workgroups in this group
        chunk = (n_tiles + groups - 1) / groups;
        base = g * chunk;
        local = slot;
    }
    __device__ bool next(int& tile) {
        if (local >= chunk) return false;
        tile = base + local;
        local += stride;
        return tile < n_tiles;
    }
};

}  // namespace eg
