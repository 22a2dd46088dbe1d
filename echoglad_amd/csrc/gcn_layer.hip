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
};

struct LayerDims {
    int n_per_frame;       // rows per frame
    int batch;
    int tiles_per_frame;
    int relu;
    int transpose_w;
    int walk_mode;
};

struct LayerArgs {         // host-side bundle only
    const float* x;
    const float* W;
    const float* scale;
    const float* shift;
    const float* residual;
    float* out;
    GraphPtrs gp;
    LayerDims d;
    int* walk_counters;
};

template <int AGG>
__device__ inline f32x2 produce_row(const float* __restrict__ dis, const int* __restrict__ rowptr,
                                    const int* __restrict__ colidx, const Topo* __restrict__ T,
                                    const float* __restrict__ xf, int n, int lane) {
    if constexpr (AGG == AGG_STENCIL) return agg_stencil(T, xf, dis, n, lane);
    else if constexpr (AGG == AGG_CSR) return agg_csr(xf, dis, rowptr, colidx, n, lane);
    else return load_row2(xf, n, lane);
}

// One workgroup = 4 waves walks 64-row tiles (persistent, XCD-aware order).  Per tile:
//   phase 1  each wave aggregates its 16 rows (natural layout: one row = one 512-B wave access) -> LDS
//   phase 2  each wave: its 32 output channels x 64 rows on the fp32 MFMA (W slice in registers)
//   phase 3  accumulators go back through the same LDS tile so the epilogue runs in the natural
//            layout again: per-lane scale/shift, ReLU, residual, and full-row (512 B) coalesced
//            residual loads / output stores.  A wave touches only its own 16 rows in phases 3 and 1,
//            so no barrier separates a tile's phase 3 from the next tile's phase 1.
template <int AGG>
__global__ __launch_bounds__(256) void k_gcn_layer(const float* __restrict__ x, const float* __restrict__ W,
                                                   const float* __restrict__ scale, const float* __restrict__ shift,
                                                   const float* __restrict__ residual, float* __restrict__ out,
                                                   const float* __restrict__ dis, const int* __restrict__ rowptr,
                                                   const int* __restrict__ colidx, const Topo* __restrict__ T,
                                                   int* __restrict__ walk_counters, const LayerDims a) {
    __shared__ __attribute__((aligned(16))) float s_a[TILE * LDA + 4];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = wave_id();
    const int j = lane & 31, h = lane >> 5;

    float wreg[64];
    load_w_slice(W, wave, lane, a.transpose_w, wreg);
    const f32x2 sc = scale ? *reinterpret_cast<const f32x2*>(scale + 2 * lane) : f32x2{1.f, 1.f};
    const f32x2 sh = shift ? *reinterpret_cast<const f32x2*>(shift + 2 * lane) : f32x2{0.f, 0.f};

    TileWalk walk(a.walk_mode, a.tiles_per_frame * a.batch, walk_counters, reinterpret_cast<int*>(&s_a[TILE * LDA]));
    int tile;
    while (walk.next(tile)) {
        const int frame = tile / a.tiles_per_frame;
        const int n0 = (tile - frame * a.tiles_per_frame) * TILE;
        const size_t frame_row0 = (size_t)frame * a.n_per_frame;
        const float* __restrict__ xf = x + frame_row0 * C;
        const int rows_here = (a.n_per_frame - n0) < TILE ? (a.n_per_frame - n0) : TILE;
        const int rl0 = wave * (TILE / 4);
        const int rl1 = (rl0 + TILE / 4) < rows_here ? (rl0 + TILE / 4) : rows_here;

        // ---- phase 1
        if constexpr (AGG == AGG_STENCIL) {
            stencil_run_to_lds(T, xf, dis, n0 + rl0, rl0, rl1, lane, s_a);
        } else {
#pragma unroll 2
            for (int rl = rl0; rl < rl1; ++rl) {
                const f32x2 v = produce_row<AGG>(dis, rowptr, colidx, T, xf, n0 + rl, lane);
                *reinterpret_cast<f32x2*>(&s_a[rl * LDA + 2 * lane]) = v;
            }
        }
        __syncthreads();

        // ---- phase 2 (rows beyond rows_here hold stale data; their accumulator columns are never read back)
        f32x16 acc0 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        f32x16 acc1 = acc0;
        mfma_rowblock(s_a, 0, lane, wreg, acc0);
        if (rows_here > 32) mfma_rowblock(s_a, 32, lane, wreg, acc1);
        __syncthreads();
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int ch0 = 32 * wave + 8 * g + 4 * h;
            *reinterpret_cast<f32x4*>(&s_a[j * LDA + ch0]) = f32x4{acc0[4 * g], acc0[4 * g + 1], acc0[4 * g + 2], acc0[4 * g + 3]};
            *reinterpret_cast<f32x4*>(&s_a[(32 + j) * LDA + ch0]) = f32x4{acc1[4 * g], acc1[4 * g + 1], acc1[4 * g + 2], acc1[4 * g + 3]};
        }
        __syncthreads();

        // ---- phase 3
        const size_t tile_off = (frame_row0 + n0) * C + 2 * lane;
#pragma unroll 8
        for (int rl = rl0; rl < rl1; ++rl) {
            f32x2 v = *reinterpret_cast<const f32x2*>(&s_a[rl * LDA + 2 * lane]);
            v = v * sc + sh;
            if (a.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); }
            if (residual) v += *reinterpret_cast<const f32x2*>(residual + tile_off + (size_t)rl * C);
            *reinterpret_cast<f32x2*>(out + tile_off + (size_t)rl * C) = v;
        }
    }
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

static int env_int(const char* name, int dflt) {
    const char* v = getenv(name);
    return v && *v ? atoi(v) : dflt;
}

// Persistent grid: what is resident at once (256 CUs x 3 workgroups at this kernel's register
// budget).  A larger grid would run a second, under-occupied round.
static int grid_for_tiles(long long n_tiles) {
    const long long cap = env_int("EG_GRID", 768);
    long long g = n_tiles < cap ? n_tiles : cap;
    g = (g + 7) / 8 * 8;                                     // static walk modes use groups of 8
    return (int)g;
}

#define LAYER_KARGS a.x, a.W, a.scale, a.shift, a.residual, a.out, a.gp.dis, a.gp.rowptr, a.gp.colidx, a.gp.topo, a.walk_counters, a.d

static int launch_layer(int agg, LayerArgs& a, hipStream_t stream) {
    const long long n_tiles = (long long)a.d.tiles_per_frame * a.d.batch;
    if (n_tiles <= 0) return EG_OK;
    if (n_tiles >= (1ll << 31)) return set_error(EG_ERR_ARG, "too many tiles");
    const dim3 grid(grid_for_tiles(n_tiles)), block(256);
    a.d.walk_mode = a.walk_counters ? env_int("EG_WALK_MODE", WALK_QUEUE) : WALK_MOD8;
    if (a.d.walk_mode == WALK_QUEUE)
        EG_HIP_TRY(hipMemsetAsync(a.walk_counters, 0, sizeof(int) * WALK_GROUPS * WALK_CTR_STRIDE, stream));
    switch (agg) {
        case AGG_NONE: hipLaunchKernelGGL(k_gcn_layer<AGG_NONE>, grid, block, 0, stream, LAYER_KARGS); break;
        case AGG_CSR: hipLaunchKernelGGL(k_gcn_layer<AGG_CSR>, grid, block, 0, stream, LAYER_KARGS); break;
        default: hipLaunchKernelGGL(k_gcn_layer<AGG_STENCIL>, grid, block, 0, stream, LAYER_KARGS); break;
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
    a.walk_counters = g->walk_counters;
    a.d.n_per_frame = (int)g->n_nodes;
    a.d.batch = batch;
    a.d.tiles_per_frame = (int)((g->n_nodes + TILE - 1) / TILE);
    agg = g->kind == GRAPH_TOPO ? AGG_STENCIL : AGG_CSR;
    return EG_OK;
}

}  // namespace eg

using namespace eg;

extern "C" {

int eg_gcn_layer_fwd(const eg_graph* g, int batch, const float* x, const float* W, const float* scale,
                     const float* shift, const float* residual, int relu, int transpose_w, float* out,
                     eg_stream_t stream) {
    if (!x || !W || !out) return set_error(EG_ERR_ARG, "x, W and out must not be NULL");
    if (out == x || out == residual) return set_error(EG_ERR_ARG, "out must not alias x or residual");
    LayerArgs a{};
    int agg;
    int rc = fill_graph_args(g, batch, a, agg);
    if (rc != EG_OK) return rc;
    a.x = x; a.W = W; a.scale = scale; a.shift = shift; a.residual = residual; a.out = out;
    a.d.relu = relu; a.d.transpose_w = transpose_w;
    return launch_layer(agg, a, (hipStream_t)stream);
}

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
    return launch_layer(AGG_NONE, a, (hipStream_t)stream);
}

}  // extern "C"
