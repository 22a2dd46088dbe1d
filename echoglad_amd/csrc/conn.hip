// Connection nodes (use_connection_nodes; reference src/core/datasets.py:1450-1456, :1512-1515): node g - 1 of a frame is wired
// to EVERY node of aux level g (g = 1 .. naux - 1: up to 4096 nodes at 224 / 7) and to the other naux connection nodes.  Such a
// row is no stencil.  A pre-pass in front of every launch of the fused layer kernel on a connection-node handle computes, per
// frame, from the layer's input x:
//     S[l]        = sum over the nodes j of level l of d_j x_j                     (two stages, fixed order: bitwise reproducible)
//     scaled[h]   = d_h x_h                                                       what every node of level h adds to its stencil sum
//     agg[h]      = d_h ( sum_h' scaled[h'] + S[h] )        (h <= naux - 2; the last two connection nodes see only each other)
// with d = (deg + 1)^-1/2.  The layer kernel then treats a connection node's aggregated row as given (SegDesc::mode 3) and adds
// scaled[level's node] to every node of the wired levels (SegDesc::aux bits 2..).  It reads 5,460 rows per frame at 224 / 7
// (7.6 % of the layer's input) in two tiny launches.
#include "tile.h"

namespace eg {

__global__ __launch_bounds__(256) void k_conn_partial(const float* __restrict__ x, const float* __restrict__ dis, const int* __restrict__ table,
                                                      float* __restrict__ scratch, int n_per_frame, int per_frame_floats) {
    __shared__ f32x2 s_red[4][64];
    const int chunk = blockIdx.x, frame = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = wave_id();
    const int row0 = table[4 * chunk + 1], rows = table[4 * chunk + 2];
    const float* __restrict__ xf = x + (size_t)frame * n_per_frame * C;
    // four rows in flight per wave (one row per iteration made the launch a chain of memory round trips: 29 us at batch 8)
    f32x2 a4[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
    for (int r = wave; r < rows; r += 16) {
        f32x2 v[4];
        float d[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int rr = r + 4 * u;
            const bool ok = rr < rows;
            d[u] = ok ? dis[row0 + rr] : 0.f;
            v[u] = load_row2(xf, row0 + (ok ? rr : r), lane);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) a4[u] += d[u] * v[u];
    }
    const f32x2 acc = (a4[0] + a4[1]) + (a4[2] + a4[3]);
    s_red[wave][lane] = acc;
    __syncthreads();
    if (wave == 0) {
        const f32x2 t = (s_red[0][lane] + s_red[1][lane]) + (s_red[2][lane] + s_red[3][lane]);
        *reinterpret_cast<f32x2*>(scratch + (size_t)frame * per_frame_floats + (size_t)chunk * C + 2 * lane) = t;
    }
}

__global__ __launch_bounds__(128) void k_conn_final(const float* __restrict__ x, const float* __restrict__ dis, const int* __restrict__ table,
                                                    float* __restrict__ scratch, int n_per_frame, int per_frame_floats, int chunks,
                                                    int n_conn, int n_aux) {
    const int frame = blockIdx.x, c = threadIdx.x;
    float* __restrict__ sf = scratch + (size_t)frame * per_frame_floats;
    const float* __restrict__ xf = x + (size_t)frame * n_per_frame * C;
    float S[MAX_LEVELS];
#pragma unroll
    for (int l = 0; l < MAX_LEVELS; ++l) S[l] = 0.f;
    for (int k = 0; k < chunks; ++k) {                     // a level's chunks in order
        const int l = table[4 * k];
        const float v = sf[(size_t)k * C + c];
#pragma unroll
        for (int q = 0; q < MAX_LEVELS; ++q) S[q] += (q == l) ? v : 0.f;
    }
    float tot = 0.f;
    for (int h = 0; h < n_conn; ++h) {
        const float hs = dis[h] * xf[(size_t)h * C + c];
        sf[(size_t)(chunks + n_conn + h) * C + c] = hs;
        tot += hs;
    }
    for (int h = 0; h < n_conn; ++h) {
        float lvl = 0.f;
#pragma unroll
        for (int q = 0; q < MAX_LEVELS; ++q) lvl += (q == h && h <= n_aux - 2) ? S[q] : 0.f;
        sf[(size_t)(chunks + h) * C + c] = dis[h] * (tot + lvl);
    }
}

}  // namespace eg

using namespace eg;

// Pre-pass of one launch on a connection-node handle: fills slice `slot` of the handle's scratch from the layer input x.
// *base receives the slice; per frame it holds [conn_chunks] partial rows, [n_conn] aggregated rows, [n_conn] scaled rows.
int eg_launch_conn_prepass(const eg_graph* g, int batch, const float* x, int slot, hipStream_t stream, const float** base) {
    if (!g || g->n_conn <= 0) return EG_ERR_UNSUPPORTED;
    const int per_frame = (g->conn_chunks + 2 * g->n_conn) * C;
    float* slice = nullptr;
    {
        std::lock_guard<std::mutex> lock(g->conn_mutex);
        if (batch > g->conn_cap || !g->conn_scratch) {
            // more frames than the scratch holds: a larger one.  Not while a stream is being captured into a HIP graph (no allocation
            // there; nn's warm-up run in front of a capture has grown it).  The smaller scratch is RETIRED, not freed: HIP graphs
            // captured earlier hold its slices in their kernel nodes and replay on them, and a launch on another thread may have
            // taken its slice pointer already (eg_graph_destroy frees the retired ones).
            hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
            if (hipStreamIsCapturing(stream, &cs) != hipSuccess) { (void)hipGetLastError(); cs = hipStreamCaptureStatusNone; }
            if (cs != hipStreamCaptureStatusNone)
                return set_error(EG_ERR_UNSUPPORTED, "connection-node scratch too small for this batch during stream capture: run the call once outside the capture");
            int cap = g->conn_cap > 0 ? g->conn_cap : 8;
            while (cap < batch) cap *= 2;
            float* bigger = nullptr;
            EG_HIP_TRY(hipMalloc((void**)&bigger, sizeof(float) * (size_t)per_frame * cap * QUEUE_SLOTS));
            if (g->conn_scratch) g->conn_retired.push_back(g->conn_scratch);
            g->conn_scratch = bigger;
            g->conn_cap = cap;
        }
        slice = g->conn_scratch + (size_t)slot * g->conn_cap * per_frame;
    }
    if (g->conn_chunks > 0)
        hipLaunchKernelGGL(k_conn_partial, dim3(g->conn_chunks, batch), dim3(256), 0, stream, x, (const float*)g->dis, (const int*)g->conn_table,
                           slice, (int)g->n_nodes, per_frame);
    hipLaunchKernelGGL(k_conn_final, dim3(batch), dim3(128), 0, stream, x, (const float*)g->dis, (const int*)g->conn_table, slice, (int)g->n_nodes,
                       per_frame, g->conn_chunks, g->n_conn, g->topo.n_aux);
    EG_HIP_TRY(hipGetLastError());
    *base = slice;
    return EG_OK;
}
