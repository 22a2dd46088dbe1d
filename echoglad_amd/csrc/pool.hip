// Average-pool pyramid of the frame embedding (gfx950): the head of the reference's create_node_pixels
// (src/core/models.py:511-521: for every frame and every aux level g, F.adaptive_avg_pool2d(frame, (2^g, 2^g))) as ONE launch
// for all levels of all frames, and its gradient as one gather pass.
//
// Under torch the seven levels of a 224 x 224 frame are seven launches of adaptive_average_pool (219 us each at batch 1: a
// thread per output walks its window) and, backwards, seven launches of atomic_adaptive_average_gradinput (254 us each, float
// atomics: not reproducible run to run) -- 3.3 ms of a batch-1 training step whose GNN stack takes 1 ms.
//
// adaptive_avg_pool2d, output (i, j) of a p x p grid over an F x F plane: mean over rows [floor(i F / p), ceil((i + 1) F / p))
// and the same columns.  Forward, two launches: (1) every level that has to be summed from the plane, a thread per output over
// all such levels at once (at 224 x 224: sides 128, 64, 32 -- windows of 2 x 2, 4 x 4, 7 x 7 pixels; the plane is read from HBM
// once and from L2 after that); (2) the levels whose windows tile the plane exactly and pair up -- side p when the finer level
// 2p is wanted too and F % 2p == 0 -- as means of 2 x 2 values of the finer level, one workgroup per plane walking the chain in
// LDS (16, 8, 4, 2).  Backward: a thread per
// pixel GATHERS -- per level, the (at most two) windows that contain its row and its column come from per-workgroup LDS tables
// -- so there is no atomic and the result is bit-reproducible.
#include <stdlib.h>

#include "common.h"

namespace eg {

constexpr int PP_MAX_LEVELS = 16;
constexpr int PP_THREADS = 256;
constexpr int PP_LDS_FLOATS = 8192;      // one derived-level source half: (p / 2) * p floats, p <= 128

struct PoolArgs {
    const float* x;                      // [planes, F, F]
    float* out[PP_MAX_LEVELS];           // [planes, p_l, p_l], levels fine to coarse
    int side[PP_MAX_LEVELS];
    int derive[PP_MAX_LEVELS];           // 1: mean of 2 x 2 values of the level in front of it
    int first[PP_MAX_LEVELS + 1];        // prefix sum of p_l^2 over the DIRECT levels (derived ones add nothing)
    int n_levels, frame;
    long long planes;
};

// every level that is summed straight from the plane: a thread per output, consecutive threads = consecutive columns of a row
__global__ __launch_bounds__(PP_THREADS) void k_pool_direct(const PoolArgs a) {
    const int per_plane = a.first[a.n_levels];
    const long long gid = (long long)blockIdx.x * PP_THREADS + threadIdx.x;
    if (gid >= a.planes * per_plane) return;
    const long long plane = gid / per_plane;
    const int o = (int)(gid - plane * per_plane);
    int l = 0;
    for (int k = 1; k < a.n_levels; ++k) l += o >= a.first[k] ? 1 : 0;
    // (a.first repeats over derived levels, so l may sit on one: step back to the direct level that owns the range)
    while (a.derive[l]) --l;
    const unsigned F = a.frame, p = a.side[l];
    const unsigned oo = o - a.first[l], i = oo / p, j = oo - i * p;
    const unsigned rs = i * F / p, re = ((i + 1) * F + p - 1) / p, cs = j * F / p, ce = ((j + 1) * F + p - 1) / p;
    const float* xp = a.x + (size_t)plane * F * F;
    float sum = 0.f;
    for (unsigned r = rs; r < re; ++r) {
        const float* row = xp + (size_t)r * F;
        for (unsigned c = cs; c < ce; ++c) sum += row[c];
    }
    a.out[l][(size_t)plane * p * p + oo] = sum / (float)((re - rs) * (ce - cs));
}

// the levels whose windows tile the plane exactly and pair up: level p = mean of 2 x 2 values of level 2p, walked fine to coarse
// inside one workgroup per plane (the chain's head comes from global memory, everything behind it from LDS)
__global__ __launch_bounds__(PP_THREADS) void k_pool_derived(const PoolArgs a) {
    __shared__ float s_lvl[2][PP_LDS_FLOATS];
    const long long plane = blockIdx.x;
    const int tid = threadIdx.x;
    int cur = 0;
    bool have = false;                    // s_lvl[cur] holds level l - 1
    for (int l = 1; l < a.n_levels; ++l) {
        if (!a.derive[l]) { have = false; continue; }
        const int p = a.side[l], pf = 2 * p;
        const float* gsrc = a.out[l - 1] + (size_t)plane * pf * pf;
        const bool keep = l + 1 < a.n_levels && a.derive[l + 1] && p * p <= PP_LDS_FLOATS;
        float* op = a.out[l] + (size_t)plane * p * p;
        for (int o = tid; o < p * p; o += PP_THREADS) {
            const int i = o / p, j = o - i * p;
            float v;
            if (have) {
                const float* q = s_lvl[cur] + (2 * i) * pf + 2 * j;
                v = 0.25f * ((q[0] + q[1]) + (q[pf] + q[pf + 1]));
            } else {
                const float* q = gsrc + (size_t)(2 * i) * pf + 2 * j;
                v = 0.25f * ((q[0] + q[1]) + (q[pf] + q[pf + 1]));
            }
            op[o] = v;
            if (keep) s_lvl[cur ^ 1][o] = v;
        }
        __syncthreads();
        have = keep;
        cur ^= 1;
    }
}

// ---- backward: dx[r, c] = g_frame[r, c] + sum over levels, over the windows (i, j) that contain (r, c), of g_l[i, j] / area(i, j)
struct PoolBwdArgs {
    const float* g[PP_MAX_LEVELS];       // [planes, p_l, p_l]  (NULL: no gradient for that level)
    const float* g_frame;                // [planes, F, F] or NULL: added to the result (the frame is the finest level of the node array itself)
    float* dx;                           // [planes, F, F]
    int side[PP_MAX_LEVELS];
    int n_levels, frame, band;           // band: rows per workgroup
    long long planes;
};

struct WinEntry { int w0, w1; float inv0, inv1; };          // the (at most two) windows that contain an index, 1 / their lengths (inv1 = 0, w1 = w0: only one)

__device__ inline WinEntry windows_of(int r, int F, int p) {
    WinEntry e{0, 0, 0.f, 0.f};
    int n = 0;
    const int i0 = (int)((float)r * (float)p / (float)F);            // within one of the true floor(r p / F): the three candidates below cover it
    for (int i = i0 - 1; i <= i0 + 1; ++i) {
        if (i < 0 || i >= p) continue;
        // start_i = floor(i F / p) <= r  <=>  i F < (r + 1) p;   r < end_i = ceil((i + 1) F / p)  <=>  r p < (i + 1) F   (F, p <= 512)
        if ((unsigned)i * (unsigned)F < (unsigned)(r + 1) * (unsigned)p && (unsigned)r * (unsigned)p < (unsigned)(i + 1) * (unsigned)F) {
            const int s = (int)((unsigned)i * (unsigned)F / (unsigned)p), t = (int)(((unsigned)(i + 1) * (unsigned)F + p - 1) / (unsigned)p);
            if (n == 0) { e.w0 = e.w1 = i; e.inv0 = 1.0f / (float)(t - s); }
            else if (n == 1) { e.w1 = i; e.inv1 = 1.0f / (float)(t - s); }     // (windows of an adaptive pooling with F >= p overlap their neighbours only)
            n += 1;
        }
    }
    return e;
}

constexpr int PB_BAND = 28;
constexpr int PB_MAX_F = 512;

// workgroup = (plane, band of rows); thread = column(s): its column's windows of every level sit in registers for the whole band, the
// row's windows come from LDS (one address per wave: a broadcast); per pixel and level four loads that depend on nothing but the tables
template <int LMAX>
__global__ __launch_bounds__(PP_THREADS) void k_pool_pyramid_bwd(const PoolBwdArgs a) {
    __shared__ WinEntry s_row[LMAX * PB_BAND];
    const int F = a.frame, L = a.n_levels;
    const int bands = (F + PB_BAND - 1) / PB_BAND;
    const long long plane = blockIdx.x / bands;
    const int r0 = (int)(blockIdx.x % bands) * PB_BAND;
    const int nrows = min(PB_BAND, F - r0);
    const int tid = threadIdx.x;
    for (int o = tid; o < L * PB_BAND; o += PP_THREADS) {
        const int rr = o % PB_BAND;
        if (rr < nrows) s_row[o] = windows_of(r0 + rr, F, a.side[o / PB_BAND]);
    }
    __syncthreads();
    const size_t pbase = (size_t)plane * F * F;
    for (int c = tid; c < F; c += PP_THREADS) {
        WinEntry ec[LMAX];
#pragma unroll
        for (int l = 0; l < LMAX; ++l) ec[l] = l < L ? windows_of(c, F, a.side[l]) : WinEntry{0, 0, 0.f, 0.f};
        // four rows at a time: every load of the four pixels (frame gradient + 4 per level) is issued before the first use
        const float* __restrict__ gf = a.g_frame;
        float* __restrict__ dxp = a.dx;
        for (int rb = 0; rb < nrows; rb += 4) {
            float acc[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int rr = rb + u < nrows ? rb + u : nrows - 1;
                acc[u] = gf ? gf[pbase + (size_t)(r0 + rr) * F + c] : 0.f;
            }
#pragma unroll
            for (int l = 0; l < LMAX; ++l) {
                if (l >= L || !a.g[l]) continue;
                const int p = a.side[l];
                const float* __restrict__ gl = a.g[l] + (size_t)plane * p * p;
                float v[4][4];
                WinEntry er[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int rr = rb + u < nrows ? rb + u : nrows - 1;
                    er[u] = s_row[l * PB_BAND + rr];
                    const float* g0 = gl + er[u].w0 * p;
                    const float* g1 = gl + er[u].w1 * p;
                    v[u][0] = g0[ec[l].w0]; v[u][1] = g0[ec[l].w1]; v[u][2] = g1[ec[l].w0]; v[u][3] = g1[ec[l].w1];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    acc[u] += (v[u][0] * ec[l].inv0 + v[u][1] * ec[l].inv1) * er[u].inv0 + (v[u][2] * ec[l].inv0 + v[u][3] * ec[l].inv1) * er[u].inv1;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (rb + u < nrows) dxp[pbase + (size_t)(r0 + rb + u) * F + c] = acc[u];
        }
    }
}

}  // namespace eg

using namespace eg;

extern "C" {

int eg_avg_pool_pyramid_fwd(const float* x, int64_t planes, int frame, const int* level_side, int n_levels, float* const* level_maps,
                            eg_stream_t stream) {
    if (!x || !level_side || !level_maps || planes < 1 || frame < 1 || frame > 32768 || n_levels < 1 || n_levels > PP_MAX_LEVELS)
        return set_error(EG_ERR_ARG, "bad argument");
    PoolArgs a{};
    a.x = x; a.frame = frame; a.n_levels = n_levels; a.planes = planes;
    // fine to coarse inside the kernels; the caller's order is coarse to fine (level_side ascending, as the node rows are laid out)
    for (int l = 0; l < n_levels; ++l) {
        const int p = level_side[n_levels - 1 - l];
        if (!level_maps[n_levels - 1 - l] || p < 1 || p > frame) return set_error(EG_ERR_ARG, "NULL level map or a side outside [1, frame]");
        if (l > 0 && p >= a.side[l - 1]) return set_error(EG_ERR_ARG, "level sides must be strictly ascending");
        a.side[l] = p; a.out[l] = level_maps[n_levels - 1 - l];
    }
    int n_derived = 0;
    long long per_plane = 0;
    for (int l = 0; l < n_levels; ++l) {
        a.derive[l] = 0;
        // the finer level's windows tile the plane exactly and pair up into this level's: the mean of 2 x 2 means
        if (l > 0 && a.side[l - 1] == 2 * a.side[l] && frame % a.side[l - 1] == 0) { a.derive[l] = 1; n_derived += 1; }
        a.first[l] = (int)per_plane;
        if (!a.derive[l]) per_plane += (long long)a.side[l] * a.side[l];
    }
    a.first[n_levels] = (int)per_plane;
    const long long total = planes * per_plane;
    if (total >= (1ll << 31) * PP_THREADS || per_plane >= (1ll << 31)) return set_error(EG_ERR_ARG, "too many pooled outputs");
    hipLaunchKernelGGL(k_pool_direct, dim3((unsigned)((total + PP_THREADS - 1) / PP_THREADS)), dim3(PP_THREADS), 0, (hipStream_t)stream, a);
    if (n_derived > 0) {
        if (planes >= (1ll << 31)) return set_error(EG_ERR_ARG, "too many planes");
        hipLaunchKernelGGL(k_pool_derived, dim3((unsigned)planes), dim3(PP_THREADS), 0, (hipStream_t)stream, a);
    }
    EG_HIP_TRY(hipGetLastError());
    return EG_OK;
}

int eg_avg_pool_pyramid_bwd(const float* const* level_grads, const float* frame_grad, int64_t planes, int frame, const int* level_side,
                            int n_levels, float* dx, eg_stream_t stream) {
    if (!level_grads || !level_side || !dx || planes < 1 || frame < 1 || frame > PB_MAX_F || n_levels < 1 || n_levels > PP_MAX_LEVELS)
        return set_error(EG_ERR_ARG, "bad argument (frames up to 512 x 512)");
    PoolBwdArgs a{};
    a.g_frame = frame_grad; a.dx = dx; a.frame = frame; a.n_levels = n_levels; a.band = PB_BAND; a.planes = planes;
    for (int l = 0; l < n_levels; ++l) {
        if (level_side[l] < 1 || level_side[l] > frame) return set_error(EG_ERR_ARG, "a side outside [1, frame]");
        a.g[l] = level_grads[l]; a.side[l] = level_side[l];
    }
    const long long bands = (frame + PB_BAND - 1) / PB_BAND;
    if (planes * bands >= (1ll << 31)) return set_error(EG_ERR_ARG, "too many planes");
    const dim3 grid((unsigned)(planes * bands));
    if (n_levels <= 8) hipLaunchKernelGGL(k_pool_pyramid_bwd<8>, grid, dim3(PP_THREADS), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(k_pool_pyramid_bwd<PP_MAX_LEVELS>, grid, dim3(PP_THREADS), 0, (hipStream_t)stream, a);
    EG_HIP_TRY(hipGetLastError());
    return EG_OK;
}

}  // extern "C"
