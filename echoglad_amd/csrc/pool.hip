// Average-pool pyramid of the frame embedding (gfx950): the head of the reference's create_node_pixels
// (src/core/models.py:511-521: for every frame and every aux level g, F.adaptive_avg_pool2d(frame, (2^g, 2^g))) as ONE launch
// for all levels of all frames, and its gradient as one gather pass.
//
// Under torch the seven levels of a 224 x 224 frame are seven launches of adaptive_average_pool (219 us each at batch 1: a
// thread per output walks its window) and, backwards, seven launches of atomic_adaptive_average_gradinput (254 us each, float
// atomics: not reproducible run to run) -- 3.3 ms of a batch-1 training step whose GNN stack takes 1 ms.
//
// adaptive_avg_pool2d, output (i, j) of a p x p grid over an F x F plane: mean over rows [floor(i F / p), ceil((i + 1) F / p))
// and the same columns.  Forward: one workgroup per (frame, channel, half of the plane's rows -- for even F the boundary F / 2
// is a window boundary of every even p); the levels are walked fine to coarse; a level whose windows tile the plane exactly
// (F % 2p == 0 for the finer level 2p that is also wanted) is the mean of 2 x 2 values of the finer level, taken from LDS;
// every other level sums its windows straight from the plane (L1 / L2 hits after the first level).  Backward: a thread per
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
    int derive[PP_MAX_LEVELS];           // 1: mean of 2 x 2 values of the level in front of it (which is in LDS)
    int keep[PP_MAX_LEVELS];             // 1: the level behind it derives from this one -> its half goes to LDS too
    int n_levels, frame, halves;
    long long planes;
};

__global__ __launch_bounds__(PP_THREADS) void k_pool_pyramid_fwd(const PoolArgs a) {
    __shared__ float s_lvl[2][PP_LDS_FLOATS];
    const long long plane = blockIdx.x / a.halves;
    const int half = blockIdx.x % a.halves;
    const int F = a.frame, tid = threadIdx.x;
    const float* xp = a.x + (size_t)plane * F * F;
    int cur = 0;
    for (int l = 0; l < a.n_levels; ++l) {
        const int p = a.side[l];
        const int rows = p / a.halves, i0 = half * rows;           // this workgroup's output rows [i0, i0 + rows)
        float* op = a.out[l] + (size_t)plane * p * p;
        float* keep = a.keep[l] ? s_lvl[cur ^ 1] : nullptr;
        if (a.derive[l]) {
            const float* src = s_lvl[cur];                          // the finer level's half: [2 rows][2 p]
            for (int o = tid; o < rows * p; o += PP_THREADS) {
                const int i = o / p, j = o - i * p;
                const float* q = src + (2 * i) * (2 * p) + 2 * j;
                const float v = 0.25f * ((q[0] + q[1]) + (q[2 * p] + q[2 * p + 1]));
                op[(size_t)(i0 + i) * p + j] = v;
                if (keep) keep[o] = v;
            }
        } else {
            for (int o = tid; o < rows * p; o += PP_THREADS) {
                const int i = i0 + o / p, j = o % p;
                const int rs = (int)(((long long)i * F) / p), re = (int)(((long long)(i + 1) * F + p - 1) / p);
                const int cs = (int)(((long long)j * F) / p), ce = (int)(((long long)(j + 1) * F + p - 1) / p);
                float sum = 0.f;
                for (int r = rs; r < re; ++r) {
                    const float* row = xp + (size_t)r * F;
                    for (int c = cs; c < ce; ++c) sum += row[c];
                }
                const float v = sum / (float)((re - rs) * (ce - cs));
                op[(size_t)i * p + j] = v;
                if (keep) keep[o] = v;
            }
        }
        if (keep) { __syncthreads(); cur ^= 1; }
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

struct WinEntry { int w0, n; float inv0, inv1; };          // first window that contains the index, how many (1 or 2), 1 / their lengths

__device__ inline WinEntry windows_of(int r, int F, int p) {
    WinEntry e{0, 0, 0.f, 0.f};
    const int i0 = (int)(((long long)r * p) / F);
    for (int i = i0 - 1; i <= i0 + 1; ++i) {
        if (i < 0 || i >= p) continue;
        const int s = (int)(((long long)i * F) / p), t = (int)(((long long)(i + 1) * F + p - 1) / p);
        if (s <= r && r < t) {
            if (e.n == 0) { e.w0 = i; e.inv0 = 1.0f / (float)(t - s); }
            else if (e.n == 1) e.inv1 = 1.0f / (float)(t - s);
            e.n += 1;
        }
    }
    if (e.n > 2) e.n = 2;                 // (windows of an adaptive pooling with F >= p overlap their neighbours only)
    return e;
}

constexpr int PB_BAND = 16;
constexpr int PB_MAX_F = 512;

__global__ __launch_bounds__(PP_THREADS) void k_pool_pyramid_bwd(const PoolBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char pb_smem[];
    const int F = a.frame, L = a.n_levels;
    WinEntry* s_col = reinterpret_cast<WinEntry*>(pb_smem);                 // [L][F]
    WinEntry* s_row = s_col + (size_t)L * F;                                // [L][band]
    const int bands = (F + a.band - 1) / a.band;
    const long long plane = blockIdx.x / bands;
    const int r0 = (int)(blockIdx.x % bands) * a.band;
    const int nrows = min(a.band, F - r0);
    const int tid = threadIdx.x;
    for (int o = tid; o < L * F; o += PP_THREADS) s_col[o] = windows_of(o % F, F, a.side[o / F]);
    for (int o = tid; o < L * a.band; o += PP_THREADS) {
        const int rr = o % a.band;
        if (rr < nrows) s_row[o] = windows_of(r0 + rr, F, a.side[o / a.band]);
    }
    __syncthreads();
    const size_t pbase = (size_t)plane * F * F;
    for (int o = tid; o < nrows * F; o += PP_THREADS) {
        const int rr = o / F, c = o - rr * F;
        float acc = a.g_frame ? a.g_frame[pbase + (size_t)(r0 + rr) * F + c] : 0.f;
        for (int l = 0; l < L; ++l) {
            const float* g = a.g[l];
            if (!g) continue;
            const int p = a.side[l];
            const WinEntry er = s_row[l * a.band + rr], ec = s_col[l * F + c];
            const float* gp = g + (size_t)plane * p * p + (size_t)er.w0 * p + ec.w0;
            float v = gp[0] * (er.inv0 * ec.inv0);
            if (ec.n > 1) v += gp[1] * (er.inv0 * ec.inv1);
            if (er.n > 1) {
                v += gp[p] * (er.inv1 * ec.inv0);
                if (ec.n > 1) v += gp[p + 1] * (er.inv1 * ec.inv1);
            }
            acc += v;
        }
        a.dx[pbase + (size_t)(r0 + rr) * F + c] = acc;
    }
}

}  // namespace eg

using namespace eg;

extern "C" {

int eg_avg_pool_pyramid_fwd(const float* x, int64_t planes, int frame, const int* level_side, int n_levels, float* const* level_maps,
                            eg_stream_t stream) {
    if (!x || !level_side || !level_maps || planes < 1 || frame < 1 || n_levels < 1 || n_levels > PP_MAX_LEVELS)
        return set_error(EG_ERR_ARG, "bad argument");
    if (planes * 2 >= (1ll << 31)) return set_error(EG_ERR_ARG, "too many planes");
    PoolArgs a{};
    a.x = x; a.frame = frame; a.n_levels = n_levels; a.planes = planes;
    // fine to coarse inside the kernel; the caller's order is coarse to fine (level_side ascending, as the node rows are laid out)
    bool all_even = frame % 2 == 0;
    for (int l = 0; l < n_levels; ++l) {
        const int p = level_side[n_levels - 1 - l];
        if (!level_maps[n_levels - 1 - l] || p < 1 || p > frame) return set_error(EG_ERR_ARG, "NULL level map or a side outside [1, frame]");
        if (l > 0 && p >= a.side[l - 1]) return set_error(EG_ERR_ARG, "level sides must be strictly ascending");
        a.side[l] = p; a.out[l] = level_maps[n_levels - 1 - l];
        all_even = all_even && p % 2 == 0;
    }
    a.halves = all_even ? 2 : 1;         // (an odd side or frame: one workgroup per plane)
    for (int l = 0; l < n_levels; ++l) {
        a.derive[l] = 0; a.keep[l] = 0;
        if (l > 0) {
            const int pf = a.side[l - 1], p = a.side[l];
            // the finer level's windows tile the plane exactly and pair up into this level's: mean of 2 x 2 means
            if (pf == 2 * p && frame % pf == 0 && (pf / a.halves) * pf <= PP_LDS_FLOATS && (pf / a.halves) % 2 == 0) {
                a.derive[l] = 1; a.keep[l - 1] = 1;
            }
        }
    }
    hipLaunchKernelGGL(k_pool_pyramid_fwd, dim3((unsigned)(planes * a.halves)), dim3(PP_THREADS), 0, (hipStream_t)stream, a);
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
    const size_t lds = sizeof(WinEntry) * (size_t)n_levels * (frame + PB_BAND);
    if (lds > 64 * 1024) {
        static std::atomic<bool> attr_set[64];
        int dev = 0;
        EG_HIP_TRY(hipGetDevice(&dev));
        if (dev < 0 || dev >= 64 || !attr_set[dev].load(std::memory_order_acquire)) {
            EG_HIP_TRY(hipFuncSetAttribute((const void*)k_pool_pyramid_bwd, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            if (dev >= 0 && dev < 64) attr_set[dev].store(true, std::memory_order_release);
        }
    }
    if (lds > 160 * 1024) return set_error(EG_ERR_UNSUPPORTED, "window tables do not fit LDS");
    hipLaunchKernelGGL(k_pool_pyramid_bwd, dim3((unsigned)(planes * bands)), dim3(PP_THREADS), lds, (hipStream_t)stream, a);
    EG_HIP_TRY(hipGetLastError());
    return EG_OK;
}

}  // extern "C"
