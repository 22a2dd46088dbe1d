// Training-mode pieces of the GNN layer for gfx950 (reference src/core/models.py:328-335 in train mode and
// its autograd backward):
//   eg_colsum128      db = sum_rows dy                                   (GCNConv.bias gradient)
//   eg_dweight128     dW[o][i] = sum_r g[r][o] * x[r][i]  on fp32 MFMA   (GCNConv.lin.weight gradient, g = A_hat dy)
//   eg_bn_stats       per-channel batch mean / biased variance over ALL rows of the batch (BatchNorm1d train)
//   eg_bn_act_fwd     y = relu(dropout(z * scale + shift)) + residual    (BN affine + Dropout + ReLU + residual)
//   eg_bn_act_bwd     dz, dgamma, dbeta from dy (two passes: reduce, apply); dropout mask regenerated from the seed
// All reductions are two-stage (per-workgroup partials in a caller-provided workspace, then one fixed-order
// pass), so results are bitwise reproducible run to run: no float atomics anywhere.
#include <stdlib.h>

#include <map>
#include <mutex>

#include "train_common.h"

namespace eg {

constexpr int RED_BLOCKS = 1024;           // stage-1 workgroups of the column reductions
constexpr int RED_THREADS = 256;

// ---- stage 1: per-workgroup column partials of up to NQ quantities ---------------------------------
// lane = channel pair; a wave walks rows; 4 waves of a block are combined through LDS.
template <int NQ, typename F>
__device__ inline void column_partials(long long rows, double* __restrict__ partial, F&& row_values) {
    __shared__ double s_red[4][NQ][C];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double acc[NQ][2];
#pragma unroll
    for (int q = 0; q < NQ; ++q) acc[q][0] = acc[q][1] = 0.0;
    for (long long r = (long long)blockIdx.x * 4 + wave; r < rows; r += (long long)gridDim.x * 4) {
        f32x2 v[NQ];
        row_values(r, lane, v);
#pragma unroll
        for (int q = 0; q < NQ; ++q) { acc[q][0] += (double)v[q].x; acc[q][1] += (double)v[q].y; }
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q) { s_red[wave][q][2 * lane] = acc[q][0]; s_red[wave][q][2 * lane + 1] = acc[q][1]; }
    __syncthreads();
    for (int i = threadIdx.x; i < NQ * C; i += RED_THREADS) {
        const int q = i / C, c = i % C;
        partial[((size_t)blockIdx.x * NQ + q) * C + c] = s_red[0][q][c] + s_red[1][q][c] + s_red[2][q][c] + s_red[3][q][c];
    }
}

// one workgroup (RED_F32_THREADS = 1024 threads) per 32 columns: thread (col = t & 31, slice = t >> 5) sums every 32nd partial
// in ascending order, the 32 slice sums are added in a fixed tree: the same bits on every run.  Launch with ceil(n / 32)
// workgroups.  (8 slices of 256 threads made every one of these launches a 20 - 40 us chain of dependent adds -- nine of them per
// training step.)
__global__ __launch_bounds__(RED_F32_THREADS) void k_reduce_f32_partials(const float* __restrict__ partial, int nblocks, int n, double* __restrict__ totals) {
    __shared__ double red[RED_F32_THREADS];
    const int t = threadIdx.x, col = blockIdx.x * 32 + (t & 31), sl = t >> 5;
    const double s = col < n ? strided_sum(partial + col, n, sl, nblocks, RED_F32_THREADS / 32) : 0.0;
    red[t] = s;
    __syncthreads();
#pragma unroll
    for (int st = RED_F32_THREADS / 64; st > 0; st >>= 1) {
        if (sl < st) red[t] += red[t + 32 * st];
        __syncthreads();
    }
    if (sl == 0 && col < n) totals[col] = red[t];
}

__device__ inline void bn_finalize_channel(const BnFinalize& a, int c, double sum, double sumsq);
__global__ void k_bn_finalize(const BnFinalize a) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= a.cc) return;
    bn_finalize_channel(a, c, a.totals[c], a.totals[a.cc + c]);
}

__device__ inline void bn_finalize_channel(const BnFinalize& a, int c, double sum, double sumsq) {
    const double m = sum / (double)a.rows;
    double v = sumsq / (double)a.rows - m * m;
    v = v > 0.0 ? v : 0.0;
    const float is = (float)(1.0 / sqrt(v + (double)a.eps));
    const float sc = a.gamma[c] * is;
    a.mean[c] = (float)m;
    a.invstd[c] = is;
    a.scale[c] = sc;
    a.shift[c] = a.beta[c] - (float)m * sc;
    if (a.momentum >= 0.f && a.running_mean && a.running_var) {
        const double unb = a.rows > 1 ? v * (double)a.rows / (double)(a.rows - 1) : v;
        a.running_mean[c] = (1.f - a.momentum) * a.running_mean[c] + a.momentum * (float)m;
        a.running_var[c] = (1.f - a.momentum) * a.running_var[c] + a.momentum * (float)unb;
    }
}

__global__ __launch_bounds__(RED_F32_THREADS) void k_bn_reduce_finalize(const float* __restrict__ partial, int nblocks, const BnFinalize a) {
    __shared__ double red[RED_F32_THREADS];
    const int t = threadIdx.x, j = t & 31, sl = t >> 5;
    const int c = blockIdx.x * 16 + (j & 15);                     // channel; j >> 4: 0 = sum, 1 = sum of squares
    const int col = (j >> 4) * a.cc + c, n = 2 * a.cc;
    const double s = c < a.cc ? strided_sum(partial + col, n, sl, nblocks, RED_F32_THREADS / 32) : 0.0;
    red[t] = s;
    __syncthreads();
#pragma unroll
    for (int st = RED_F32_THREADS / 64; st > 0; st >>= 1) {
        if (sl < st) red[t] += red[t + 32 * st];
        __syncthreads();
    }
    if (t < 16 && c < a.cc) bn_finalize_channel(a, c, red[t], red[t + 16]);
}

// stage 2: fixed-order sum over the workgroup partials -> totals[NQ][C] (double, first NQ*C of the workspace tail)
template <int NQ>
__global__ __launch_bounds__(256) void k_reduce_partials(const double* __restrict__ partial, int nblocks, double* __restrict__ totals) {
    // one workgroup per 8 columns: thread (col = t & 7, slice = t >> 3) sums every 32nd partial in ascending order, the 32
    // slice sums are then added in a fixed tree -> the same bits on every run, and no single thread walks all partials
    __shared__ double red[256];
    const int t = threadIdx.x, col = blockIdx.x * 8 + (t & 7), sl = t >> 3;
    double s = 0.0;
    if (col < NQ * C)
        for (int b = sl; b < nblocks; b += 32) s += partial[(size_t)b * NQ * C + col];
    red[t] = s;
    __syncthreads();
#pragma unroll
    for (int st = 16; st > 0; st >>= 1) {
        if (sl < st) red[t] += red[t + 8 * st];
        __syncthreads();
    }
    if (sl == 0 && col < NQ * C) totals[col] = red[t];
}

__global__ __launch_bounds__(RED_THREADS) void k_colsum_partial(const float* __restrict__ x, long long rows, double* __restrict__ partial) {
    column_partials<1>(rows, partial, [&](long long r, int lane, f32x2 (&v)[1]) {
        v[0] = *reinterpret_cast<const f32x2*>(x + (size_t)r * C + 2 * lane);
    });
}

__global__ __launch_bounds__(RED_THREADS) void k_stats_partial(const float* __restrict__ x, long long rows, double* __restrict__ partial) {
    column_partials<2>(rows, partial, [&](long long r, int lane, f32x2 (&v)[2]) {
        v[0] = *reinterpret_cast<const f32x2*>(x + (size_t)r * C + 2 * lane);
        v[1] = v[0] * v[0];
    });
}

__global__ void k_colsum_final(const double* __restrict__ totals, float* __restrict__ out) {
    const int c = threadIdx.x;
    if (c < C) out[c] = (float)totals[c];
}

__global__ void k_stats_final(const double* __restrict__ totals, long long rows, float* __restrict__ mean, float* __restrict__ var) {
    const int c = threadIdx.x;
    if (c >= C) return;
    const double m = totals[c] / (double)rows;
    double v = totals[C + c] / (double)rows - m * m;
    mean[c] = (float)m;
    var[c] = (float)(v > 0.0 ? v : 0.0);
}

// ---- BN affine + dropout + ReLU + residual, forward ------------------------------------------------
// Streaming pass: every operand is read once (non-temporal loads keep them out of the caches' way) and four 16-B loads per
// stream are in flight per lane before the first use.
__global__ __launch_bounds__(256) void k_bn_act_fwd(const float* __restrict__ z, const float* __restrict__ scale,
                                                    const float* __restrict__ shift, const float* __restrict__ residual,
                                                    float* __restrict__ out, const ActArgs a_) {
    const ActArgs a = resolved(a_);
    const long long n4 = a.rows * (C / 4);
    const long long stride = (long long)gridDim.x * blockDim.x;
    // (stride is a multiple of 32 float4 = one row: a thread always works on the same 4 channels)
    const int c4 = (int)(((long long)blockIdx.x * blockDim.x + threadIdx.x) % (C / 4)) * 4;
    const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + c4);
    const f32x4 sh = *reinterpret_cast<const f32x4*>(shift + c4);
    for (long long i0 = (long long)blockIdx.x * blockDim.x + threadIdx.x; i0 < n4; i0 += 4 * stride) {
        f32x4 zz[4], rr[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long long i = i0 + u * stride;
            zz[u] = i < n4 ? ldnt4(z + i * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
            rr[u] = (residual && i < n4) ? ldnt4(residual + i * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long long i = i0 + u * stride;
            f32x4 v = zz[u] * sc + sh;
            if (a.p > 0.f) v *= keep_scale4(a.seed, (unsigned long long)i * 4, a.p, a.inv_keep);
            if (a.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            v += rr[u];
            if (i < n4) *reinterpret_cast<f32x4*>(out + i * 4) = v;
        }
    }
}

// ---- backward: g = dy * keep * relu'(v);  dbeta = sum g, dgamma = sum g * xhat ----------------------
__device__ inline f32x2 act_grad2(const float* __restrict__ dy, const float* __restrict__ z, const float* __restrict__ mean,
                                  const float* __restrict__ invstd, const float* __restrict__ gamma,
                                  const float* __restrict__ beta, const ActArgs& a, long long r, int lane, f32x2& xhat) {
    const size_t off = (size_t)r * C + 2 * lane;
    const f32x2 d = *reinterpret_cast<const f32x2*>(dy + off);
    const f32x2 zz = *reinterpret_cast<const f32x2*>(z + off);
    const f32x2 m = *reinterpret_cast<const f32x2*>(mean + 2 * lane);
    const f32x2 is = *reinterpret_cast<const f32x2*>(invstd + 2 * lane);
    const f32x2 ga = *reinterpret_cast<const f32x2*>(gamma + 2 * lane);
    const f32x2 be = *reinterpret_cast<const f32x2*>(beta + 2 * lane);
    xhat = (zz - m) * is;
    const f32x2 v = xhat * ga + be;
    f32x2 g = d;
    if (a.p > 0.f) g *= keep_scale2(a.seed, (unsigned long long)off, a.p, a.inv_keep);
    if (a.relu) { g.x = v.x > 0.f ? g.x : 0.f; g.y = v.y > 0.f ? g.y : 0.f; }
    return g;
}

// Sums pass of the BatchNorm backward: per-workgroup partials of sum g and sum g * xhat.  Paired-row accesses (one wave load =
// two 512-B rows, 16 B per lane), two of them in flight per stream; fp64 accumulators; the four waves and the two row
// parities are combined in a fixed order.
__global__ __launch_bounds__(RED_THREADS) void k_bn_bwd_partial(const float* __restrict__ dy, const float* __restrict__ z,
                                                                const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                double* __restrict__ partial, const ActArgs a_) {
    const ActArgs a = resolved(a_);
    __shared__ double s_red[8][2][C];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5, q = lane & 31;
    const int c4 = 4 * q;
    const f32x4 mn = *reinterpret_cast<const f32x4*>(mean + c4), is = *reinterpret_cast<const f32x4*>(invstd + c4);
    const f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + c4), be = *reinterpret_cast<const f32x4*>(beta + c4);
    double sg[4] = {0.0, 0.0, 0.0, 0.0}, sx[4] = {0.0, 0.0, 0.0, 0.0};
    const long long pairs = (a.rows + 1) / 2;
    const long long step = (long long)gridDim.x * 4;
    for (long long pr = (long long)blockIdx.x * 4 + wave; pr < pairs; pr += 2 * step) {
        f32x4 d[2], zz[2];
        long long r[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            r[u] = 2 * (pr + u * step) + h;
            const bool ok = r[u] < a.rows;
            const size_t off = (size_t)(ok ? r[u] : 0) * C + c4;
            d[u] = ok ? ldnt4(dy + off) : f32x4{0.f, 0.f, 0.f, 0.f};
            zz[u] = ldnt4(z + off);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            f32x4 g = d[u];
            if (a.p > 0.f) g *= keep_scale4(a.seed, (unsigned long long)r[u] * C + c4, a.p, a.inv_keep);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float xh = (zz[u][e] - mn[e]) * is[e];
                const float v = xh * ga[e] + be[e];
                float ge = g[e];
                if (a.relu) ge = v > 0.f ? ge : 0.f;
                sg[e] += (double)ge;
                sx[e] += (double)(ge * xh);
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) { s_red[2 * wave + h][0][c4 + e] = sg[e]; s_red[2 * wave + h][1][c4 + e] = sx[e]; }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * C; i += RED_THREADS) {
        const int qq = i / C, c = i % C;
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < 8; ++w) t += s_red[w][qq][c];
        partial[((size_t)blockIdx.x * 2 + qq) * C + c] = t;
    }
}

// The sums pass when the caller already holds the sums over rows [lo, lo + n_valid) of every frame (taken where dy was produced:
// eg_classifier_bwd_sums): totals = presum + the sums over the other rows of every frame (coordinate / connection nodes: a few
// rows per frame).  One workgroup: thread (channel c, row group t >> 7), fp64, fixed order.
__global__ __launch_bounds__(1024) void k_bn_bwd_presum(const double* __restrict__ presum, const float* __restrict__ dy,
                                                        const float* __restrict__ z, const float* __restrict__ mean,
                                                        const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, const RowMap m, int batch,
                                                        double* __restrict__ totals, const ActArgs a_, const float* __restrict__ taps,
                                                        float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ db) {
    const ActArgs a = resolved(a_);
    __shared__ double s_red[8][2][C];
    const int c = threadIdx.x & 127, grp = threadIdx.x >> 7;
    const int extra = m.stride - m.n_valid;                                    // rows per frame outside the range
    const long long n = (long long)batch * extra;
    const float mn = mean[c], is = invstd[c], ga = gamma[c], be = beta[c];
    double sg = 0.0, sx = 0.0;
#pragma unroll 4
    for (long long i = grp; i < n; i += 8) {
        const long long f = i / extra;
        const int e = (int)(i - f * extra);
        const size_t off = (size_t)(f * m.stride + (e < m.lo ? e : e + m.n_valid)) * C + c;
        float g = dy[off];
        if (a.p > 0.f) g *= keep_scale(a.seed, (unsigned long long)off, a.p, a.inv_keep);
        const float xh = (z[off] - mn) * is;
        const float v = xh * ga + be;
        if (a.relu) g = v > 0.f ? g : 0.f;
        sg += (double)g;
        sx += (double)(g * xh);
    }
    if (taps) {                                    // what the bilinear backward added inside the summed range afterwards: frames
        for (int f = grp; f < batch; f += 8) {     // f = grp, grp + 8, ... in ascending order per group, the groups in order below
            sg += (double)taps[((size_t)f * 2 + 0) * C + c];
            sx += (double)taps[((size_t)f * 2 + 1) * C + c];
        }
    }
    s_red[grp][0][c] = sg;
    s_red[grp][1][c] = sx;
    __syncthreads();
    if (grp < 2) {
        double t = presum[grp * C + c];
#pragma unroll
        for (int k = 0; k < 8; ++k) t += s_red[k][grp][c];
        totals[grp * C + c] = t;
        if (grp == 0) { dbeta[c] = (float)t; if (db) db[c] = 0.f; }     // (k_bn_bwd_final's job, and the bias gradient's memset, here)
        else dgamma[c] = (float)t;
    }
}

// ---- the lower layer's sums from the dX launch's per-tile partials (gcn_layer_ps.hip MODE 3): float [n_tiles][2][128] ->
// stage 1: one workgroup per (32 columns, chunk of tiles): double [chunks][256];  stage 2: chunks in ascending order, and
// sum g * z  ->  sum g * xhat = invstd * (sum g * z - mean * sum g).  Fixed order throughout.
constexpr int TILE_SUM_CHUNKS = 64;
__global__ __launch_bounds__(RED_F32_THREADS) void k_tile_sums_stage1(const float* __restrict__ partial, int n_tiles, int per,
                                                                      double* __restrict__ out) {
    __shared__ double red[RED_F32_THREADS];
    const int t = threadIdx.x, col = blockIdx.x * 32 + (t & 31), sl = t >> 5;
    const int lo = blockIdx.y * per, hi = lo + per < n_tiles ? lo + per : n_tiles;
    const double s = strided_sum(partial + col, 2 * C, lo + sl, hi, RED_F32_THREADS / 32);
    red[t] = s;
    __syncthreads();
#pragma unroll
    for (int st = RED_F32_THREADS / 64; st > 0; st >>= 1) {
        if (sl < st) red[t] += red[t + 32 * st];
        __syncthreads();
    }
    if (sl == 0) out[(size_t)blockIdx.y * 2 * C + col] = red[t];
}
__global__ __launch_bounds__(1024) void k_tile_sums_stage2(const double* __restrict__ chunks, int n_chunks, const float* __restrict__ mean,
                                                           const float* __restrict__ invstd, double* __restrict__ sums) {
    __shared__ double red[8][2][C];
    const int c = threadIdx.x & 127, sl = threadIdx.x >> 7;       // 8 slices of the chunk list, each in ascending order
    double s1 = 0.0, t1 = 0.0;
#pragma unroll 4
    for (int k = sl; k < n_chunks; k += 8) { s1 += chunks[(size_t)k * 2 * C + c]; t1 += chunks[(size_t)k * 2 * C + C + c]; }
    red[sl][0][c] = s1;
    red[sl][1][c] = t1;
    __syncthreads();
    if (sl == 0) {
        s1 = 0.0; t1 = 0.0;
#pragma unroll
        for (int k = 0; k < 8; ++k) { s1 += red[k][0][c]; t1 += red[k][1][c]; }
        sums[c] = s1;
        sums[C + c] = (double)invstd[c] * (t1 - (double)mean[c] * s1);
    }
}
// Up to TILE_SUMS_SMALL tiles (batch 1 - 3 at 224 / 7): ONE launch of C / 16 workgroups, each owning 16 channels x {sum g, sum g z}:
// thread (column, slice) sums every 32nd tile in ascending order, the 32 slices are added in a fixed tree, the sum g xhat transform
// follows in the same workgroup.  (Two stages for a list this short are two nodes of a captured batch-1 step, ~4.5 us each whatever
// they do; one stage for a long list is 8 workgroups pulling 37 MB at batch 32.)
constexpr int TILE_SUMS_SMALL = 4096;
__global__ __launch_bounds__(RED_F32_THREADS) void k_tile_sums_small(const float* __restrict__ partial, int n_tiles, const float* __restrict__ mean,
                                                                     const float* __restrict__ invstd, double* __restrict__ sums) {
    __shared__ double red[RED_F32_THREADS];
    const int t = threadIdx.x, j = t & 31, sl = t >> 5;
    const int c = blockIdx.x * 16 + (j & 15), col = (j >> 4) * C + c;          // j >> 4: 0 = sum g, 1 = sum g z
    const double s = strided_sum(partial + col, 2 * C, sl, n_tiles, RED_F32_THREADS / 32);
    red[t] = s;
    __syncthreads();
#pragma unroll
    for (int st = RED_F32_THREADS / 64; st > 0; st >>= 1) {
        if (sl < st) red[t] += red[t + 32 * st];
        __syncthreads();
    }
    if (t < 16) {
        const double s1 = red[t], t1 = red[t + 16];
        sums[c] = s1;
        sums[C + c] = (double)invstd[c] * (t1 - (double)mean[c] * s1);
    }
}

__global__ void k_bn_bwd_final(const double* __restrict__ totals, float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ db) {
    const int c = threadIdx.x;
    if (c < C) { dbeta[c] = (float)totals[c]; dgamma[c] = (float)totals[C + c]; if (db) db[c] = 0.f; }
}

__global__ __launch_bounds__(256) void k_bn_bwd_apply(const float* __restrict__ dy, const float* __restrict__ z,
                                                      const float* __restrict__ mean, const float* __restrict__ invstd,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      const double* __restrict__ totals, float* __restrict__ dz,
                                                      const ActArgs a_) {
    const ActArgs a = resolved(a_);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const double inv_n = 1.0 / (double)a.rows;
    const f32x2 mg = {(float)(totals[2 * lane] * inv_n), (float)(totals[2 * lane + 1] * inv_n)};             // mean of g
    const f32x2 mgx = {(float)(totals[C + 2 * lane] * inv_n), (float)(totals[C + 2 * lane + 1] * inv_n)};   // mean of g*xhat
    const f32x2 ga = *reinterpret_cast<const f32x2*>(gamma + 2 * lane);
    const f32x2 is = *reinterpret_cast<const f32x2*>(invstd + 2 * lane);
    for (long long r = (long long)blockIdx.x * 4 + wave; r < a.rows; r += (long long)gridDim.x * 4) {
        f32x2 xhat;
        const f32x2 g = act_grad2(dy, z, mean, invstd, gamma, beta, a, r, lane, xhat);
        const f32x2 o = ga * is * (g - mg - xhat * mgx);
        *reinterpret_cast<f32x2*>(dz + (size_t)r * C + 2 * lane) = o;
    }
}

// ---- dW = g^T x on the fp32 MFMA ---------------------------------------------------------------------
// A workgroup (4 waves) owns a contiguous chunk of rows and the whole 128x128 result: wave w computes the
// 32 x 128 slab dW[32w..32w+31][:] as four 32x32 accumulators.  Row tiles of 32 rows of g and x go through
// LDS; v_mfma_f32_32x32x2_f32: A[i][k] = g[row k][32w + i] (lane i = l&31, k = l>>5), B[k][j] = x[row k][32jb + j].
constexpr int DW_ROWS = 32;
constexpr int DW_BLOCKS = 768;            // 3 workgroups per CU (k_bn_bwd_apply_dw: 168 VGPRs)
// workspace layout: [0, WS_RED_BYTES) column-reduction partials (doubles) + totals; then DW_BLOCKS slabs of [128,128] floats
constexpr size_t WS_RED_BYTES = (((size_t)RED_BLOCKS * 2 * C + 2 * C) * sizeof(double) + 4095) / 4096 * 4096;

// xm.n_valid > 0: row r of g pairs with row map_row(xm, r) of x (x is the unfiltered array of a node-type filter)
__global__ __launch_bounds__(256) void k_dweight_partial(const float* __restrict__ g, const float* __restrict__ x, long long rows,
                                                         float* __restrict__ partial, const RowMap xm) {
    __shared__ __attribute__((aligned(16))) float s_g[DW_ROWS * C];
    __shared__ __attribute__((aligned(16))) float s_x[DW_ROWS * C];
    const int tid = threadIdx.x, lane = tid & 63, wave = wave_id();
    const int i = lane & 31, kh = lane >> 5;
    f32x16 acc[4];
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[b][e] = 0.f;
    const long long n_tiles = (rows + DW_ROWS - 1) / DW_ROWS;
    for (long long t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const long long r0 = t * DW_ROWS;
        // 32 rows x 128 floats = 1024 float4 per matrix; 256 threads -> 4 each
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int e = tid + 256 * q;                 // float4 index
            const long long r = r0 + e / 32;
            f32x4 vg = {0.f, 0.f, 0.f, 0.f}, vx = vg;
            if (r < rows) {
                const long long rx = xm.n_valid > 0 ? map_row(xm, r) : r;
                vg = *reinterpret_cast<const f32x4*>(g + (size_t)r * C + (e % 32) * 4);
                vx = *reinterpret_cast<const f32x4*>(x + (size_t)rx * C + (e % 32) * 4);
            }
            *reinterpret_cast<f32x4*>(&s_g[e * 4]) = vg;
            *reinterpret_cast<f32x4*>(&s_x[e * 4]) = vx;
        }
        __syncthreads();
#pragma unroll 4
        for (int s = 0; s < DW_ROWS / 2; ++s) {
            const int r = 2 * s + kh;
            const float a = s_g[r * C + 32 * wave + i];
#pragma unroll
            for (int jb = 0; jb < 4; ++jb)
                acc[jb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, s_x[r * C + 32 * jb + i], acc[jb], 0, 0, 0);
        }
        __syncthreads();
    }
    // D[m][n]: n = lane&31 -> input channel 32jb + n; m = (reg&3) + 8*(reg>>2) + 4*kh -> output channel 32w + m
    float* p = partial + (size_t)blockIdx.x * C * C;
#pragma unroll
    for (int jb = 0; jb < 4; ++jb)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int m = (e & 3) + 8 * (e >> 2) + 4 * kh;
            p[(size_t)(32 * wave + m) * C + 32 * jb + i] = acc[jb][e];
        }
}

// fixed-order sum of the per-workgroup slabs: workgroup = 32 elements x 8 slices of the slab list (launch with C*C/32 workgroups)
__global__ __launch_bounds__(RED_F32_THREADS) void k_dweight_final(const float* __restrict__ partial, int nblocks, float* __restrict__ dw, const ExtraReduce extra) {
    __shared__ double red[RED_F32_THREADS];
    const int t = threadIdx.x, sl = t >> 5;
    if (blockIdx.x >= C * C / 32) {                          // the extra column reduction: k_reduce_f32_partials' arithmetic
        const int col = (blockIdx.x - C * C / 32) * 32 + (t & 31);
        const double s = col < extra.n ? strided_sum(extra.partial + col, extra.n, sl, extra.nblocks, RED_F32_THREADS / 32) : 0.0;
        red[t] = s;
        __syncthreads();
#pragma unroll
        for (int st = RED_F32_THREADS / 64; st > 0; st >>= 1) {
            if (sl < st) red[t] += red[t + 32 * st];
            __syncthreads();
        }
        if (sl == 0 && col < extra.n) extra.totals[col] = red[t];
        return;
    }
    const int idx = blockIdx.x * 32 + (t & 31);
    const double s = strided_sum(partial + idx, C * C, sl, nblocks, RED_F32_THREADS / 32);
    red[t] = s;
    __syncthreads();
#pragma unroll
    for (int st = RED_F32_THREADS / 64; st > 0; st >>= 1) {
        if (sl < st) red[t] += red[t + 32 * st];
        __syncthreads();
    }
    if (sl == 0) dw[idx] = (float)red[t];
}

// ---- BN backward apply fused with the weight gradient: dz = BN'(dy * mask) is written AND fed (through LDS) to
// dW += dz^T x, so dz is not read back for the weight gradient.  Same tiling as k_dweight_partial.
typedef unsigned u32x4v __attribute__((ext_vector_type(4)));

// DIRECT (the arrays below 2 GB each; x only when it has no row map): the 12 row loads and 4 stores of a tile are buffer accesses off whole-array
// descriptors -- lane part in one VGPR per row group, tile offset in an SGPR, rows past the end read as zero / are dropped --
// so they cost no address arithmetic and no branches (vector instructions of the 3 co-resident workgroups serialise with the
// MFMAs of the SIMD).
template <bool DIRECT, bool MAPPED>
__global__ __launch_bounds__(256, 3) void k_bn_bwd_apply_dw(const float* __restrict__ dy, const float* __restrict__ z, const float* __restrict__ x,
                                                         const float* __restrict__ mean, const float* __restrict__ invstd,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         const double* __restrict__ totals, float* __restrict__ dz,
                                                         float* __restrict__ partial, const ActArgs a_, const RowMap xm) {
    const ActArgs a = resolved(a_);
    __shared__ __attribute__((aligned(16))) float s_g[DW_ROWS * C];
    __shared__ __attribute__((aligned(16))) float s_x[DW_ROWS * C];
    const int tid = threadIdx.x, lane = tid & 63, wave = wave_id();
    const int i = lane & 31, kh = lane >> 5;
    // every float4 this thread touches has the same 4 channels: c4 = 4 (tid % 32)
    const int c4 = (tid & 31) * 4;
    const double inv_n = 1.0 / (double)a.rows;
    const f32x4 mn = *reinterpret_cast<const f32x4*>(mean + c4), is = *reinterpret_cast<const f32x4*>(invstd + c4);
    const f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + c4), be = *reinterpret_cast<const f32x4*>(beta + c4);
    f32x4 mg, mgx;
#pragma unroll
    for (int u = 0; u < 4; ++u) { mg[u] = (float)(totals[c4 + u] * inv_n); mgx[u] = (float)(totals[C + c4 + u] * inv_n); }
    f32x16 acc[4];
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[b][e] = 0.f;
    const long long n_tiles = (a.rows + DW_ROWS - 1) / DW_ROWS;
    // The 12 row loads of a tile (dy, z, x: 4 float4 each per thread) are issued one tile ahead, right before the MFMA
    // phase of the current tile, so that the memory round trip runs under the 64 MFMAs instead of in front of them.
    f32x4 pd[4], pz[4], px[4];
    const int bytes = DIRECT ? (int)(a.rows * (C * 4)) : 0;
    const __amdgpu_buffer_rsrc_t rs_dy = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(dy), 0, bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_z = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(z), 0, bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, MAPPED ? 0 : bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_dz = __builtin_amdgcn_make_buffer_rsrc(dz, 0, dz ? bytes : 0, 0x00020000);
    int voff[4];                                           // row (tid >> 5) + 8 q of a tile, this thread's 4 channels
#pragma unroll
    for (int q = 0; q < 4; ++q) voff[q] = ((tid >> 5) + 8 * q) * (C * 4) + c4 * 4;
    auto issue = [&](long long t) {
        if (DIRECT) {
            const int soff = (int)(t * DW_ROWS) * (C * 4);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                pd[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_dy, voff[q], soff, 0));
                pz[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_z, voff[q], soff, 0));
            }
            if (MAPPED) {                                  // x through a row map (the heads' filtered rows): flat loads
                const long long r0 = t * DW_ROWS + (tid >> 5);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const long long r = r0 + 8 * q;
                    px[q] = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (r < a.rows) px[q] = *reinterpret_cast<const f32x4*>(x + (size_t)map_row(xm, r) * C + c4);
                }
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) px[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, voff[q], soff, 0));
            }
            return;
        }
        const long long r0 = t * DW_ROWS + (tid >> 5);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const long long r = r0 + 8 * q;
            pd[q] = pz[q] = px[q] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (r < a.rows) {
                const size_t off = (size_t)r * C + c4;
                pd[q] = *reinterpret_cast<const f32x4*>(dy + off);
                pz[q] = *reinterpret_cast<const f32x4*>(z + off);
                const long long rx = xm.n_valid > 0 ? map_row(xm, r) : r;
                px[q] = *reinterpret_cast<const f32x4*>(x + (size_t)rx * C + c4);
            }
        }
    };
    if ((long long)blockIdx.x < n_tiles) issue(blockIdx.x);
    for (long long t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const long long r0 = t * DW_ROWS;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int e = tid + 256 * q;
            const long long r = r0 + e / 32;
            f32x4 vg = {0.f, 0.f, 0.f, 0.f};
            if (DIRECT || r < a.rows) {                   // (DIRECT: rows past the end hold zeros, their store is dropped)
                // element index of this float4 (the dropout mask's counter): a uniform 64-bit tile base + the lane's 32-bit part --
                // kept as a per-lane 64-bit value for every q it cost four spilled registers that were re-loaded inside the loop,
                // each reload behind an s_waitcnt vmcnt(0) that also drained the next tile's prefetched rows
#ifdef EG_OLD_DW_OFF           // (A/B: the round-4 form with its spills)
                const size_t off = (size_t)r * C + c4;
#else
                const size_t off = DIRECT ? (size_t)r0 * C + (unsigned)(voff[q] >> 2) : (size_t)r * C + c4;
#endif
                const f32x4 zz = pz[q];
                f32x4 d = pd[q];
#ifdef EG_ABL_HASH_APPLY      // (timing-only: what would stored keep-bits be worth in this kernel?)
                d *= a.inv_keep; (void)off;
#else
                if (a.p > 0.f) d *= keep_scale4(a.seed, (unsigned long long)off, a.p, a.inv_keep);
#endif
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float xh = (zz[u] - mn[u]) * is[u];
                    const float v = xh * ga[u] + be[u];
                    float g = d[u];
                    if (a.relu) g = v > 0.f ? g : 0.f;
                    vg[u] = ga[u] * is[u] * (g - mg[u] - xh * mgx[u]);
                }
                if (DIRECT) {
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4v, vg), rs_dz, voff[q], (int)r0 * (C * 4), 0);
                    asm volatile("s_nop 1" :: "v"(vg.x), "v"(vg.y), "v"(vg.z), "v"(vg.w) : "memory");       // (store data: DESIGN 5.26)
                } else if (dz) {
                    *reinterpret_cast<f32x4*>(dz + off) = vg;               // (NULL: only the weight gradient is wanted)
                }
            }
            *reinterpret_cast<f32x4*>(&s_g[e * 4]) = vg;
            *reinterpret_cast<f32x4*>(&s_x[e * 4]) = px[q];
        }
        __syncthreads();
        if (t + gridDim.x < n_tiles) issue(t + gridDim.x);
#pragma unroll 4
        for (int s = 0; s < DW_ROWS / 2; ++s) {
            const int r = 2 * s + kh;
            const float av = s_g[r * C + 32 * wave + i];
#pragma unroll
            for (int jb = 0; jb < 4; ++jb)
                acc[jb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, s_x[r * C + 32 * jb + i], acc[jb], 0, 0, 0);
        }
        __syncthreads();
    }
    float* p = partial + (size_t)blockIdx.x * C * C;
#pragma unroll
    for (int jb = 0; jb < 4; ++jb)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int m = (e & 3) + 8 * (e >> 2) + 4 * kh;
            p[(size_t)(32 * wave + m) * C + 32 * jb + i] = acc[jb][e];
        }
}

static int red_blocks(long long rows) {
    long long b = (rows + 3) / 4;
    if (b > RED_BLOCKS) b = RED_BLOCKS;
    return (int)(b < 1 ? 1 : b);
}

}  // namespace eg

using namespace eg;

int eg_launch_bn_act_tiles(const eg_graph* g, int batch, const float* z, const float* scale, const float* shift, const float* residual,
                           int relu, float dropout_p, unsigned long long seed, float* out, float* kout, hipStream_t stream);

int eg_launch_dweight(const float* g, const float* x, long long rows, const eg::RowMap* xmap, void* workspace, float* dw, hipStream_t stream) {
    if (!g || !x || !workspace || !dw || rows < 1) return set_error(EG_ERR_ARG, "bad argument");
    long long nt = (rows + DW_ROWS - 1) / DW_ROWS;
    const int nb = (int)(nt < DW_BLOCKS ? nt : DW_BLOCKS);
    const RowMap xm = xmap ? *xmap : RowMap{0, 0, 0};
    float* slabs = (float*)((char*)workspace + WS_RED_BYTES);
    hipLaunchKernelGGL(k_dweight_partial, dim3(nb), dim3(256), 0, stream, g, x, rows, slabs, xm);
    hipLaunchKernelGGL(k_dweight_final, dim3(C * C / 32), dim3(RED_F32_THREADS), 0, stream, (const float*)slabs, nb, dw, ExtraReduce{});
    EG_HIP_TRY(hipGetLastError());
    return EG_OK;
}

// ---- dropout epoch: one 64-bit word per device that every mask-generating kernel adds to its seed (train_common.h) -----------
namespace eg {
__global__ void k_epoch_update(unsigned long long* __restrict__ e, unsigned long long v, int add) { *e = add ? *e + v : v; }

static std::mutex epoch_mu;
static unsigned long long* epoch_dev[64];

// tickets of last_workgroup_out (train_common.h): one zeroed word per (device, stream, slot)
static std::mutex ticket_mu;
static std::map<std::pair<int, void*>, unsigned*> ticket_dev;
unsigned* eg_ticket_ptr(void* stream, int slot) {
    int dev = 0;
    if (slot < 0 || slot >= EG_TICKET_SLOTS || hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lk(ticket_mu);
    unsigned*& p = ticket_dev[{dev, stream}];
    if (!p) {
        unsigned* q = nullptr;
        if (hipMalloc((void**)&q, EG_TICKET_SLOTS * sizeof(unsigned)) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        if (hipMemset(q, 0, EG_TICKET_SLOTS * sizeof(unsigned)) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(q); return nullptr; }
        p = q;
    }
    return p + slot;
}

const unsigned long long* eg_epoch_ptr() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    std::lock_guard<std::mutex> lk(epoch_mu);
    if (!epoch_dev[dev]) {                                   // (first train-mode launch on this device; never inside a stream capture:
        unsigned long long* p = nullptr;                     //  a warm-up step in front of the capture has been here)
        if (hipMalloc((void**)&p, sizeof(unsigned long long)) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        if (hipMemset(p, 0, sizeof(unsigned long long)) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(p); return nullptr; }
        epoch_dev[dev] = p;
    }
    return epoch_dev[dev];
}
int eg_epoch_required(float dropout_p) {
    if (dropout_p > 0.f && !eg_epoch_ptr())
        return set_error(EG_ERR_HIP, "the dropout epoch word could not be allocated (first train-mode launch of this device inside a stream capture?)");
    return EG_OK;
}
}  // namespace eg

static int epoch_update(uint64_t v, int add, eg_stream_t stream) {
    unsigned long long* e = const_cast<unsigned long long*>(eg_epoch_ptr());
    if (!e) return set_error(EG_ERR_HIP, "the dropout epoch word could not be allocated (first use inside a stream capture?)");
    hipLaunchKernelGGL(k_epoch_update, dim3(1), dim3(1), 0, (hipStream_t)stream, e, (unsigned long long)v, add);
    EG_HIP_TRY(hipGetLastError());
    return EG_OK;
}

extern "C" {
int eg_dropout_epoch_add(uint64_t delta, eg_stream_t stream) { return epoch_update(delta, 1, stream); }
int eg_dropout_epoch_set(uint64_t value, eg_stream_t stream) { return epoch_update(value, 0, stream); }
int eg_debug_dropout_epoch(uint64_t* out_host) {
    if (!out_host) return set_error(EG_ERR_ARG, "NULL argument");
    const unsigned long long* e = eg_epoch_ptr();
    if (!e) return set_error(EG_ERR_HIP, "the dropout epoch word could not be allocated");
    EG_HIP_TRY(hipDeviceSynchronize());
    unsigned long long v = 0;
    EG_HIP_TRY(hipMemcpy(&v, e, sizeof(v), hipMemcpyDeviceToHost));
    *out_host = v;
    return EG_OK;
}
}

static ActArgs make_act(int64_t rows, int relu, float p, uint64_t seed) {
    ActArgs a{};
    a.rows = rows; a.relu = relu; a.p = p; a.inv_keep = p > 0.f ? 1.0f / (1.0f - p) : 1.0f; a.seed = seed; a.epoch = eg_epoch_ptr();
    return a;
}

// BatchNorm backward of  y = relu?(dropout(BN_train(z)))  given dy, optionally fused with the weight gradient dW = dz^T x:
// sums pass -> (dgamma, dbeta) -> apply pass writing dz (and accumulating dW when x / dw are given).
int eg_launch_bn_bwd(const float* dy, const float* z, long long rows, const float* mean, const float* invstd, const float* gamma,
                     const float* beta, int relu, float dropout_p, unsigned long long seed, void* workspace, float* dz,
                     float* dgamma, float* dbeta, const float* x, const eg::RowMap* xmap, float* dw, hipStream_t stream,
                     const double* presum, const eg::RowMap* presum_rows, int presum_batch, const float* presum_taps, float* db_zero) {
    if (!dy || !z || !mean || !invstd || !gamma || !beta || !workspace || !dgamma || !dbeta || rows < 1)
        return set_error(EG_ERR_ARG, "bad argument");
    if (!dz && !(dw && x)) return set_error(EG_ERR_ARG, "dz may only be NULL when the fused weight gradient is computed");
    if (dropout_p < 0.f || dropout_p >= 1.f) return set_error(EG_ERR_ARG, "dropout_p must be in [0, 1)");
    if (int rc = eg_epoch_required(dropout_p)) return rc;
    const ActArgs a = make_act(rows, relu, dropout_p, seed);
    double* partial = (double*)workspace;
    const int nb = red_blocks(rows);
    double* totals = partial + (size_t)RED_BLOCKS * 2 * C;
    if (presum) {          // the sums over most rows exist: add the few rows they leave out
        hipLaunchKernelGGL(k_bn_bwd_presum, dim3(1), dim3(1024), 0, stream, presum, dy, z, mean, invstd, gamma, beta, *presum_rows,
                           presum_batch, totals, a, presum_taps, dgamma, dbeta, db_zero);
    } else {
        hipLaunchKernelGGL(k_bn_bwd_partial, dim3(nb), dim3(RED_THREADS), 0, stream, dy, z, mean, invstd, gamma, beta, partial, a);
        hipLaunchKernelGGL(k_reduce_partials<2>, dim3(2 * C / 8), dim3(256), 0, stream, partial, nb, totals);
        hipLaunchKernelGGL(k_bn_bwd_final, dim3(1), dim3(128), 0, stream, totals, dgamma, dbeta, db_zero);
    }
    if (dw && x) {
        long long nt = (rows + DW_ROWS - 1) / DW_ROWS;
        const int nd = (int)(nt < DW_BLOCKS ? nt : DW_BLOCKS);
        const RowMap xm = xmap ? *xmap : RowMap{0, 0, 0};
        float* slabs = (float*)((char*)workspace + WS_RED_BYTES);
        const bool direct = rows * (long long)(C * 4) < (1ll << 31);
        auto launch = [&](auto kernel) {
            hipLaunchKernelGGL(kernel, dim3(nd), dim3(256), 0, stream, dy, z, x, mean, invstd, gamma, beta, totals, dz, slabs, a, xm);
        };
        // with a row map (the heads' filtered rows) the flat-address form is the faster one (1.12 vs 1.35 ms at B = 32)
        if (direct && xm.n_valid == 0) launch(k_bn_bwd_apply_dw<true, false>);
        else launch(k_bn_bwd_apply_dw<false, true>);                   // (the generic form reads the map at run time)
        hipLaunchKernelGGL(k_dweight_final, dim3(C * C / 32), dim3(RED_F32_THREADS), 0, stream, (const float*)slabs, nd, dw, ExtraReduce{});
    } else {
        long long blocks = (rows + 3) / 4;
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(k_bn_bwd_apply, dim3((unsigned)blocks), dim3(256), 0, stream, dy, z, mean, invstd, gamma, beta, totals, dz, a);
    }
    EG_HIP_TRY(hipGetLastError());
    return EG_OK;
}

extern "C" {

// doubles needed by the two-stage reductions: RED_BLOCKS * 2 * 128 partials + 2 * 128 totals; dW needs
// DW_BLOCKS * 128 * 128 floats.  One buffer of eg_workspace_bytes() serves every call below.
size_t eg_workspace_bytes(void) { return WS_RED_BYTES + (size_t)DW_BLOCKS * C * C * sizeof(float); }

int eg_colsum128(const float* x, int64_t rows, void* workspace, float* out, eg_stream_t stream_) {
    if (!x || !workspace || !out || rows < 1) return set_error(EG_ERR_ARG, "bad argument");
    hipStream_t stream = (hipStream_t)stream_;
    double* partial = (double*)workspace;
    const int nb = red_blocks(rows);
    double* totals = partial + (size_t)RED_BLOCKS * 2 * C;
    hipLaunchKernelGGL(k_colsum_partial, dim3(nb), dim3(RED_THREADS), 0, stream, x, (long long)rows, partial);
    hipLaunchKernelGGL(k_reduce_partials<1>, dim3(C / 8), dim3(256), 0, stream, partial, nb, totals);
    hipLaunchKernelGGL(k_colsum_final, dim3(1), dim3(128), 0, stream, totals, out);
    EG_HIP_TRY(hipGetLastError());
    return EG_OK;
}

int eg_bn_stats(const float* x, int64_t rows, void* workspace, float* mean, float* var, eg_stream_t stream_) {
    if (!x || !workspace || !mean || !var || rows < 1) return set_error(EG_ERR_ARG, "bad argument");
    hipStream_t stream = (hipStream_t)stream_;
    double* partial = (double*)workspace;
    const int nb = red_blocks(rows);
    double* totals = partial + (size_t)RED_BLOCKS * 2 * C;
    hipLaunchKernelGGL(k_stats_partial, dim3(nb), dim3(RED_THREADS), 0, stream, x, (long long)rows, partial);
    hipLaunchKernelGGL(k_reduce_partials<2>, dim3(2 * C / 8), dim3(256), 0, stream, partial, nb, totals);
    hipLaunchKernelGGL(k_stats_final, dim3(1), dim3(128), 0, stream, totals, (long long)rows, mean, var);
    EG_HIP_TRY(hipGetLastError());
    return EG_OK;
}

int eg_bn_act_fwd(const float* z, int64_t rows, const float* scale, const float* shift, const float* residual,
                  int relu, float dropout_p, uint64_t seed, float* out, eg_stream_t stream) {
    if (!z || !scale || !shift || !out || rows < 1) return set_error(EG_ERR_ARG, "bad argument");
    if (dropout_p < 0.f || dropout_p >= 1.f) return set_error(EG_ERR_ARG, "dropout_p must be in [0, 1)");
    if (int rc = eg_epoch_required(dropout_p)) return rc;
    const ActArgs a = make_act(rows, relu, dropout_p, seed);
    long long blocks = (rows * (C / 4) + 255) / 256;
    if (blocks > 65536) blocks = 65536;        // (measured at batch 32: 4.6 - 4.9 TB/s with 1k - 16k blocks, 5.2 TB/s with 64k)
    hipLaunchKernelGGL(k_bn_act_fwd, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, z, scale, shift, residual, out, a);
    EG_HIP_TRY(hipGetLastError());
    return EG_OK;
}

int eg_bn_act_bwd(const float* dy, const float* z, int64_t rows, const float* mean, const float* invstd,
                  const float* gamma, const float* beta, int relu, float dropout_p, uint64_t seed, void* workspace,
                  float* dz, float* dgamma, float* dbeta, eg_stream_t stream) {
    return eg_launch_bn_bwd(dy, z, rows, mean, invstd, gamma, beta, relu, dropout_p, seed, workspace, dz, dgamma, dbeta, nullptr,
                            nullptr, nullptr, (hipStream_t)stream, nullptr, nullptr, 0, nullptr, nullptr);
}

int eg_dweight128(const float* g, const float* x, int64_t rows, void* workspace, float* dw, eg_stream_t stream) {
    return eg_launch_dweight(g, x, rows, nullptr, workspace, dw, (hipStream_t)stream);
}

// ---- one whole train-mode GNN layer: forward and backward (reference src/core/models.py:328-335, :431-435) --------------
//   forward   z = A_hat x W^T + b (the aggregated tile A_hat x is kept: `agg`), batch statistics of z, running-stat update,
//             out = relu?(dropout(BN(z))) + (residual ? x : 0)
//   backward  dz = BN'(dy * mask), dgamma, dbeta;  dX = (A_hat^T dz) W + (residual ? dy : 0);  dW = dz^T (A_hat x);  db = 0
int eg_gcn_layer_train_fwd(const eg_graph* g, int batch, const float* x, const float* W, const float* bias, const float* gamma,
                           const float* beta, float* running_mean, float* running_var, float momentum, float eps, int relu,
                           float dropout_p, uint64_t seed, int residual, void* workspace, float* z, float* agg, float* bn,
                           float* out, const float* kidsum_in, float* kidsum_out, eg_stream_t stream_) {
    if (!g || !x || !W || !gamma || !beta || !workspace || !z || !bn) return set_error(EG_ERR_ARG, "NULL argument");
    if (!out && kidsum_out) return set_error(EG_ERR_ARG, "kidsum_out are the child sums of out: out must not be NULL");
    if (dropout_p < 0.f || dropout_p >= 1.f) return set_error(EG_ERR_ARG, "dropout_p must be in [0, 1)");
    if (int rc = eg_epoch_required(dropout_p)) return rc;
    if (batch < 1) return set_error(EG_ERR_ARG, "batch must be >= 1");
    if ((kidsum_in || kidsum_out) && (g->kind != GRAPH_TOPO || g->kid_rows == 0))
        return set_error(EG_ERR_UNSUPPORTED, "child sums need a topology handle with eg_graph_kidsum_rows() > 0");
    if (kidsum_in && (kidsum_in == kidsum_out || kidsum_in == x || kidsum_in == z || kidsum_in == agg || kidsum_in == out))
        return set_error(EG_ERR_ARG, "kidsum_in must not alias kidsum_out or a row array");
    if (kidsum_out && (kidsum_out == x || kidsum_out == z || kidsum_out == agg || kidsum_out == out))
        return set_error(EG_ERR_ARG, "kidsum_out must not alias a row array");
    hipStream_t stream = (hipStream_t)stream_;
    // the layer kernel's epilogue leaves the column sums of z and z^2 per workgroup: no separate statistics pass
    float* partial = (float*)workspace;
    double* totals = (double*)workspace + (size_t)RED_BLOCKS * 2 * C;
    int grid = 0;
    // implicit topologies: the producer / consumer kernel in its train form (static tile walk: bit-reproducible partial sums);
    // anything else, or EG_TRAIN_PS=0: the symmetric kernel
    static const bool train_ps = !(getenv("EG_TRAIN_PS") && atoi(getenv("EG_TRAIN_PS")) == 0);
    int rc = EG_ERR_UNSUPPORTED;
    if (train_ps)
        rc = eg_launch_layer_ps(g, batch, x, W, nullptr, bias, nullptr, 0, 0, z, kidsum_in, nullptr, nullptr, stream, nullptr, nullptr,
                                agg, partial, &grid);
    // (the symmetric kernel pulls the children as rows: it needs no child sums, kidsum_in is simply not used)
    if (rc == EG_ERR_UNSUPPORTED) rc = eg_launch_layer_sym(g, batch, x, W, nullptr, bias, nullptr, 0, 0, z, agg, partial, &grid, stream);
    if (rc != EG_OK) return public_rc(rc);
    const long long rows = (long long)g->n_nodes * batch;
    BnFinalize f{totals, rows, C, gamma, beta, eps, momentum, running_mean, running_var, bn, bn + C, bn + 2 * C, bn + 3 * C};
    hipLaunchKernelGGL(k_bn_reduce_finalize, dim3(C / 16), dim3(RED_F32_THREADS), 0, stream, (const float*)partial, grid, f);
    EG_HIP_TRY(hipGetLastError());
    if (!out) return EG_OK;                       // z, agg and the batch statistics only
    // the activation pass in tile order leaves the child sums of `out` behind for the next layer's train forward
    if (kidsum_out) {
        rc = eg_launch_bn_act_tiles(g, batch, z, bn + 2 * C, bn + 3 * C, residual ? x : nullptr, relu, dropout_p, seed, out, kidsum_out, stream);
        if (rc != EG_ERR_UNSUPPORTED) return rc;
        if (kidsum_out) return set_error(EG_ERR_UNSUPPORTED, "child sums of the output: frames of 2 GB and more are not covered");
    }
    return eg_bn_act_fwd(z, rows, bn + 2 * C, bn + 3 * C, residual ? x : nullptr, relu, dropout_p, seed, out, stream_);
}

static int gcn_layer_bwd(const eg_graph* g_bwd, int batch, const float* dy, const float* z, const float* agg, const float* W,
                         const float* gamma, const float* beta, const float* bn, int relu, float dropout_p, uint64_t seed,
                         int residual, void* workspace, float* dz_scratch, float* dx, float* dw, float* db, float* dgamma,
                         float* dbeta, const double* presum, const RowMap* presum_rows, int presum_frames, eg_stream_t stream_,
                         const float* presum_taps = nullptr, const eg_lower_sums* lower = nullptr) {
    if (!g_bwd || !dy || !z || !W || !gamma || !beta || !bn || !workspace || !dgamma || !dbeta)
        return set_error(EG_ERR_ARG, "NULL argument");
    static const bool train_ps_on = !(getenv("EG_TRAIN_PS") && atoi(getenv("EG_TRAIN_PS")) == 0);
    if (lower) {
        // decided before anything is launched: the by-product exists only on the producer / consumer kernel's dX launch
        if (!lower->z || !lower->bn || !lower->tile_scratch || !lower->sums_out || lower->row_hi < 1 || lower->row_hi > g_bwd->n_nodes)
            return set_error(EG_ERR_ARG, "incomplete eg_lower_sums");
        if (lower->dropout_p < 0.f || lower->dropout_p >= 1.f) return set_error(EG_ERR_ARG, "dropout_p must be in [0, 1)");
        if (!dx || !residual || !train_ps_on || g_bwd->kind != GRAPH_TOPO) return EG_ERR_UNSUPPORTED;
        if (int rc = eg_epoch_required(lower->dropout_p)) return rc;
    }
    if (!dz_scratch && (dx || !dw)) return set_error(EG_ERR_ARG, "dz_scratch may only be NULL when dx is not wanted and dw is");
    if (dw && !agg) return set_error(EG_ERR_ARG, "dW needs the aggregated input kept by eg_gcn_layer_train_fwd");
    hipStream_t stream = (hipStream_t)stream_;
    const long long rows = (long long)g_bwd->n_nodes * batch;
    int rc = eg_launch_bn_bwd(dy, z, rows, bn, bn + C, gamma, beta, relu, dropout_p, seed, workspace, dz_scratch, dgamma, dbeta,
                              dw ? agg : nullptr, nullptr, dw, stream, presum, presum_rows, presum_frames, presum_taps, db);
    if (rc != EG_OK) return rc;
    if (dx) {
        // dX = (A_hat dz) W + dy: the producer / consumer kernel with the residual as a tensor of its own (implicit topologies)
        const bool train_ps = train_ps_on;
        rc = EG_ERR_UNSUPPORTED;
        if (lower) {
            const LowerSums ls{lower->z, lower->bn + 2 * C, lower->bn + 3 * C, lower->relu, lower->dropout_p,
                               lower->dropout_p > 0.f ? 1.0f / (1.0f - lower->dropout_p) : 1.0f, (unsigned long long)lower->seed,
                               eg_epoch_ptr(), (int)lower->row_hi, lower->tile_scratch};
            rc = eg_launch_layer_ps(g_bwd, batch, dz_scratch, W, nullptr, nullptr, dy, 0, 1, dx, nullptr, nullptr, nullptr, stream,
                                    nullptr, nullptr, nullptr, nullptr, nullptr, &ls);
            if (rc == EG_ERR_UNSUPPORTED) return set_error(EG_ERR_HIP, "the dX launch with the lower layer's sums was refused after the layer's own passes had run");
            if (rc != EG_OK) return public_rc(rc);
            const int n_tiles = g_bwd->n_tiles * batch;
            const int chunks = n_tiles < TILE_SUM_CHUNKS ? n_tiles : TILE_SUM_CHUNKS;
            const int per = (n_tiles + chunks - 1) / chunks;
            double* chunk_sums = (double*)workspace;             // (the reduction area of the workspace: this layer's own passes are done with it)
            if (n_tiles <= TILE_SUMS_SMALL) {
                hipLaunchKernelGGL(k_tile_sums_small, dim3(C / 16), dim3(RED_F32_THREADS), 0, stream, (const float*)lower->tile_scratch, n_tiles,
                                   lower->bn, lower->bn + C, lower->sums_out);
            } else {
                hipLaunchKernelGGL(k_tile_sums_stage1, dim3(2 * C / 32, chunks), dim3(RED_F32_THREADS), 0, stream, (const float*)lower->tile_scratch,
                                   n_tiles, per, chunk_sums);
                hipLaunchKernelGGL(k_tile_sums_stage2, dim3(1), dim3(1024), 0, stream, (const double*)chunk_sums, chunks, lower->bn, lower->bn + C,
                                   lower->sums_out);
            }
            EG_HIP_TRY(hipGetLastError());
        } else if (train_ps && residual)
            rc = eg_launch_layer_ps(g_bwd, batch, dz_scratch, W, nullptr, nullptr, dy, 0, 1, dx, nullptr, nullptr, nullptr, stream);
        if (rc == EG_ERR_UNSUPPORTED && !lower)
            rc = eg_launch_layer_sym(g_bwd, batch, dz_scratch, W, nullptr, nullptr, residual ? dy : nullptr, 0, 1, dx, nullptr, nullptr,
                                     nullptr, stream);
        if (rc != EG_OK) return public_rc(rc);
    }
    // (db = 0 -- a bias in front of a train-mode BatchNorm -- is written by the sums' final kernel: no memset node)
    return EG_OK;
}

int eg_gcn_layer_bwd(const eg_graph* g_bwd, int batch, const float* dy, const float* z, const float* agg, const float* W,
                     const float* gamma, const float* beta, const float* bn, int relu, float dropout_p, uint64_t seed,
                     int residual, void* workspace, float* dz_scratch, float* dx, float* dw, float* db, float* dgamma,
                     float* dbeta, eg_stream_t stream) {
    return gcn_layer_bwd(g_bwd, batch, dy, z, agg, W, gamma, beta, bn, relu, dropout_p, seed, residual, workspace, dz_scratch, dx, dw,
                         db, dgamma, dbeta, nullptr, nullptr, 0, stream);
}

int eg_gcn_layer_bwd_presummed(const eg_graph* g_bwd, int batch, const float* dy, const float* z, const float* agg, const float* W,
                               const float* gamma, const float* beta, const float* bn, int relu, float dropout_p, uint64_t seed,
                               int residual, void* workspace, float* dz_scratch, float* dx, float* dw, float* db, float* dgamma,
                               float* dbeta, const double* dy_sums, int frames, int64_t row_lo, int64_t n_valid, eg_stream_t stream) {
    if (!g_bwd || !dy_sums) return set_error(EG_ERR_ARG, "NULL argument");
    if (frames < 1 || (long long)g_bwd->n_nodes * batch % frames != 0) return set_error(EG_ERR_ARG, "frames must divide the layer's rows");
    const long long n_per_frame = (long long)g_bwd->n_nodes * batch / frames;
    if (row_lo < 0 || n_valid < 1 || row_lo + n_valid > n_per_frame) return set_error(EG_ERR_ARG, "bad row range");
    const RowMap m{(int)n_valid, (int)n_per_frame, (int)row_lo};
    // (frames, not `batch`, counts the row ranges: a CSR handle of a whole batch has batch == 1)
    return gcn_layer_bwd(g_bwd, batch, dy, z, agg, W, gamma, beta, bn, relu, dropout_p, seed, residual, workspace, dz_scratch, dx, dw,
                         db, dgamma, dbeta, dy_sums, &m, frames, stream);
}


int eg_gcn_layer_bwd_lower(const eg_graph* g_bwd, int batch, const float* dy, const float* z, const float* agg, const float* W,
                           const float* gamma, const float* beta, const float* bn, int relu, float dropout_p, uint64_t seed,
                           int residual, void* workspace, float* dz_scratch, float* dx, float* dw, float* db, float* dgamma,
                           float* dbeta, const eg_given_sums* given, const eg_lower_sums* lower, eg_stream_t stream) {
    if (!g_bwd) return set_error(EG_ERR_ARG, "NULL argument");
    RowMap m{0, 0, 0};
    if (given) {
        if (!given->sums || given->frames < 1 || (long long)g_bwd->n_nodes * batch % given->frames != 0)
            return set_error(EG_ERR_ARG, "given sums: frames must divide the layer's rows");
        const long long n_per_frame = (long long)g_bwd->n_nodes * batch / given->frames;
        if (given->row_lo < 0 || given->n_valid < 1 || given->row_lo + given->n_valid > n_per_frame) return set_error(EG_ERR_ARG, "bad row range");
        m = RowMap{(int)given->n_valid, (int)n_per_frame, (int)given->row_lo};
    }
    return gcn_layer_bwd(g_bwd, batch, dy, z, agg, W, gamma, beta, bn, relu, dropout_p, seed, residual, workspace, dz_scratch, dx, dw,
                         db, dgamma, dbeta, given ? given->sums : nullptr, given ? &m : nullptr, given ? given->frames : 0, stream,
                         given ? given->taps : nullptr, lower);
}

}  // extern "C"
