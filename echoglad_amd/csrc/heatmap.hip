// Losses on the logits and landmark decode for the hierarchical heat maps (gfx950).
// Replaces, on the device and without host round trips:
//   src/core/criterion.py:13-27     WeightedBCEWithLogitsLoss (numpy round trip for the weights, :18-21)
//   src/core/criterion.py:93-151    ExpectedLandmarkMSE: per level softmax over the nodes -> expected (h, w), ground
//                                   truth (h, w) from the label heat map, mean of `valid` per (frame, level, channel)
//   src/core/evaluators.py:291-391  the same decode on the last F*F rows (+ hard argmax, models' landmark index)
//
// The logits of a frame are [n_rows, 4] with the levels stacked row-major (2x2, 4x4, ..., FxF).  All per-level
// statistics of one (frame, level, channel) come out of ONE pass structure:
//   k_hm_partial  one workgroup per 1024-row chunk of a level: chunk max, first arg max, sum exp(x - max),
//                 sum exp * h, sum exp * w (fp64 accumulators, fixed tree order), label max / min h / min w of
//                 the label maxima, sum of valid
//   k_hm_final    merges the chunks of a level in order (softmax merge in fp64)
// so the result is bitwise reproducible and independent of the grid.  Bandwidth is trivial (16 B per node).
#include "train_common.h"

namespace eg {

constexpr int HM_MAX_LEVELS = 16;
constexpr int HM_CHUNK = 1024;
constexpr int HM_THREADS = 256;
constexpr int HM_RPT = HM_CHUNK / HM_THREADS;      // rows per thread
constexpr int HM_REC = 10;          // doubles per (frame, chunk, channel): m, s, sh, sw, best, bidx, g, gh, gw, vsum

struct HmLevels {
    int n_levels, n_rows, batch, total_chunks;
    int start[HM_MAX_LEVELS], side[HM_MAX_LEVELS], chunk0[HM_MAX_LEVELS + 1];
};

__device__ inline double block_sum(double v, double* red, int tid) {
    __syncthreads();
    red[tid] = v;
    __syncthreads();
#pragma unroll
    for (int s = HM_THREADS / 2; s > 0; s >>= 1) {
        if (tid < s) red[tid] += red[tid + s];
        __syncthreads();
    }
    return red[0];
}

// ---- wave-level merges (64 lanes, xor tree: a fixed order, so the same bits on every run); a workgroup combines its 4 waves through
// one LDS hand-over.  (The first version ran every reduction as an 8-step LDS tree with a barrier per step: ~200 barriers per
// workgroup, 31 us for a pass over 1.2 MB at batch 1 -- all of it barrier latency.)
__device__ inline double wave_sum(double v) {
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) v += __shfl_xor(v, s);
    return v;
}
// (by value, with selects: references into the per-channel arrays and branches around the updates kept those arrays in scratch memory --
//  and a dispatch that needs scratch waits for its set-up: 30 us for a 5-us kernel at batch 1)
struct ArgMax { float v; int i; };
struct GtMax { float g; int h, w; };
__device__ inline ArgMax wave_argmax(ArgMax a) {                   // max, LOWEST index among equal maxima
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) {
        const float o = __shfl_xor(a.v, s); const int oi = __shfl_xor(a.i, s);
        const bool take = o > a.v || (o == a.v && oi < a.i);
        a.v = take ? o : a.v; a.i = take ? oi : a.i;
    }
    return a;
}
__device__ inline GtMax wave_gtmax(GtMax a) {                      // max, min h and min w over all positions that hold it
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) {
        const float o = __shfl_xor(a.g, s); const int oh = __shfl_xor(a.h, s), ow = __shfl_xor(a.w, s);
        const bool gt = o > a.g, eq = o == a.g;
        a.h = gt ? oh : (eq ? min(a.h, oh) : a.h);
        a.w = gt ? ow : (eq ? min(a.w, ow) : a.w);
        a.g = gt ? o : a.g;
    }
    return a;
}
// ... N of them at once: the shuffles of one step are issued back to back (each is an LDS round trip of ~120 cycles; reduced one after
// the other, 26 reductions x 6 steps were 10 us of dependent latency in a workgroup that lives 19 us)
template <int N>
__device__ inline void wave_sum_n(double (&v)[N]) {
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) {
        double o[N];
#pragma unroll
        for (int k = 0; k < N; ++k) o[k] = __shfl_xor(v[k], s);
#pragma unroll
        for (int k = 0; k < N; ++k) v[k] += o[k];
    }
}
__device__ inline void wave_argmax4(float (&v)[4], int (&idx)[4]) {
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) {
        float o[4]; int oi[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) { o[c] = __shfl_xor(v[c], s); oi[c] = __shfl_xor(idx[c], s); }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const bool take = o[c] > v[c] || (o[c] == v[c] && oi[c] < idx[c]);
            v[c] = take ? o[c] : v[c]; idx[c] = take ? oi[c] : idx[c];
        }
    }
}
__device__ inline void wave_gtmax4(float (&g)[4], int (&gh)[4], int (&gw)[4]) {
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) {
        float o[4]; int oh[4], ow[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) { o[c] = __shfl_xor(g[c], s); oh[c] = __shfl_xor(gh[c], s); ow[c] = __shfl_xor(gw[c], s); }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const bool gt = o[c] > g[c], eq = o[c] == g[c];
            gh[c] = gt ? oh[c] : (eq ? min(gh[c], oh[c]) : gh[c]);
            gw[c] = gt ? ow[c] : (eq ? min(gw[c], ow[c]) : gw[c]);
            g[c] = gt ? o[c] : g[c];
        }
    }
}

__device__ inline float bce_logits(float x, float y) { return fmaxf(x, 0.f) - x * y + log1pf(expf(-fabsf(x))); }

// BCE: the levels tile the frame's rows (the training step's criteria: eg_criteria_fwd checks it), so this pass sees every logit, label
// and valid flag once anyway -- the weighted BCE's partial sums (k_bce_partial's arithmetic per element) come out of it as well, one
// (sum, sum valid) pair per workgroup at bce_part[2 * blockIdx.x]: no pass of its own over the same three arrays.
template <bool BCE>
__global__ __launch_bounds__(HM_THREADS) void k_hm_partial(const float* __restrict__ logits, const float* __restrict__ y,
                                                           const float* __restrict__ valid, double* __restrict__ part,
                                                           const HmLevels L, float ones_weight, double* __restrict__ bce_part) {
    constexpr int NW = HM_THREADS / 64;
    __shared__ float s_m[NW][4], s_g[NW][4];
    __shared__ int s_bi[NW][4], s_gh[NW][4], s_gw[NW][4];
    __shared__ double s_vs[NW][4], s_s[NW][4][3], s_b[NW][2];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int b = blockIdx.x / L.total_chunks, chunk = blockIdx.x - b * L.total_chunks;
    // (the level's entries by static indices: a run-time index into the by-value table puts the table into scratch memory, and a
    //  dispatch that needs scratch waits for its set-up)
    int side = L.side[0], start = L.start[0], c0 = L.chunk0[0];
#pragma unroll
    for (int k = 1; k < HM_MAX_LEVELS; ++k)
        if (k < L.n_levels && chunk >= L.chunk0[k]) { side = L.side[k]; start = L.start[k]; c0 = L.chunk0[k]; }
    const int n_lvl = side * side;
    const int r0 = (chunk - c0) * HM_CHUNK, r1 = min(r0 + HM_CHUNK, n_lvl);
    const size_t base = ((size_t)b * L.n_rows + start) * 4;
    const float NEG = -__builtin_inff();

    float m[4] = {NEG, NEG, NEG, NEG}, g[4] = {NEG, NEG, NEG, NEG};
    int bi[4] = {0x7fffffff, 0x7fffffff, 0x7fffffff, 0x7fffffff};
    int gh[4] = {0x7fffffff, 0x7fffffff, 0x7fffffff, 0x7fffffff}, gw[4] = {0x7fffffff, 0x7fffffff, 0x7fffffff, 0x7fffffff};
    double vs[4] = {0, 0, 0, 0}, bce_a = 0, bce_v = 0;
    // the thread's HM_RPT rows: every load of the kernel is issued here, before the first use (a loop that loads, waits and updates row
    // after row is HM_RPT dependent memory round trips -- 2/3 of this kernel's 30 us at batch 1 with 8 rows per thread and a second
    // pass that loaded the logits again); the logits stay in registers for the sums below
    f32x4 xs[HM_RPT], ys[HM_RPT], vv[HM_RPT];
#pragma unroll
    for (int i = 0; i < HM_RPT; ++i) {
        const int r = r0 + tid + i * HM_THREADS;
        const size_t o = base + (size_t)(r < r1 ? r : r0) * 4;
        xs[i] = *reinterpret_cast<const f32x4*>(logits + o);
        ys[i] = y ? *reinterpret_cast<const f32x4*>(y + o) : f32x4{0.f, 0.f, 0.f, 0.f};
        vv[i] = valid ? *reinterpret_cast<const f32x4*>(valid + o) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int i = 0; i < HM_RPT; ++i) {
        const int r = r0 + tid + i * HM_THREADS;
        if (r >= r1) continue;
        const float xv[4] = {xs[i].x, xs[i].y, xs[i].z, xs[i].w};
#pragma unroll
        for (int c = 0; c < 4; ++c) { const bool take = xv[c] > m[c]; m[c] = take ? xv[c] : m[c]; bi[c] = take ? r : bi[c]; }   // rows ascend: first max kept
        if (y) {
            const float tv[4] = {ys[i].x, ys[i].y, ys[i].z, ys[i].w};
            const int h = r / side, w = r - h * side;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const bool gt = tv[c] > g[c], eq = tv[c] == g[c];
                gh[c] = gt ? h : (eq ? min(gh[c], h) : gh[c]);
                gw[c] = gt ? w : (eq ? min(gw[c], w) : gw[c]);
                g[c] = gt ? tv[c] : g[c];
            }
        }
        if (valid) { vs[0] += vv[i].x; vs[1] += vv[i].y; vs[2] += vv[i].z; vs[3] += vv[i].w; }
        if (BCE) {
            const float tv[4] = {ys[i].x, ys[i].y, ys[i].z, ys[i].w}, va[4] = {vv[i].x, vv[i].y, vv[i].z, vv[i].w};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float wgt = (ones_weight > 1.0f && tv[c] == 1.0f) ? ones_weight : 1.0f;
                bce_a += (double)(wgt * bce_logits(xv[c], tv[c])) * va[c];
                bce_v += va[c];
            }
        }
    }
    wave_argmax4(m, bi);
    if (y) wave_gtmax4(g, gh, gw);
    if (valid) wave_sum_n<4>(vs);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        if (lane == 0) { s_m[wv][c] = m[c]; s_bi[wv][c] = bi[c]; s_g[wv][c] = g[c]; s_gh[wv][c] = gh[c]; s_gw[wv][c] = gw[c]; s_vs[wv][c] = vs[c]; }
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 4; ++c) {                  // every thread merges the 4 waves in order (it needs the chunk maximum below)
        m[c] = s_m[0][c]; bi[c] = s_bi[0][c]; g[c] = s_g[0][c]; gh[c] = s_gh[0][c]; gw[c] = s_gw[0][c]; vs[c] = s_vs[0][c];
#pragma unroll
        for (int k = 1; k < NW; ++k) {
            const float om = s_m[k][c], og = s_g[k][c];
            const int oi = s_bi[k][c], oh = s_gh[k][c], ow = s_gw[k][c];
            const bool take = om > m[c] || (om == m[c] && oi < bi[c]);
            m[c] = take ? om : m[c]; bi[c] = take ? oi : bi[c];
            const bool gt = og > g[c], eq = og == g[c];
            gh[c] = gt ? oh : (eq ? min(gh[c], oh) : gh[c]);
            gw[c] = gt ? ow : (eq ? min(gw[c], ow) : gw[c]);
            g[c] = gt ? og : g[c];
            vs[c] += s_vs[k][c];
        }
    }
    double s[4] = {0, 0, 0, 0}, sh[4] = {0, 0, 0, 0}, sw[4] = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < HM_RPT; ++i) {
        const int r = r0 + tid + i * HM_THREADS;
        if (r >= r1) continue;
        const float xv[4] = {xs[i].x, xs[i].y, xs[i].z, xs[i].w};
        const int h = r / side, w = r - h * side;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const double e = (double)expf(xv[c] - m[c]);
            s[c] += e; sh[c] += e * h; sw[c] += e * w;
        }
    }
    {
        double all[14] = {s[0], s[1], s[2], s[3], sh[0], sh[1], sh[2], sh[3], sw[0], sw[1], sw[2], sw[3], bce_a, bce_v};
        wave_sum_n<14>(all);
#pragma unroll
        for (int c = 0; c < 4; ++c) { s[c] = all[c]; sh[c] = all[4 + c]; sw[c] = all[8 + c]; }
        bce_a = all[12]; bce_v = all[13];
    }
#pragma unroll
    for (int c = 0; c < 4; ++c)
        if (lane == 0) { s_s[wv][c][0] = s[c]; s_s[wv][c][1] = sh[c]; s_s[wv][c][2] = sw[c]; }
    if (BCE && lane == 0) { s_b[wv][0] = bce_a; s_b[wv][1] = bce_v; }
    __syncthreads();
    if (BCE && tid == 64) {
        double A = 0, V = 0;
#pragma unroll
        for (int k = 0; k < NW; ++k) { A += s_b[k][0]; V += s_b[k][1]; }
        bce_part[2 * (size_t)blockIdx.x] = A; bce_part[2 * (size_t)blockIdx.x + 1] = V;
    }
    if (tid < 4) {
        const int c = tid;
        double S = 0, SH = 0, SW = 0;
#pragma unroll
        for (int k = 0; k < NW; ++k) { S += s_s[k][c][0]; SH += s_s[k][c][1]; SW += s_s[k][c][2]; }
        double* q = part + ((size_t)blockIdx.x * 4 + c) * HM_REC;
        // (channel c's values by selects over static indices: `m[c]` with c = tid is a run-time index, and ONE of those puts all six
        //  per-channel arrays into scratch memory -- a dispatch that needs scratch waits for its set-up: this 5-us kernel took 30)
        float mc = m[0], gc = g[0];
        int bic = bi[0], ghc = gh[0], gwc = gw[0];
        double vsc = vs[0];
#pragma unroll
        for (int k = 1; k < 4; ++k) {
            mc = c == k ? m[k] : mc; gc = c == k ? g[k] : gc; bic = c == k ? bi[k] : bic; ghc = c == k ? gh[k] : ghc; gwc = c == k ? gw[k] : gwc;
            vsc = c == k ? vs[k] : vsc;
        }
        q[0] = mc; q[1] = S; q[2] = SH; q[3] = SW; q[4] = mc; q[5] = bic;
        q[6] = gc; q[7] = ghc; q[8] = gwc; q[9] = vsc;
    }
}

// one WAVE per (frame, level, channel): lane k merges chunks k, k + 64, ... of the level in ascending order, the lanes merge
// through a fixed xor tree (softmax merge in fp64)
__device__ inline void hm_final_wave(const double* __restrict__ part, float* __restrict__ expect, float* __restrict__ stats,
                                     int64_t* __restrict__ argmax, float* __restrict__ gt, float* __restrict__ vmean, const HmLevels& L) {
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (t >= L.batch * L.n_levels * 4) return;
    const int c = t & 3, l = (t >> 2) % L.n_levels, b = (t >> 2) / L.n_levels;
    const int k0 = L.chunk0[l], k1 = L.chunk0[l + 1];
    double M = -__builtin_inf();
    for (int k = k0 + lane; k < k1; k += 64) M = fmax(M, part[(((size_t)b * L.total_chunks + k) * 4 + c) * HM_REC]);
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) M = fmax(M, __shfl_xor(M, s));
    double S = 0, SH = 0, SW = 0, VS = 0, best = -__builtin_inf(), bidx = 9.0e15, g = -__builtin_inf();
    double gh = 2147483647.0, gw = 2147483647.0;
    for (int k = k0 + lane; k < k1; k += 64) {
        const double* q = part + (((size_t)b * L.total_chunks + k) * 4 + c) * HM_REC;
        const double f = q[0] == M ? 1.0 : exp(q[0] - M);
        S += q[1] * f; SH += q[2] * f; SW += q[3] * f; VS += q[9];
        if (q[4] > best) { best = q[4]; bidx = q[5]; }                 // chunks ascend: first maximum kept
        if (q[6] > g) { g = q[6]; gh = q[7]; gw = q[8]; }
        else if (q[6] == g) { gh = fmin(gh, q[7]); gw = fmin(gw, q[8]); }
    }
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) {
        S += __shfl_xor(S, s); SH += __shfl_xor(SH, s); SW += __shfl_xor(SW, s); VS += __shfl_xor(VS, s);
        const double ob = __shfl_xor(best, s), oi = __shfl_xor(bidx, s);
        if (ob > best || (ob == best && oi < bidx)) { best = ob; bidx = oi; }      // (row indices ascend with the chunks: the first maximum)
        const double og = __shfl_xor(g, s), oh = __shfl_xor(gh, s), ow = __shfl_xor(gw, s);
        if (og > g) { g = og; gh = oh; gw = ow; }
        else if (og == g) { gh = fmin(gh, oh); gw = fmin(gw, ow); }
    }
    if (lane != 0) return;
    const size_t o = ((size_t)b * L.n_levels + l) * 4 + c;
    expect[o * 2] = (float)(SH / S);
    expect[o * 2 + 1] = (float)(SW / S);
    if (stats) { stats[o * 2] = (float)M; stats[o * 2 + 1] = (float)S; }
    if (argmax) argmax[o] = (int64_t)bidx;
    if (gt) { gt[o * 2] = (float)gh; gt[o * 2 + 1] = (float)gw; }
    if (vmean) vmean[o] = (float)(VS / ((double)L.side[l] * L.side[l]));
}
__global__ __launch_bounds__(256) void k_hm_final(const double* __restrict__ part, float* __restrict__ expect, float* __restrict__ stats,
                                                  int64_t* __restrict__ argmax, float* __restrict__ gt, float* __restrict__ vmean, const HmLevels L) {
    hm_final_wave(part, expect, stats, argmax, gt, vmean, L);
}

// d logits[r, c] = p * ((h - E_h) * g_h + (w - E_w) * g_w),  p = exp(x - m) / s
__global__ __launch_bounds__(HM_THREADS) void k_hm_bwd(const float* __restrict__ logits, const float* __restrict__ expect,
                                                       const float* __restrict__ stats, const float* __restrict__ d_expect,
                                                       float* __restrict__ d_logits, const HmLevels L) {
    const long long t = (long long)blockIdx.x * HM_THREADS + threadIdx.x;
    if (t >= (long long)L.batch * L.n_rows) return;
    const int b = (int)(t / L.n_rows), r = (int)(t - (long long)b * L.n_rows);
    int l = -1;
    for (int k = 0; k < L.n_levels; ++k) if (r >= L.start[k] && r < L.start[k] + L.side[k] * L.side[k]) l = k;
    float4 out = {0.f, 0.f, 0.f, 0.f};
    if (l >= 0) {
        const int rr = r - L.start[l], side = L.side[l];
        const float h = (float)(rr / side), w = (float)(rr - (rr / side) * side);
        const float4 x = *reinterpret_cast<const float4*>(logits + (size_t)t * 4);
        const float xv[4] = {x.x, x.y, x.z, x.w};
        float o[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const size_t q = (((size_t)b * L.n_levels + l) * 4 + c) * 2;
            const float p = expf(xv[c] - stats[q]) / stats[q + 1];
            o[c] = p * ((h - expect[q]) * d_expect[q] + (w - expect[q + 1]) * d_expect[q + 1]);
        }
        out = float4{o[0], o[1], o[2], o[3]};
    }
    *reinterpret_cast<float4*>(d_logits + (size_t)t * 4) = out;
}

// ---- weighted BCE with logits ---------------------------------------------------------------------------------
constexpr int BCE_BLOCKS = 2048;


// 16 B per operand and lane, two float4 groups in flight per stream (one element per lane and iteration made the pass a chain
// of dependent memory round trips: 84 us for 110 MB at batch 32); n4 = n / 4 whole groups, the last n % 4 elements by lane 0 of
// the last workgroup.  fp64 sums per lane, fixed tree per workgroup, fixed order over workgroups (k_bce_final).
__global__ __launch_bounds__(HM_THREADS) void k_bce_partial(const float* __restrict__ x, const float* __restrict__ y,
                                                            const float* __restrict__ valid, long long n, float ones_weight,
                                                            double* __restrict__ part) {
    __shared__ double red[HM_THREADS];
    const int tid = threadIdx.x;
    double a = 0, v = 0;
    auto one = [&](float xi, float yi, float vi) {
        const float w = (ones_weight > 1.0f && yi == 1.0f) ? ones_weight : 1.0f;
        a += (double)(w * bce_logits(xi, yi)) * vi;
        v += vi;
    };
    const long long n4 = n >> 2, stride = (long long)gridDim.x * HM_THREADS;
    const float4 ones = {1.f, 1.f, 1.f, 1.f};
    for (long long i = (long long)blockIdx.x * HM_THREADS + tid; i < n4; i += 2 * stride) {
        const long long j = i + stride;
        const bool two = j < n4;
        const float4 x0 = reinterpret_cast<const float4*>(x)[i], y0 = reinterpret_cast<const float4*>(y)[i];
        const float4 v0 = valid ? reinterpret_cast<const float4*>(valid)[i] : ones;
        const float4 x1 = two ? reinterpret_cast<const float4*>(x)[j] : ones, y1 = two ? reinterpret_cast<const float4*>(y)[j] : ones;
        const float4 v1 = (two && valid) ? reinterpret_cast<const float4*>(valid)[j] : ones;
        one(x0.x, y0.x, v0.x); one(x0.y, y0.y, v0.y); one(x0.z, y0.z, v0.z); one(x0.w, y0.w, v0.w);
        if (two) { one(x1.x, y1.x, v1.x); one(x1.y, y1.y, v1.y); one(x1.z, y1.z, v1.z); one(x1.w, y1.w, v1.w); }
    }
    if (blockIdx.x == gridDim.x - 1 && tid == 0)
        for (long long i = n4 << 2; i < n; ++i) one(x[i], y[i], valid ? valid[i] : 1.0f);
    a = block_sum(a, red, tid);
    v = block_sum(v, red, tid);
    if (tid == 0) { part[2 * blockIdx.x] = a; part[2 * blockIdx.x + 1] = v; }
}

__global__ __launch_bounds__(64) void k_bce_final(const double* __restrict__ part, int blocks, float* __restrict__ out) {
    // one wave: lane l sums partials l, l+64, ... in ascending order, then a fixed xor tree (same bits every run)
    const int l = threadIdx.x;
    double a = 0, v = 0;
    for (int k = l; k < blocks; k += 64) { a += part[2 * k]; v += part[2 * k + 1]; }
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) { a += __shfl_xor(a, s); v += __shfl_xor(v, s); }
    if (l == 0) { out[0] = (float)a; out[1] = (float)v; out[2] = (float)(a / v); }
}

__global__ __launch_bounds__(HM_THREADS) void k_bce_bwd(const float* __restrict__ x, const float* __restrict__ y,
                                                        const float* __restrict__ valid, long long n, float ones_weight,
                                                        const float* __restrict__ scale, float* __restrict__ dx) {
    const long long i = (long long)blockIdx.x * HM_THREADS + threadIdx.x;
    if (i >= n) return;
    const float yi = y[i], vi = valid ? valid[i] : 1.0f;
    const float w = (ones_weight > 1.0f && yi == 1.0f) ? ones_weight : 1.0f;
    const float sg = 1.0f / (1.0f + expf(-x[i]));
    dx[i] = (sg - yi) * w * vi * scale[0];
}

static int fill_levels(int batch, int64_t n_rows, const int* level_start, const int* level_side, int n_levels, HmLevels& L) {
    if (batch < 1 || n_rows < 1 || n_levels < 1 || n_levels > HM_MAX_LEVELS || !level_start || !level_side)
        return set_error(EG_ERR_ARG, "bad level table");
    if (n_rows * (int64_t)batch >= (1ll << 31)) return set_error(EG_ERR_ARG, "batch * rows exceeds int32");
    L.n_levels = n_levels; L.n_rows = (int)n_rows; L.batch = batch;
    int chunks = 0;
    for (int l = 0; l < n_levels; ++l) {
        const long long sz = (long long)level_side[l] * level_side[l];
        if (level_side[l] < 1 || level_start[l] < 0 || level_start[l] + sz > n_rows) return set_error(EG_ERR_ARG, "level outside the frame's rows");
        L.start[l] = level_start[l]; L.side[l] = level_side[l]; L.chunk0[l] = chunks;
        chunks += (int)((sz + HM_CHUNK - 1) / HM_CHUNK);
    }
    L.chunk0[n_levels] = chunks;
    L.total_chunks = chunks;
    return EG_OK;
}

// ---- ExpectedLandmarkMSE's combination of the per-(frame, level, channel) expectations (criterion.py:133-151) -----------------
//   loss = w * sum_{l,c,xy} [ sum_b ((e - gt) / side_l)^2 * vmean ] / nv_{l,c},   nv = sum_b vmean (1 where that is 0)
// and its gradient with respect to the expectations, in ONE single-workgroup launch (torch needed 9 launches forward and 4
// backward for these [B, L, 4, 2] tensors; a launch costs ~5 us of GPU time and ~10 us of host time whatever it computes).
// Thread t < L * 8 owns (l, c, xy) and walks the frames in order; the L * 8 partial sums are added in index order (fp64).
__global__ __launch_bounds__(128) void k_elm_reduce(const float* __restrict__ expect, const float* __restrict__ gt,
                                                    const float* __restrict__ vmean, const float* __restrict__ inv_side, int batch,
                                                    int n_levels, float weight, float* __restrict__ loss, float* __restrict__ d_expect) {
    __shared__ double s_part[128];
    const int t = threadIdx.x, n = n_levels * 8;
    double part = 0.0;
    if (t < n) {
        const int l = t >> 3, c = (t >> 1) & 3;
        float nv = 0.f;
        for (int b = 0; b < batch; ++b) nv += vmean[((size_t)b * n_levels + l) * 4 + c];
        if (nv == 0.f) nv = 1.f;
        const float is = inv_side[l];
        for (int b = 0; b < batch; ++b) {
            const size_t i = (size_t)b * n + t;
            const float diff = (expect[i] - gt[i]) * is;
            const float wv = vmean[((size_t)b * n_levels + l) * 4 + c] / nv;
            part += (double)(diff * diff * wv);
            d_expect[i] = 2.0f * weight * diff * wv * is;
        }
    }
    s_part[t] = part;
    __syncthreads();
    if (t == 0) {
        double tot = 0.0;
        for (int k = 0; k < n; ++k) tot += s_part[k];
        *loss = (float)(tot * (double)weight);
    }
}


// ---- the three criteria of a training step as ONE node (engine.py:582-600 + criterion.py:13-27, :36-48, :93-151) -----------------
// forward: k_bce_partial + k_hm_partial + k_hm_final + k_criteria_final (4 launches); backward: k_criteria_bwd (1 launch).  As
// separate autograd nodes with torch's glue around them the same arithmetic was ~33 launches of ~5 us each -- a sixth of a
// batch-1 training step captured into a HIP graph.
struct CriteriaFinal {
    const float *expect, *gt, *vmean, *inv_side;
    int batch, n_levels;
    float w_elm;
    float* d_expect;
    const double* bce_part;
    int bce_blocks;
    float w_bce;
    const float *coord_pred, *coord_y;      // nullable: no coordinate criterion
    int n_coord;
    float w_coord;
    float* d_coord;                         // [n_coord] = w_coord * 2 (pred - y) / n_coord
    float *total, *bce, *elm, *coord;       // one float each
    float* bce_scale;                       // w_bce / sum(valid): what every element of the BCE gradient is multiplied by
};

// (the first 128 threads of the workgroup work; every thread of it must come here: one barrier)
__device__ inline void criteria_final_body(const CriteriaFinal& a) {
    __shared__ double s_part[128];
    __shared__ double s_bce[2][128];
    __shared__ double s_coord[128];
    const int t = threadIdx.x, n = a.n_levels * 8;
    if (t >= 128) { __syncthreads(); return; }
    // ExpectedLandmarkMSE (k_elm_reduce's arithmetic)
    double part = 0.0;
    if (t < n) {
        const int l = t >> 3, c = (t >> 1) & 3;
        // (loads in batches of 8 in front of their uses, the additions in the plain loops' order: one workgroup walking a batch of 32
        //  frame by frame was 26 us of dependent round trips)
        const size_t vs = (size_t)a.n_levels * 4;
        const float* vm = a.vmean + (size_t)l * 4 + c;
        float nv = 0.f;
        int b = 0;
        for (; b + 8 <= a.batch; b += 8) {
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = vm[(size_t)(b + k) * vs];
#pragma unroll
            for (int k = 0; k < 8; ++k) nv += v[k];
        }
        for (; b < a.batch; ++b) nv += vm[(size_t)b * vs];
        if (nv == 0.f) nv = 1.f;
        const float is = a.inv_side[l];
        for (b = 0; b + 8 <= a.batch; b += 8) {
            float ex[8], gg[8], v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) { const size_t i = (size_t)(b + k) * n + t; ex[k] = a.expect[i]; gg[k] = a.gt[i]; v[k] = vm[(size_t)(b + k) * vs]; }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const size_t i = (size_t)(b + k) * n + t;
                const float diff = (ex[k] - gg[k]) * is;
                const float wv = v[k] / nv;
                part += (double)(diff * diff * wv);
                a.d_expect[i] = 2.0f * a.w_elm * diff * wv * is;
            }
        }
        for (; b < a.batch; ++b) {
            const size_t i = (size_t)b * n + t;
            const float diff = (a.expect[i] - a.gt[i]) * is;
            const float wv = vm[(size_t)b * vs] / nv;
            part += (double)(diff * diff * wv);
            a.d_expect[i] = 2.0f * a.w_elm * diff * wv * is;
        }
    }
    s_part[t] = part;
    // weighted BCE: the partials of k_bce_partial, thread t sums partials t, t + 128, ... in ascending order
    double ba = 0.0, bv = 0.0;
    {
        int k = t;
        for (; k + 7 * 128 < a.bce_blocks; k += 8 * 128) {
            double pa[8], pv[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) { pa[j] = a.bce_part[2 * (k + 128 * j)]; pv[j] = a.bce_part[2 * (k + 128 * j) + 1]; }
#pragma unroll
            for (int j = 0; j < 8; ++j) { ba += pa[j]; bv += pv[j]; }
        }
        for (; k < a.bce_blocks; k += 128) { ba += a.bce_part[2 * k]; bv += a.bce_part[2 * k + 1]; }
    }
    s_bce[0][t] = ba;
    s_bce[1][t] = bv;
    // coordinate MSE (mean over all elements) and its gradient
    double cs = 0.0;
    if (a.coord_pred) {
        const float inv_n = 1.0f / (float)a.n_coord;
        for (int i = t; i < a.n_coord; i += 128) {
            const float d = a.coord_pred[i] - a.coord_y[i];
            cs += (double)(d * d);
            a.d_coord[i] = a.w_coord * 2.0f * d * inv_n;
        }
    }
    s_coord[t] = cs;
    __syncthreads();
    if (t == 0) {
        double elm = 0.0, A = 0.0, V = 0.0, cc = 0.0;
        for (int k = 0; k < n; ++k) elm += s_part[k];
        for (int k = 0; k < 128; ++k) { A += s_bce[0][k]; V += s_bce[1][k]; cc += s_coord[k]; }
        const float f_elm = (float)(elm * (double)a.w_elm);
        const float f_bce = a.w_bce * (float)(A / V);
        const float f_coord = a.coord_pred ? a.w_coord * (float)(cc / (double)a.n_coord) : 0.f;
        *a.elm = f_elm;
        *a.bce = f_bce;
        if (a.coord) *a.coord = f_coord;
        *a.total = (f_bce + f_elm) + f_coord;
        *a.bce_scale = a.w_bce / (float)V;
    }
}
__global__ __launch_bounds__(128) void k_criteria_final(const CriteriaFinal a) { criteria_final_body(a); }
// k_hm_final and, in the workgroup that finishes last (last_workgroup_out), k_criteria_final: one launch
__global__ __launch_bounds__(256) void k_hm_final_criteria(const double* __restrict__ part, float* __restrict__ expect, float* __restrict__ stats,
                                                           float* __restrict__ gt, float* __restrict__ vmean, const HmLevels L,
                                                           const CriteriaFinal a, unsigned* ticket) {
    hm_final_wave(part, expect, stats, (int64_t*)nullptr, gt, vmean, L);
    if (last_workgroup_out(ticket, gridDim.x)) criteria_final_body(a);
}

struct CriteriaBwd {
    const float *logits, *labels, *valid;    // valid nullable (= 1)
    const float *expect, *stats, *d_expect;
    const float* bce_scale;
    float ones_weight;
    const float *g_total, *g_bce, *g_elm, *g_coord;      // upstream gradients (device scalars), each nullable
    const float* d_coord;
    int n_coord;
    float* d_logits;
    float* d_coord_out;                      // nullable
};

__global__ __launch_bounds__(HM_THREADS) void k_criteria_bwd(const CriteriaBwd a, const HmLevels L) {
    const float gt = a.g_total ? *a.g_total : 0.f;
    const float s_bce = (gt + (a.g_bce ? *a.g_bce : 0.f)) * a.bce_scale[0];
    const float s_elm = gt + (a.g_elm ? *a.g_elm : 0.f);
    const long long t = (long long)blockIdx.x * HM_THREADS + threadIdx.x;
    if (a.d_coord_out && t < a.n_coord) a.d_coord_out[t] = (gt + (a.g_coord ? *a.g_coord : 0.f)) * a.d_coord[t];
    if (t >= (long long)L.batch * L.n_rows) return;
    const int b = (int)(t / L.n_rows), r = (int)(t - (long long)b * L.n_rows);
    int l = -1;
    for (int k = 0; k < L.n_levels; ++k) if (r >= L.start[k] && r < L.start[k] + L.side[k] * L.side[k]) l = k;
    const float4 x = *reinterpret_cast<const float4*>(a.logits + (size_t)t * 4);
    const float4 y = *reinterpret_cast<const float4*>(a.labels + (size_t)t * 4);
    const float4 v = a.valid ? *reinterpret_cast<const float4*>(a.valid + (size_t)t * 4) : float4{1.f, 1.f, 1.f, 1.f};
    const float xv[4] = {x.x, x.y, x.z, x.w}, yv[4] = {y.x, y.y, y.z, y.w}, vv[4] = {v.x, v.y, v.z, v.w};
    float o[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const float w = (a.ones_weight > 1.0f && yv[c] == 1.0f) ? a.ones_weight : 1.0f;
        const float sg = 1.0f / (1.0f + expf(-xv[c]));
        o[c] = (sg - yv[c]) * w * vv[c] * s_bce;
    }
    if (l >= 0) {
        const int rr = r - L.start[l], side = L.side[l];
        const float h = (float)(rr / side), w = (float)(rr - (rr / side) * side);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const size_t q = (((size_t)b * L.n_levels + l) * 4 + c) * 2;
            const float p = expf(xv[c] - a.stats[q]) / a.stats[q + 1];
            o[c] += s_elm * (p * ((h - a.expect[q]) * a.d_expect[q] + (w - a.expect[q + 1]) * a.d_expect[q + 1]));
        }
    }
    *reinterpret_cast<float4*>(a.d_logits + (size_t)t * 4) = float4{o[0], o[1], o[2], o[3]};
}

}  // namespace eg

using namespace eg;

extern "C" {

size_t eg_heatmap_workspace_bytes(int batch, const int* level_side, int n_levels) {
    if (batch < 1 || !level_side || n_levels < 1 || n_levels > HM_MAX_LEVELS) return 0;
    size_t chunks = 0;
    for (int l = 0; l < n_levels; ++l) chunks += ((size_t)level_side[l] * level_side[l] + HM_CHUNK - 1) / HM_CHUNK;
    const size_t hm = (size_t)batch * chunks * 4 * HM_REC * sizeof(double);
    const size_t bce = (size_t)BCE_BLOCKS * 2 * sizeof(double);
    return hm > bce ? hm : bce;
}

int eg_heatmap_expect_fwd(const float* logits, const float* labels, const float* valid, int batch, int64_t n_rows,
                          const int* level_start, const int* level_side, int n_levels, void* workspace, float* expect,
                          float* stats, int64_t* argmax, float* gt, float* vmean, eg_stream_t stream) {
    if (!logits || !workspace || !expect) return set_error(EG_ERR_ARG, "logits, workspace and expect must not be NULL");
    if ((gt && !labels) || (vmean && !valid)) return set_error(EG_ERR_ARG, "gt needs labels, vmean needs valid");
    HmLevels L{};
    int rc = fill_levels(batch, n_rows, level_start, level_side, n_levels, L);
    if (rc != EG_OK) return rc;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_hm_partial<false>, dim3((unsigned)(batch * L.total_chunks)), dim3(HM_THREADS), 0, s, logits,
                       gt ? labels : nullptr, vmean ? valid : nullptr, (double*)workspace, L, 0.f, (double*)nullptr);
    const int n_out = batch * n_levels * 4;
    hipLaunchKernelGGL(k_hm_final, dim3((unsigned)((n_out + 3) / 4)), dim3(256), 0, s, (const double*)workspace, expect,
                       stats, argmax, gt, vmean, L);
    EG_HIP_TRY(hipGetLastError());
    return EG_OK;
}

int eg_heatmap_expect_bwd(const float* logits, const float* expect, const float* stats, const float* d_expect, int batch,
                          int64_t n_rows, const int* level_start, const int* level_side, int n_levels, float* d_logits,
                          eg_stream_t stream) {
    if (!logits || !expect || !stats || !d_expect || !d_logits) return set_error(EG_ERR_ARG, "NULL argument");
    HmLevels L{};
    int rc = fill_levels(batch, n_rows, level_start, level_side, n_levels, L);
    if (rc != EG_OK) return rc;
    const long long n = (long long)batch * n_rows;
    hipLaunchKernelGGL(k_hm_bwd, dim3((unsigned)((n + HM_THREADS - 1) / HM_THREADS)), dim3(HM_THREADS), 0, (hipStream_t)stream,
                       logits, expect, stats, d_expect, d_logits, L);
    EG_HIP_TRY(hipGetLastError());
    return EG_OK;
}

int eg_bce_logits_fwd(const float* logits, const float* labels, const float* valid, int64_t n, float ones_weight,
                      void* workspace, float* out3, eg_stream_t stream) {
    if (!logits || !labels || !workspace || !out3 || n < 1) return set_error(EG_ERR_ARG, "bad argument");
    if (((uintptr_t)logits | (uintptr_t)labels | (uintptr_t)valid) & 15) return set_error(EG_ERR_ARG, "logits / labels / valid must be 16-byte aligned");
    long long blocks = ((n >> 2) + 2 * HM_THREADS - 1) / (2 * HM_THREADS);
    if (blocks < 1) blocks = 1;
    if (blocks > BCE_BLOCKS) blocks = BCE_BLOCKS;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_bce_partial, dim3((unsigned)blocks), dim3(HM_THREADS), 0, s, logits, labels, valid, (long long)n,
                       ones_weight, (double*)workspace);
    hipLaunchKernelGGL(k_bce_final, dim3(1), dim3(64), 0, s, (const double*)workspace, (int)blocks, out3);
    EG_HIP_TRY(hipGetLastError());
    return EG_OK;
}

int eg_bce_logits_bwd(const float* logits, const float* labels, const float* valid, int64_t n, float ones_weight,
                      const float* scale_dev, float* d_logits, eg_stream_t stream) {
    if (!logits || !labels || !scale_dev || !d_logits || n < 1) return set_error(EG_ERR_ARG, "bad argument");
    hipLaunchKernelGGL(k_bce_bwd, dim3((unsigned)((n + HM_THREADS - 1) / HM_THREADS)), dim3(HM_THREADS), 0, (hipStream_t)stream,
                       logits, labels, valid, (long long)n, ones_weight, scale_dev, d_logits);
    EG_HIP_TRY(hipGetLastError());
    return EG_OK;
}

int eg_elm_reduce(const float* expect, const float* gt, const float* vmean, const float* inv_side, int batch, int n_levels, float weight,
                  float* loss, float* d_expect, eg_stream_t stream) {
    if (!expect || !gt || !vmean || !inv_side || !loss || !d_expect || batch < 1 || n_levels < 1 || n_levels > 16)
        return set_error(EG_ERR_ARG, "bad argument");
    hipLaunchKernelGGL(k_elm_reduce, dim3(1), dim3(128), 0, (hipStream_t)stream, expect, gt, vmean, inv_side, batch, n_levels, weight, loss, d_expect);
    EG_HIP_TRY(hipGetLastError());
    return EG_OK;
}

// (sum, sum valid) pairs of the BCE behind the heat-map records: one per heat-map workgroup, or k_bce_partial's BCE_BLOCKS
static size_t criteria_bce_slots(int batch, size_t chunks) {
    const size_t w = (size_t)batch * chunks;
    return w > (size_t)BCE_BLOCKS ? w : (size_t)BCE_BLOCKS;
}

size_t eg_criteria_workspace_bytes(int batch, const int* level_side, int n_levels) {
    if (batch < 1 || !level_side || n_levels < 1 || n_levels > HM_MAX_LEVELS) return 0;
    size_t chunks = 0;
    for (int l = 0; l < n_levels; ++l) chunks += ((size_t)level_side[l] * level_side[l] + HM_CHUNK - 1) / HM_CHUNK;
    return (size_t)batch * chunks * 4 * HM_REC * sizeof(double) + criteria_bce_slots(batch, chunks) * 2 * sizeof(double) +
           (size_t)batch * n_levels * 12 * sizeof(float);           // + gt [B,L,4,2] and vmean [B,L,4]
}

int eg_criteria_fwd(const float* logits, const float* labels, const float* valid, int batch, int64_t n_rows, const int* level_start,
                    const int* level_side, int n_levels, const float* inv_side, float bce_ones_weight, float w_bce, float w_elm,
                    const float* coord_pred, const float* coord_y, int64_t n_coord, float w_coord, void* workspace, float* expect,
                    float* stats, float* d_expect, float* d_coord, float* bce_scale, float* total, float* bce, float* elm, float* coord,
                    eg_stream_t stream) {
    if (!logits || !labels || !valid || !inv_side || !workspace || !expect || !stats || !d_expect || !bce_scale || !total || !bce || !elm)
        return set_error(EG_ERR_ARG, "NULL argument");
    if ((coord_pred != nullptr) != (coord_y != nullptr) || (coord_pred && (!d_coord || !coord || n_coord < 1 || n_coord >= (1 << 30))))
        return set_error(EG_ERR_ARG, "the coordinate criterion needs predictions, targets, d_coord and coord");
    if (((uintptr_t)logits | (uintptr_t)labels | (uintptr_t)valid) & 15) return set_error(EG_ERR_ARG, "logits / labels / valid must be 16-byte aligned");
    HmLevels L{};
    int rc = fill_levels(batch, n_rows, level_start, level_side, n_levels, L);
    if (rc != EG_OK) return rc;
    hipStream_t s = (hipStream_t)stream;
    double* hm_part = (double*)workspace;
    double* bce_part = hm_part + (size_t)batch * L.total_chunks * 4 * HM_REC;
    const long long n = (long long)batch * n_rows * 4;
    // do the levels tile the frame's rows (each row in exactly one level)?  Then the heat-map pass takes the BCE's partial sums as well
    bool tiled = true;
    {
        long long covered = 0;
        for (int l = 0; l < n_levels; ++l) covered += (long long)level_side[l] * level_side[l];
        if (covered != n_rows) tiled = false;
        for (int l = 0; l < n_levels && tiled; ++l)
            for (int k = 0; k < l; ++k) {
                const long long a0 = level_start[l], a1 = a0 + (long long)level_side[l] * level_side[l];
                const long long b0 = level_start[k], b1 = b0 + (long long)level_side[k] * level_side[k];
                if (a0 < b1 && b0 < a1) tiled = false;
            }
    }
    long long blocks = (long long)batch * L.total_chunks;
    // gt / vmean of the heat maps live behind the expectations' statistics in `stats`'s sibling buffers: the caller's d_expect doubles
    // as scratch for neither -- they are written into the tail of the workspace the final kernel reads them from
    float* gt = (float*)(bce_part + criteria_bce_slots(batch, L.total_chunks) * 2);
    float* vmean = gt + (size_t)batch * n_levels * 8;
    if (tiled) {
        hipLaunchKernelGGL(k_hm_partial<true>, dim3((unsigned)(batch * L.total_chunks)), dim3(HM_THREADS), 0, s, logits, labels, valid, hm_part, L,
                           bce_ones_weight, bce_part);
    } else {
        blocks = ((n >> 2) + 2 * HM_THREADS - 1) / (2 * HM_THREADS);
        if (blocks < 1) blocks = 1;
        if (blocks > BCE_BLOCKS) blocks = BCE_BLOCKS;
        hipLaunchKernelGGL(k_bce_partial, dim3((unsigned)blocks), dim3(HM_THREADS), 0, s, logits, labels, valid, n, bce_ones_weight, bce_part);
        hipLaunchKernelGGL(k_hm_partial<false>, dim3((unsigned)(batch * L.total_chunks)), dim3(HM_THREADS), 0, s, logits, labels, valid, hm_part, L,
                           0.f, (double*)nullptr);
    }
    const int n_out = batch * n_levels * 4;
    CriteriaFinal a{expect, gt, vmean, inv_side, batch, n_levels, w_elm, d_expect, bce_part, (int)blocks, w_bce, coord_pred, coord_y,
                    (int)n_coord, w_coord, d_coord, total, bce, elm, coord, bce_scale};
    const unsigned n_wg = (unsigned)((n_out + 3) / 4);
    if (n_wg <= 16) {
        // a handful of workgroups (batch 1 - 2): the last one out runs the criteria's final step -- one launch (a device-scope release per
        // workgroup is what it costs: at batch 32, 256 of them made this launch 39 us against 8 + 6 for the two below)
        unsigned* ticket = eg_ticket_ptr((void*)s, 1);
        if (!ticket) return set_error(EG_ERR_HIP, "no device memory for a ticket word");
        hipLaunchKernelGGL(k_hm_final_criteria, dim3(n_wg), dim3(256), 0, s, (const double*)hm_part, expect, stats, gt, vmean, L, a, ticket);
    } else {
        hipLaunchKernelGGL(k_hm_final, dim3(n_wg), dim3(256), 0, s, (const double*)hm_part, expect, stats, (int64_t*)nullptr, gt, vmean, L);
        hipLaunchKernelGGL(k_criteria_final, dim3(1), dim3(128), 0, s, a);
    }
    EG_HIP_TRY(hipGetLastError());
    return EG_OK;
}

int eg_criteria_bwd(const float* logits, const float* labels, const float* valid, int batch, int64_t n_rows, const int* level_start,
                    const int* level_side, int n_levels, float bce_ones_weight, const float* expect, const float* stats,
                    const float* d_expect, const float* bce_scale, const float* d_coord, int64_t n_coord, const float* g_total,
                    const float* g_bce, const float* g_elm, const float* g_coord, float* d_logits, float* d_coord_out,
                    eg_stream_t stream) {
    if (!logits || !labels || !expect || !stats || !d_expect || !bce_scale || !d_logits) return set_error(EG_ERR_ARG, "NULL argument");
    if (d_coord_out && (!d_coord || n_coord < 1)) return set_error(EG_ERR_ARG, "d_coord_out needs d_coord");
    HmLevels L{};
    int rc = fill_levels(batch, n_rows, level_start, level_side, n_levels, L);
    if (rc != EG_OK) return rc;
    const long long n = (long long)batch * n_rows;
    if (d_coord_out && n_coord > n) return set_error(EG_ERR_ARG, "more coordinate elements than logit rows");
    const CriteriaBwd a{logits, labels, valid, expect, stats, d_expect, bce_scale, bce_ones_weight, g_total, g_bce, g_elm, g_coord,
                        d_coord, (int)n_coord, d_logits, d_coord_out};
    hipLaunchKernelGGL(k_criteria_bwd, dim3((unsigned)((n + HM_THREADS - 1) / HM_THREADS)), dim3(HM_THREADS), 0, (hipStream_t)stream, a, L);
    EG_HIP_TRY(hipGetLastError());
    return EG_OK;
}

}  // extern "C"
