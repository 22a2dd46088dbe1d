// Coordinate-graph resampling for gfx950: the reference's dense bilinear_interpolation
// (src/core/models.py:539-553: hat weights relu(1-|c-i|) along h and w, outer product, weighted sum over
// the whole [C,F,F] map = 103 MB of temporaries per frame at 224x224) is mathematically a 4-tap sample.
// One wave per landmark gathers the <= 4 node rows of the main grid it touches (lane = channel pair).
#include "coord_common.h"

namespace eg {

__global__ __launch_bounds__(256) void k_bilinear4_fwd(const float* __restrict__ h, const float* __restrict__ coords,
                                                       float* __restrict__ out, const BilArgs a) {
    const int lane = threadIdx.x & 63;
    const int p = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (p >= a.batch * a.points) return;
    const int frame = p / a.points;
    const f32x2 acc = bilinear_sample(h + ((size_t)frame * a.n_per_frame + a.main_base) * C + 2 * lane, coords[2 * p + 0], coords[2 * p + 1], a.frame);
    *reinterpret_cast<f32x2*>(out + (size_t)frame * a.row_stride + (size_t)(p - frame * a.points) * C + 2 * lane) = acc;
}

// one wave per FRAME, so two landmarks that touch the same pixel never race (4 points: coord_common.h's form with every load up front;
// any other count: the walk, landmark after landmark)
__global__ __launch_bounds__(256) void k_bilinear4_bwd(const float* __restrict__ dout, const float* __restrict__ h,
                                                       const float* __restrict__ coords, float* __restrict__ dh,
                                                       float* __restrict__ dcoords, const BilArgs a, const TapSums ts) {
    const int lane = threadIdx.x & 63;
    const int frame = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (frame >= a.batch) return;
    const size_t fbase = ((size_t)frame * a.n_per_frame + a.main_base) * C + 2 * lane;
    f32x2 tmn = {0.f, 0.f}, tis = tmn, tsc = tmn, tsh = tmn, ts1 = tmn, ts2 = tmn;
    unsigned long long tseed = 0;
    if (ts.z) {
        tmn = *reinterpret_cast<const f32x2*>(ts.bn + 2 * lane);         tis = *reinterpret_cast<const f32x2*>(ts.bn + C + 2 * lane);
        tsc = *reinterpret_cast<const f32x2*>(ts.bn + 2 * C + 2 * lane); tsh = *reinterpret_cast<const f32x2*>(ts.bn + 3 * C + 2 * lane);
        tseed = ts.seed + epoch_now(ts.epoch);
    }
    if (a.points == 4) {
        float gh[4], gw[4];
        const float* dout_f = dout + (size_t)frame * a.row_stride;
        const size_t fb = ((size_t)frame * a.n_per_frame + a.main_base) * C;
        if (ts.z) bilinear_bwd_frame4<true>(dout_f, h, coords + 8 * frame, dh, fb, a.frame, ts, tseed, lane, gh, gw, ts1, ts2);
        else bilinear_bwd_frame4<false>(dout_f, h, coords + 8 * frame, dh, fb, a.frame, ts, tseed, lane, gh, gw, ts1, ts2);
        if (dcoords && lane < 8) {
            float vsel = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) { vsel = lane == 2 * q ? gh[q] : vsel; vsel = lane == 2 * q + 1 ? gw[q] : vsel; }
            dcoords[8 * frame + lane] = vsel;
        }
    } else
    for (int q = 0; q < a.points; ++q) {
        const int p = frame * a.points + q;
        const Taps th = taps_1d(coords[2 * p + 0], a.frame), tw = taps_1d(coords[2 * p + 1], a.frame);
        const f32x2 g = *reinterpret_cast<const f32x2*>(dout + (size_t)frame * a.row_stride + (size_t)q * C + 2 * lane);
        float gh = 0.f, gw = 0.f;
#pragma unroll
        for (int ka = 0; ka < 2; ++ka)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                const size_t off = fbase + ((size_t)th.i[ka] * a.frame + tw.i[kb]) * C;
                const f32x2 v = *reinterpret_cast<const f32x2*>(h + off);
                const float dot = g.x * v.x + g.y * v.y;
                gh += th.dw[ka] * tw.w[kb] * dot;
                gw += th.w[ka] * tw.dw[kb] * dot;
                const float w = th.w[ka] * tw.w[kb];
                if (dh && w != 0.f) {
                    f32x2 cur = *reinterpret_cast<const f32x2*>(dh + off);
                    cur += w * g;
                    *reinterpret_cast<f32x2*>(dh + off) = cur;
                    if (ts.z) {
                        const f32x2 zz = *reinterpret_cast<const f32x2*>(ts.z + off);
                        f32x2 add = w * g;
                        if (ts.p > 0.f) add *= keep_scale2(tseed, (unsigned long long)off, ts.p, ts.inv_keep);
                        const f32x2 v = zz * tsc + tsh, xh = (zz - tmn) * tis;
                        if (ts.relu) { add.x = v.x > 0.f ? add.x : 0.f; add.y = v.y > 0.f ? add.y : 0.f; }
                        ts1 += add;
                        ts2 += add * xh;
                    }
                }
            }
        for (int o = 32; o > 0; o >>= 1) { gh += __shfl_xor(gh, o); gw += __shfl_xor(gw, o); }
        if (dcoords && lane == 0) { dcoords[2 * p + 0] = gh; dcoords[2 * p + 1] = gw; }
    }
    if (ts.z) {
        *reinterpret_cast<f32x2*>(ts.out + (size_t)frame * 2 * C + 2 * lane) = ts1;
        *reinterpret_cast<f32x2*>(ts.out + (size_t)frame * 2 * C + C + 2 * lane) = ts2;
    }
}

}  // namespace eg

using namespace eg;

extern "C" {

int eg_bilinear4_fwd_rows(const float* h, const float* coords, int batch, int points, int64_t n_per_frame, int64_t main_base,
                          int frame, float* out, int64_t out_frame_stride, eg_stream_t stream) {
    if (!h || !coords || !out || batch < 1 || points < 1 || frame < 1 || main_base < 0 ||
        main_base + (int64_t)frame * frame > n_per_frame || out_frame_stride < (int64_t)points * C)
        return set_error(EG_ERR_ARG, "bad argument");
    const BilArgs a{batch, points, frame, (long long)n_per_frame, (long long)main_base, (long long)out_frame_stride};
    const int n = batch * points;
    hipLaunchKernelGGL(k_bilinear4_fwd, dim3((n + 3) / 4), dim3(256), 0, (hipStream_t)stream, h, coords, out, a);
    EG_HIP_TRY(hipGetLastError());
    return EG_OK;
}

int eg_bilinear4_fwd(const float* h, const float* coords, int batch, int points, int64_t n_per_frame, int64_t main_base,
                     int frame, float* out, eg_stream_t stream) {
    return eg_bilinear4_fwd_rows(h, coords, batch, points, n_per_frame, main_base, frame, out, (int64_t)points * C, stream);
}

/* dh (may be NULL) is ACCUMULATED into: dh[tap rows] += w * dout; dcoords (may be NULL) is overwritten. */
static int bilinear4_bwd_rows(const float* dout, int64_t dout_frame_stride, const float* h, const float* coords, int batch, int points,
                             int64_t n_per_frame, int64_t main_base, int frame, float* dh, float* dcoords, const TapSums& ts,
                             eg_stream_t stream) {
    if (!dout || !h || !coords || batch < 1 || points < 1 || frame < 1 || main_base < 0 ||
        main_base + (int64_t)frame * frame > n_per_frame || dout_frame_stride < (int64_t)points * C)
        return set_error(EG_ERR_ARG, "bad argument");
    const BilArgs a{batch, points, frame, (long long)n_per_frame, (long long)main_base, (long long)dout_frame_stride};
    hipLaunchKernelGGL(k_bilinear4_bwd, dim3((batch + 3) / 4), dim3(256), 0, (hipStream_t)stream, dout, h, coords, dh, dcoords, a, ts);
    EG_HIP_TRY(hipGetLastError());
    return EG_OK;
}

int eg_bilinear4_bwd_rows(const float* dout, int64_t dout_frame_stride, const float* h, const float* coords, int batch, int points,
                          int64_t n_per_frame, int64_t main_base, int frame, float* dh, float* dcoords, eg_stream_t stream) {
    return bilinear4_bwd_rows(dout, dout_frame_stride, h, coords, batch, points, n_per_frame, main_base, frame, dh, dcoords, TapSums{}, stream);
}

int eg_bilinear4_bwd_rows_sums(const float* dout, int64_t dout_frame_stride, const float* h, const float* coords, int batch, int points,
                               int64_t n_per_frame, int64_t main_base, int frame, float* dh, float* dcoords,
                               const eg_lower_sums* lower, float* tap_sums, eg_stream_t stream) {
    if (!lower || !lower->z || !lower->bn || !tap_sums || !dh) return set_error(EG_ERR_ARG, "NULL argument");
    if (lower->dropout_p < 0.f || lower->dropout_p >= 1.f) return set_error(EG_ERR_ARG, "dropout_p must be in [0, 1)");
    if (int rc = eg_epoch_required(lower->dropout_p)) return rc;
    const TapSums ts{lower->z, lower->bn, lower->relu, lower->dropout_p, lower->dropout_p > 0.f ? 1.0f / (1.0f - lower->dropout_p) : 1.0f,
                     (unsigned long long)lower->seed, eg_epoch_ptr(), tap_sums};
    return bilinear4_bwd_rows(dout, dout_frame_stride, h, coords, batch, points, n_per_frame, main_base, frame, dh, dcoords, ts, stream);
}

int eg_bilinear4_bwd(const float* dout, const float* h, const float* coords, int batch, int points, int64_t n_per_frame,
                     int64_t main_base, int frame, float* dh, float* dcoords, eg_stream_t stream) {
    return eg_bilinear4_bwd_rows(dout, (int64_t)points * C, h, coords, batch, points, n_per_frame, main_base, frame, dh, dcoords, stream);
}

}  // extern "C"
