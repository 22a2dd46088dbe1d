// Node-type filter + the 4 classifier heads in one gfx950 kernel.
// Replaces reference src/core/models.py:485-490 (h[node_type == 0], 4 x nn.Sequential, torch.cat)
// with the head structure of :363-377.  Eval-mode BatchNorm is folded by the caller.
//
// Same 8-wave tile machinery as the layer kernel (tile.h): the four first layers Linear(128,32) are
// one [128 -> 4*32] product on the fp32 MFMA (wave w owns 16 of the 128 hidden channels), written
// back to the LDS tile with BN+ReLU applied.  Then each wave finishes its own 8 rows: lane
// (head = l>>4, o = l&15) keeps row o of that head's Linear(32,16) in 32 registers and reads the
// row's 32 hidden values as LDS broadcasts; Linear(16,1) is a 16-lane reduction.  The valid rows
// are a contiguous range per frame (the filter drops the leading connection rows / trailing
// coordinate rows), so the filter is an address offset, not a gather.
#include "tile.h"

namespace eg {

struct ClsDims {
    int n_per_frame, row_lo, n_valid, batch, tiles_per_frame, sigmoid;
};

constexpr int CLS_THREADS = 512;

__global__ __launch_bounds__(CLS_THREADS, 4) void k_classifier(const float* __restrict__ h, const float* __restrict__ w1,
                                                           const float* __restrict__ s1, const float* __restrict__ t1,
                                                           const float* __restrict__ w2, const float* __restrict__ s2,
                                                           const float* __restrict__ t2, const float* __restrict__ w3,
                                                           const float* __restrict__ b3, float* __restrict__ logits,
                                                           const ClsDims a) {
    __shared__ __attribute__((aligned(16))) float s_a[TILE * LDA + 4];
    __shared__ __attribute__((aligned(16))) float s_out[TILE * 4];

    const int tid = threadIdx.x;
    const int lane_k = tid & 63;
    const int wave = wave_id();

    float wreg[32];
    load_w_slice16(w1, wave, lane_k, 0, wreg);
    // first-layer BN for the 4 channels this lane holds after the MFMA (D layout: 16w + 4q + i)
    const int ch_d = 16 * wave + 4 * (lane_k >> 4);
    const f32x4 s1v = *reinterpret_cast<const f32x4*>(s1 + ch_d);
    const f32x4 t1v = *reinterpret_cast<const f32x4*>(t1 + ch_d);
    // second / third layer: lane (head, o)
    const int head = lane_k >> 4, o = lane_k & 15;
    float w2r[32];
    {
        const f32x4* p = reinterpret_cast<const f32x4*>(w2 + (size_t)(head * 16 + o) * 32);
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const f32x4 q = p[t];
            w2r[4 * t + 0] = q.x; w2r[4 * t + 1] = q.y; w2r[4 * t + 2] = q.z; w2r[4 * t + 3] = q.w;
        }
    }
    const float s2v = s2[head * 16 + o], t2v = t2[head * 16 + o], w3v = w3[head * 16 + o], b3v = b3[head];

    TileWalk walk(WALK_MOD8, a.tiles_per_frame * a.batch, nullptr, reinterpret_cast<int*>(&s_a[TILE * LDA]));
    int tile;
    while (walk.next(tile)) {
        int lane = lane_k;
        asm volatile("" : "+v"(lane));
        const int frame = tile / a.tiles_per_frame;
        const int n0 = (tile - frame * a.tiles_per_frame) * TILE;            // first valid-row index of the tile
        const float* __restrict__ hf = h + ((size_t)frame * a.n_per_frame + a.row_lo) * C;
        const int rows_here = (a.n_valid - n0) < TILE ? (a.n_valid - n0) : TILE;
        const int rl0 = 8 * wave;
        const int last = a.n_valid - 1;
        {
            f32x2 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = load_row2(hf, n0 + rl0 + u < last ? n0 + rl0 + u : last, lane);
#pragma unroll
            for (int u = 0; u < 8; ++u) *reinterpret_cast<f32x2*>(&s_a[(rl0 + u) * LDA + 2 * lane]) = v[u];
        }
        __syncthreads();
        f32x4v acc[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[b] = f32x4v{0.f, 0.f, 0.f, 0.f};
        mfma16_pair<false>(s_a, 0, lane, wreg, acc[0], acc[1]);
        if (rows_here > 32) mfma16_pair<false>(s_a, 32, lane, wreg, acc[2], acc[3]);
        __syncthreads();
        {   // hidden layer 1 (BN + ReLU) back into the tile: row j, channels 16w + 4q .. +3
            const int j = lane & 15, q4 = lane >> 4;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                f32x4v hv;
                hv.x = fmaxf(acc[b].x * s1v.x + t1v.x, 0.f);
                hv.y = fmaxf(acc[b].y * s1v.y + t1v.y, 0.f);
                hv.z = fmaxf(acc[b].z * s1v.z + t1v.z, 0.f);
                hv.w = fmaxf(acc[b].w * s1v.w + t1v.w, 0.f);
                *reinterpret_cast<f32x4v*>(&s_a[(16 * b + j) * LDA + 16 * wave + 4 * q4]) = hv;
            }
        }
        __syncthreads();
        // this wave's 8 rows: Linear(32,16) + BN + ReLU per lane (head, o), Linear(16,1) as a 16-lane sum
#pragma unroll 1
        for (int u = 0; u < 8; ++u) {
            const f32x4* hp = reinterpret_cast<const f32x4*>(&s_a[(rl0 + u) * LDA + 32 * head]);
            float p = 0.f;
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const f32x4 q = hp[t];
                p += w2r[4 * t + 0] * q.x + w2r[4 * t + 1] * q.y + w2r[4 * t + 2] * q.z + w2r[4 * t + 3] * q.w;
            }
            float y = w3v * fmaxf(p * s2v + t2v, 0.f);
            y += __shfl_xor(y, 8, 16);
            y += __shfl_xor(y, 4, 16);
            y += __shfl_xor(y, 2, 16);
            y += __shfl_xor(y, 1, 16);
            y += b3v;
            if (a.sigmoid) y = 1.0f / (1.0f + __expf(-y));
            if (o == 0) s_out[(rl0 + u) * 4 + head] = y;
        }
        // the same wave stores its 8 rows of logits (its own LDS writes are visible to it in order)
        if (lane < 8) {
            const int n = n0 + rl0 + lane;
            if (n < a.n_valid)
                *reinterpret_cast<f32x4*>(logits + ((size_t)frame * a.n_valid + n) * 4) =
                    *reinterpret_cast<const f32x4*>(&s_out[(rl0 + lane) * 4]);
        }
    }
}

}  // namespace eg

using namespace eg;

extern "C" int eg_classifier_fwd(const float* h, int batch, int64_t n_per_frame, int64_t row_lo, int64_t n_valid,
                                 const float* w1, const float* s1, const float* t1, const float* w2, const float* s2,
                                 const float* t2, const float* w3, const float* b3, int sigmoid, float* logits,
                                 eg_stream_t stream) {
    if (!h || !w1 || !s1 || !t1 || !w2 || !s2 || !t2 || !w3 || !b3 || !logits)
        return set_error(EG_ERR_ARG, "NULL argument");
    if (batch < 1 || n_per_frame < 1 || row_lo < 0 || n_valid < 0 || row_lo + n_valid > n_per_frame)
        return set_error(EG_ERR_ARG, "bad row range");
    if (n_per_frame * (int64_t)batch >= (1ll << 31)) return set_error(EG_ERR_ARG, "batch * nodes exceeds int32");
    if (n_valid == 0) return EG_OK;
    ClsDims a{};
    a.n_per_frame = (int)n_per_frame; a.row_lo = (int)row_lo; a.n_valid = (int)n_valid; a.batch = batch;
    a.tiles_per_frame = (int)((n_valid + TILE - 1) / TILE);
    a.sigmoid = sigmoid;
    long long n_tiles = (long long)a.tiles_per_frame * batch;
    long long g = n_tiles < 512 ? n_tiles : 512;                  // 2 resident 8-wave workgroups per CU
    g = (g + 7) / 8 * 8;
    hipLaunchKernelGGL(k_classifier, dim3((unsigned)g), dim3(CLS_THREADS), 0, (hipStream_t)stream, h, w1, s1, t1, w2,
                       s2, t2, w3, b3, logits, a);
    EG_HIP_TRY(hipGetLastError());
    return EG_OK;
}
