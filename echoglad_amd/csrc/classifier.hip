// Node-type filter + the 4 classifier heads in one gfx950 kernel.
// Replaces reference src/core/models.py:485-490 (h[node_type == 0], 4 x nn.Sequential, torch.cat)
// with the head structure of :363-377.  Eval-mode BatchNorm is folded by the caller.
//
// Same 8-wave tile machinery as the layer kernel (tile.h): the four first layers Linear(128,32) are
// one [128 -> 4*32] product on the fp32 MFMA (wave w owns 16 of the 128 hidden channels), written
// back to the LDS tile with BN+ReLU applied.  Linear(32,16) of head hd is a K = 32 product again on the
// MFMA: wave w takes head w & 3 and the 32-row half w >> 2 (its 16 x 32 weight slice sits in 8 registers),
// 8 MFMAs per 16 rows; BN + ReLU and Linear(16,1) follow on the accumulator (4 outputs per lane, two
// cross-lane adds).  The valid rows
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
    // second / third layer: wave -> (head, 32-row half); MFMA A operand lane (o = l&15, kq = l>>4): W2[head][o][8kq + s]
    const int head = wave & 3, rhalf = wave >> 2;
    float w2a[8];
    {
        const f32x4* p = reinterpret_cast<const f32x4*>(w2 + (size_t)(head * 16 + (lane_k & 15)) * 32 + 8 * (lane_k >> 4));
        const f32x4 q0 = p[0], q1 = p[1];
        w2a[0] = q0.x; w2a[1] = q0.y; w2a[2] = q0.z; w2a[3] = q0.w; w2a[4] = q1.x; w2a[5] = q1.y; w2a[6] = q1.z; w2a[7] = q1.w;
    }
    // accumulator layout: lane (row j = l&15, q = l>>4) holds outputs 4q .. 4q+3 of the head
    const int o4 = head * 16 + 4 * (lane_k >> 4);
    const f32x4 s2v = *reinterpret_cast<const f32x4*>(s2 + o4);
    const f32x4 t2v = *reinterpret_cast<const f32x4*>(t2 + o4);
    const f32x4 w3v = *reinterpret_cast<const f32x4*>(w3 + o4);
    const float b3v = b3[head];

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
        {   // paired-row loads: one wave access = 2 consecutive rows (16 B per lane)
            const PairLane pl{lane >> 5, lane & 31};
            f32x4 v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int n = n0 + rl0 + 2 * k + pl.h;
                v[k] = *reinterpret_cast<const f32x4*>(hf + ((unsigned)(n < last ? n : last) * (unsigned)C + 4u * pl.q));
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) *reinterpret_cast<f32x4*>(&s_a[(rl0 + 2 * k + pl.h) * LDA + 4 * pl.q]) = v[k];
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
        // Linear(32,16) + BN + ReLU + Linear(16,1) for (head, 32-row half): K = 32 on the MFMA
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            const int row0 = 32 * rhalf + 16 * rb;
            const int j = lane & 15, kq = lane >> 4;
            const f32x4* hp = reinterpret_cast<const f32x4*>(&s_a[(row0 + j) * LDA + 32 * head + 8 * kq]);
            const f32x4 b0 = hp[0], b1 = hp[1];
            f32x4v z = {0.f, 0.f, 0.f, 0.f};
            z = __builtin_amdgcn_mfma_f32_16x16x4f32(w2a[0], b0.x, z, 0, 0, 0);
            z = __builtin_amdgcn_mfma_f32_16x16x4f32(w2a[1], b0.y, z, 0, 0, 0);
            z = __builtin_amdgcn_mfma_f32_16x16x4f32(w2a[2], b0.z, z, 0, 0, 0);
            z = __builtin_amdgcn_mfma_f32_16x16x4f32(w2a[3], b0.w, z, 0, 0, 0);
            z = __builtin_amdgcn_mfma_f32_16x16x4f32(w2a[4], b1.x, z, 0, 0, 0);
            z = __builtin_amdgcn_mfma_f32_16x16x4f32(w2a[5], b1.y, z, 0, 0, 0);
            z = __builtin_amdgcn_mfma_f32_16x16x4f32(w2a[6], b1.z, z, 0, 0, 0);
            z = __builtin_amdgcn_mfma_f32_16x16x4f32(w2a[7], b1.w, z, 0, 0, 0);
            float y = w3v.x * fmaxf(z.x * s2v.x + t2v.x, 0.f) + w3v.y * fmaxf(z.y * s2v.y + t2v.y, 0.f) +
                      w3v.z * fmaxf(z.z * s2v.z + t2v.z, 0.f) + w3v.w * fmaxf(z.w * s2v.w + t2v.w, 0.f);
            y += __shfl_xor(y, 16);
            y += __shfl_xor(y, 32);
            y += b3v;
            if (a.sigmoid) y = 1.0f / (1.0f + __expf(-y));
            if (kq == 0) s_out[(row0 + j) * 4 + head] = y;
        }
        __syncthreads();
        if (tid < TILE) {
            const int n = n0 + tid;
            if (n < a.n_valid)
                *reinterpret_cast<f32x4*>(logits + ((size_t)frame * a.n_valid + n) * 4) =
                    *reinterpret_cast<const f32x4*>(&s_out[tid * 4]);
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
