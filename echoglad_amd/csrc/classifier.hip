// Node-type filter + the 4 classifier heads in one gfx950 kernel.
// Replaces reference src/core/models.py:485-490 (h[node_type == 0], 4 x nn.Sequential, torch.cat)
// with the head structure of :363-377.  Eval-mode BatchNorm is folded by the caller.
//
// The four first layers Linear(128,32) are one [128 -> 4*32] product on the fp32 MFMA tile of
// tile.h: wave w owns head w's 32 hidden channels.  After the MFMA chain lane (j, h) holds 16 of
// the 32 hidden values of row j, so Linear(32,16) is 16 x 16 FMAs per lane on broadcast LDS
// weights plus one cross-half exchange, and Linear(16,1) is 16 FMAs.  The valid rows are a
// contiguous range per frame (the filter drops the leading connection rows / trailing
// coordinate rows), so the filter is an address offset, not a gather.
#include "tile.h"

namespace eg {

struct ClsArgs {
    const float* h;
    const float* w1; const float* s1; const float* t1;
    const float* w2; const float* s2; const float* t2;
    const float* w3; const float* b3;
    float* logits;
    int n_per_frame, row_lo, n_valid, batch, tiles_per_frame, sigmoid;
};

__global__ __launch_bounds__(256) void k_classifier(const ClsArgs a) {
    __shared__ __attribute__((aligned(16))) float s_a[TILE * LDA];
    __shared__ __attribute__((aligned(16))) float s_s1[C];
    __shared__ __attribute__((aligned(16))) float s_t1[C];
    __shared__ __attribute__((aligned(16))) float s_w2[4 * 2 * 16 * 16];   // [head][half][out][16 of its 32 inputs]
    __shared__ __attribute__((aligned(16))) float s_s2[64];
    __shared__ __attribute__((aligned(16))) float s_t2[64];
    __shared__ __attribute__((aligned(16))) float s_w3[64];
    __shared__ __attribute__((aligned(16))) float s_out[TILE * 4];
    __shared__ int s_slot;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = wave_id();
    const int j = lane & 31, h = lane >> 5;

    float wreg[64];
    load_w_slice(a.w1, wave, lane, 0, wreg);
    if (tid < C) { s_s1[tid] = a.s1[tid]; s_t1[tid] = a.t1[tid]; }
    if (tid < 64) { s_s2[tid] = a.s2[tid]; s_t2[tid] = a.t2[tid]; s_w3[tid] = a.w3[tid]; }
    for (int idx = tid; idx < 4 * 2 * 16 * 16; idx += 256) {
        const int q = idx & 15, o = (idx >> 4) & 15, hh = (idx >> 8) & 1, w = idx >> 9;
        const int c = 8 * (q >> 2) + 4 * hh + (q & 3);          // the MFMA accumulator's channel map
        s_w2[idx] = a.w2[(w * 16 + o) * 32 + c];
    }
    const float b3 = a.b3[wave];

    TileWalk walk(WALK_MOD8, a.tiles_per_frame * a.batch, nullptr, &s_slot);
    int tile;
    while (walk.next(tile)) {
        const int frame = tile / a.tiles_per_frame;
        const int n0 = (tile - frame * a.tiles_per_frame) * TILE;            // first valid-row index of the tile
        const float* hf = a.h + ((size_t)frame * a.n_per_frame + a.row_lo) * C;
#pragma unroll 4
        for (int q = 0; q < TILE / 4; ++q) {
            const int rl = wave * (TILE / 4) + q;
            int n = n0 + rl;
            n = n < a.n_valid ? n : a.n_valid - 1;
            *reinterpret_cast<f32x2*>(&s_a[rl * LDA + 2 * lane]) = load_row2(hf, n, lane);
        }
        __syncthreads();
#pragma unroll 1
        for (int rb = 0; rb < TILE / 32; ++rb) {
            if (n0 + rb * 32 >= a.n_valid) break;
            f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            mfma_rowblock(s_a, rb * 32, lane, wreg, acc);
            // Linear(128,32) epilogue: BN + ReLU on this lane's 16 hidden values
            float v[16];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int ch0 = 32 * wave + 8 * g + 4 * h;
                const f32x4 sc = *reinterpret_cast<const f32x4*>(&s_s1[ch0]);
                const f32x4 sh = *reinterpret_cast<const f32x4*>(&s_t1[ch0]);
                v[4 * g + 0] = fmaxf(acc[4 * g + 0] * sc.x + sh.x, 0.f);
                v[4 * g + 1] = fmaxf(acc[4 * g + 1] * sc.y + sh.y, 0.f);
                v[4 * g + 2] = fmaxf(acc[4 * g + 2] * sc.z + sh.z, 0.f);
                v[4 * g + 3] = fmaxf(acc[4 * g + 3] * sc.w + sh.w, 0.f);
            }
            // Linear(32,16): partial over this half's 16 inputs, then add the other half's partial
            const float* w2p = &s_w2[((wave * 2 + h) * 16) * 16];
            float y = 0.f;
#pragma unroll
            for (int o = 0; o < 16; ++o) {
                float p = 0.f;
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    const f32x4 ww = *reinterpret_cast<const f32x4*>(w2p + o * 16 + 4 * q4);
                    p += ww.x * v[4 * q4 + 0] + ww.y * v[4 * q4 + 1] + ww.z * v[4 * q4 + 2] + ww.w * v[4 * q4 + 3];
                }
                p += __shfl_xor(p, 32);
                const float z = fmaxf(p * s_s2[16 * wave + o] + s_t2[16 * wave + o], 0.f);
                y += s_w3[16 * wave + o] * z;                        // Linear(16,1)
            }
            y += b3;
            if (a.sigmoid) y = 1.0f / (1.0f + __expf(-y));
            if (h == 0) s_out[(rb * 32 + j) * 4 + wave] = y;
        }
        __syncthreads();
        if (tid < TILE) {
            const int n = n0 + tid;
            if (n < a.n_valid)
                *reinterpret_cast<f32x4*>(a.logits + ((size_t)frame * a.n_valid + n) * 4) =
                    *reinterpret_cast<const f32x4*>(&s_out[tid * 4]);
        }
        __syncthreads();
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
    ClsArgs a{};
    a.h = h; a.w1 = w1; a.s1 = s1; a.t1 = t1; a.w2 = w2; a.s2 = s2; a.t2 = t2; a.w3 = w3; a.b3 = b3;
    a.logits = logits;
    a.n_per_frame = (int)n_per_frame; a.row_lo = (int)row_lo; a.n_valid = (int)n_valid; a.batch = batch;
    a.tiles_per_frame = (int)((n_valid + TILE - 1) / TILE);
    a.sigmoid = sigmoid;
    long long n_tiles = (long long)a.tiles_per_frame * batch;
    long long g = n_tiles < 768 ? n_tiles : 768;                  // 3 resident workgroups per CU (45 KB LDS each)
    g = (g + 7) / 8 * 8;
    hipLaunchKernelGGL(k_classifier, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, a);
    EG_HIP_TRY(hipGetLastError());
    return EG_OK;
}
