// Train-mode classifier heads for gfx950: forward and backward of
//     h[node_type == 0]  ->  4 x [Linear(128,32)-BN-ReLU-Drop-Linear(32,16)-BN-ReLU-Drop-Linear(16,1)]  ->  cat
// (reference src/core/models.py:363-377 heads, :485 node-type filter, :488-490 cat) with BATCH statistics in both
// BatchNorm layers, as ONE stacked network: the four first layers are one [128 -> 128] product (4 x BatchNorm1d(32) on the
// stacked output IS BatchNorm1d(128) with stacked parameters), the four second layers one block-diagonal [128 -> 64]
// product, the third a 16-wide dot per head.
//
// What is kept for the backward: z1 [R,128] and z2 [R,64] (pre-BatchNorm activations of the two hidden layers, R = valid
// rows) and the BatchNorm statistics; the hidden activations h1 / h2 themselves are never written: every kernel that needs
// them recomputes BN + ReLU + Dropout (a counter-based mask, a pure function of (seed, element)) from z on load.
//
//   forward   k_lin128_map<stats>   z1 = h[valid rows] W1^T + b1, column sums of z1, z1^2          (fp32 MFMA 16x16x4)
//             k_cls_mid_fwd         h1 = drop(relu(bn1(z1)));  z2 = h1 W2^T + b2 per head, column sums of z2, z2^2
//             k_cls_out_fwd         h2 = drop(relu(bn2(z2)));  logit = h2 . w3 + b3
//   backward  k_cls_out_bwd_sums    g2 = dlogit w3 mask2:  sum g2, sum g2 xhat2 (-> dgamma2, dbeta2), dw3, db3
//             k_cls_mid_bwd         dz2 = bn2'(g2);  dW2 += dz2^T h1;  dh1 = dz2 W2   (pre-mask gradient of h1)
//             eg_bn_act_bwd         dz1 = bn1'(dh1 mask1), dgamma1, dbeta1                                  (train.hip)
//             k_lin128_map          dh[valid rows] = dz1 W1                                          (transposed W)
//             k_dweight_partial     dW1 = dz1^T h[valid rows]                                               (train.hip)
// Biases in front of a train-mode BatchNorm (b1, b2) have an identically zero gradient (the batch mean absorbs them).
// Every reduction is two-stage with a fixed order: bitwise reproducible, no float atomics.
#include <stdlib.h>

#include "train_common.h"

namespace eg {

typedef unsigned u32x4v __attribute__((ext_vector_type(4)));

constexpr int H1 = 128;             // 4 heads x 32
constexpr int H2 = 64;              // 4 heads x 16
constexpr int LDZ = H2 + 4;         // LDS row stride of a [64][64] tile
constexpr int CT_THREADS = 256;
constexpr int CT_MAX_BLOCKS = 1024; // partial slabs per kernel
constexpr int LIN_GRID = 512;        // k_lin128_map: 2 workgroups of 8 waves per CU (124 VGPRs; at 3 per CU it spills and is 20 % slower)

struct ClsBn {                      // per-channel vectors of one BatchNorm layer (device pointers)
    const float *mean, *invstd, *scale, *shift, *gamma;
};
struct ClsDrop {
    float p, inv_keep;
    unsigned long long seed;
    const unsigned long long* epoch;       // (train_common.h: seed + epoch is what the kernels hash with)
};
__device__ inline ClsDrop resolved(const ClsDrop& d) {
    ClsDrop r = d;
    r.seed = d.seed + epoch_now(d.epoch);
    return r;
}

// ---- z1 = x[in rows] W^T (+ bias) -> out[out rows]  (+ column sums of the result) ---------------------------------
// 8 waves, 64-row tiles that never straddle a frame; wave w owns output channels 16w..16w+15 (W slice in 32 VGPRs).
struct LinMapDims {
    int n_valid, tiles_per_frame, batch, transpose_w;
    int in_stride, in_lo, out_stride, out_lo;
};

// ACT: the input rows do not exist yet -- x = relu|id(dropout(z * scale + shift)) + residual is the activation pass of the
// GNN layer in front of the heads (eg_gcn_layer_train_fwd with out == NULL left z and the statistics).  The tiles then cover
// EVERY row of a frame (the heads' filter only masks the product's output rows), each row of x is written to `h` on the way
// into LDS, and the separate activation pass + its 512 B per row of re-read are gone.
struct LinAct {
    const float *z, *scale, *shift, *residual;
    float* h;
    ActArgs a;
    int h_skip_lo, h_skip_hi;       // rows [h_skip_lo, h_skip_hi) of every frame are NOT written to h (0, 0: all rows are): the heads' backward
                                    // recomputes them from z and the residual (eg_classifier_bwd_sums recompute_h), so only the rows the
                                    // heads' filter drops -- the coordinate rows the landmark MLP reads -- ever reach memory
};

template <bool STATS, bool ACT>
__global__ __launch_bounds__(512, 4) void k_lin128_map(const float* __restrict__ x, const float* __restrict__ W,
                                                       const float* __restrict__ bias, float* __restrict__ out,
                                                       float* __restrict__ partial, const LinMapDims a, const LinAct act_) {
    LinAct act = act_;
    act.a = resolved(act_.a);
    __shared__ __attribute__((aligned(16))) float s_a[TILE * LDA + 4];
    const int tid = threadIdx.x, lane_k = tid & 63, wave = wave_id();
    float wreg[32];
    load_w_slice16(W, wave, lane_k, a.transpose_w, wreg);
    const int ch_d = 16 * wave + 4 * (lane_k >> 4);
    const f32x4 bv = bias ? *reinterpret_cast<const f32x4*>(bias + ch_d) : f32x4{0.f, 0.f, 0.f, 0.f};
    float cs = 0.f, cq = 0.f;                              // STATS: thread -> channel tid & 127, rows 16 (tid >> 7) ..
    const int n_tiles = a.tiles_per_frame * a.batch;
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        int lane = lane_k;
        asm volatile("" : "+v"(lane));
        const int frame = tile / a.tiles_per_frame;
        const int n0 = (tile - frame * a.tiles_per_frame) * TILE;
        const int span = ACT ? a.in_stride : a.n_valid;    // rows of a frame the tiles walk over
        const int rows_here = (span - n0) < TILE ? (span - n0) : TILE;
        // rows [lo, hi) of the tile have an output row (ACT: the heads' row filter; otherwise all of them)
        const int lo = ACT ? (a.in_lo > n0 ? a.in_lo - n0 : 0) : 0;
        const int hi = ACT ? ((a.in_lo + a.n_valid - n0) < rows_here ? (a.in_lo + a.n_valid - n0) : rows_here) : rows_here;
        const size_t in_row0 = (size_t)frame * a.in_stride + (ACT ? 0 : a.in_lo) + n0;
        float* __restrict__ of = out + ((size_t)frame * a.out_stride + a.out_lo) * C + ((long long)n0 - (ACT ? a.in_lo : 0)) * C;
        const int rl0 = 8 * wave;
        const PairLane pl{lane >> 5, lane & 31};
        if (ACT) {
            f32x4 zz[4], rr[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int r = rl0 + 2 * k + pl.h;
                const size_t off = (in_row0 + (unsigned)(r < rows_here ? r : rows_here - 1)) * C + 4u * pl.q;
                zz[k] = ldnt4(act.z + off);
                rr[k] = act.residual ? ldnt4(act.residual + off) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
            const f32x4 sc = *reinterpret_cast<const f32x4*>(act.scale + 4 * pl.q);
            const f32x4 sh = *reinterpret_cast<const f32x4*>(act.shift + 4 * pl.q);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int r = rl0 + 2 * k + pl.h;
                const size_t off = (in_row0 + (unsigned)(r < rows_here ? r : rows_here - 1)) * C + 4u * pl.q;
                f32x4 v = zz[k] * sc + sh;
                if (act.a.p > 0.f) v *= keep_scale4(act.a.seed, (unsigned long long)off, act.a.p, act.a.inv_keep);
                if (act.a.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                v += rr[k];
                *reinterpret_cast<f32x4*>(act.h + off) = v;             // (a missing row: the last row's value once more)
                *reinterpret_cast<f32x4*>(&s_a[(rl0 + 2 * k + pl.h) * LDA + 4 * pl.q]) = v;
            }
        } else {
            const float* __restrict__ xf = x + in_row0 * C;
            f32x4 v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int r = rl0 + 2 * k + pl.h;
                v[k] = *reinterpret_cast<const f32x4*>(xf + ((unsigned)(r < rows_here ? r : rows_here - 1) * (unsigned)C + 4u * pl.q));
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
        {
            const int j = lane & 15, q4 = lane >> 4;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                f32x4v o;
                o.x = acc[b].x + bv.x; o.y = acc[b].y + bv.y; o.z = acc[b].z + bv.z; o.w = acc[b].w + bv.w;
                *reinterpret_cast<f32x4v*>(&s_a[(16 * b + j) * LDA + 16 * wave + 4 * q4]) = o;
            }
        }
        __syncthreads();
        if (ACT) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int r = rl0 + 2 * k + pl.h;
                if (r >= lo && r < hi)
                    *reinterpret_cast<f32x4*>(of + (long long)r * C + 4 * pl.q) = *reinterpret_cast<const f32x4*>(&s_a[r * LDA + 4 * pl.q]);
            }
        } else if (rl0 < rows_here) {                       // (uniform per wave) whole 512-B rows, a duplicate store for a missing row
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int r = rl0 + 2 * k + pl.h;
                const int rr = r < rows_here ? r : rows_here - 1;
                *reinterpret_cast<f32x4*>(of + (size_t)rr * C + 4 * pl.q) = *reinterpret_cast<const f32x4*>(&s_a[rr * LDA + 4 * pl.q]);
            }
        }
        if (STATS) {
            const int c = tid & 127, r0 = 16 * (tid >> 7);
#pragma unroll 4
            for (int r = r0; r < r0 + 16; ++r) {
                const float v = (r >= lo && r < hi) ? s_a[r * LDA + c] : 0.f;
                cs += v; cq += v * v;
            }
        }
        __syncthreads();
    }
    if (STATS) {
        float* red = s_a;                                   // [4 row groups][2][128]
        red[(tid >> 7) * 256 + (tid & 127)] = cs;
        red[(tid >> 7) * 256 + 128 + (tid & 127)] = cq;
        __syncthreads();
        if (tid < 256) partial[(size_t)blockIdx.x * 256 + tid] = red[tid] + red[256 + tid] + red[512 + tid] + red[768 + tid];
    }
}

// The ACT + STATS form as a kernel of its own: 4 waves (each owns two 16-channel groups of the product, W slices in 64 VGPRs),
// two workgroups per CU, and the 16 row loads per thread of the NEXT tile (z and the residual) issued right behind the first
// barrier of the current one -- their round trip runs under this tile's products and write-back instead of in front of the next
// tile's activation (k_lin128_map<true, true>, 8 waves without the prefetch: 1.07 ms at batch 32; this one: see DESIGN 3.3).
// Same tiles per workgroup, same MFMA chain per output block, same per-thread sums as k_lin128_map: the results are the bits
// eg_bn_act_fwd + eg_classifier_train_fwd give.
__global__ __launch_bounds__(256, 2) void k_act_lin128(const float* __restrict__ W, const float* __restrict__ bias, float* __restrict__ out,
                                                       float* __restrict__ partial, const LinMapDims a, const LinAct act_) {
    LinAct act = act_;
    act.a = resolved(act_.a);
    __shared__ __attribute__((aligned(16))) float s_a[TILE * LDA + 4];
    const int tid = threadIdx.x, lane_k = tid & 63, wave = wave_id();
    float wreg0[32], wreg1[32];
    load_w_slice16(W, 2 * wave, lane_k, 0, wreg0);
    load_w_slice16(W, 2 * wave + 1, lane_k, 0, wreg1);
    const int ch_d = 32 * wave + 4 * (lane_k >> 4);
    const f32x4 bv0 = bias ? *reinterpret_cast<const f32x4*>(bias + ch_d) : f32x4{0.f, 0.f, 0.f, 0.f};
    const f32x4 bv1 = bias ? *reinterpret_cast<const f32x4*>(bias + ch_d + 16) : f32x4{0.f, 0.f, 0.f, 0.f};
    float cs[2] = {0.f, 0.f}, cq[2] = {0.f, 0.f};          // thread -> channel tid & 127, rows 16 g .. 16 g + 15, g = (tid >> 7) + 2 u
    const int n_tiles = a.tiles_per_frame * a.batch;
    const PairLane pl{lane_k >> 5, lane_k & 31};
    const int rl0 = 16 * wave;
    f32x4 pz[8], pr[8];
    auto issue = [&](int tile) {
        const int frame = tile / a.tiles_per_frame;
        const int n0 = (tile - frame * a.tiles_per_frame) * TILE;
        const int rows_here = (a.in_stride - n0) < TILE ? (a.in_stride - n0) : TILE;
        const size_t in_row0 = (size_t)frame * a.in_stride + n0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int r = rl0 + 2 * k + pl.h;
            const size_t off = (in_row0 + (unsigned)(r < rows_here ? r : rows_here - 1)) * C + 4u * pl.q;
            pz[k] = ldnt4(act.z + off);
            pr[k] = act.residual ? ldnt4(act.residual + off) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    if ((int)blockIdx.x < n_tiles) issue(blockIdx.x);
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        int lane = lane_k;
        asm volatile("" : "+v"(lane));
        const int frame = tile / a.tiles_per_frame;
        const int n0 = (tile - frame * a.tiles_per_frame) * TILE;
        const int rows_here = (a.in_stride - n0) < TILE ? (a.in_stride - n0) : TILE;
        const int lo = a.in_lo > n0 ? a.in_lo - n0 : 0;
        const int hi = (a.in_lo + a.n_valid - n0) < rows_here ? (a.in_lo + a.n_valid - n0) : rows_here;
        const size_t in_row0 = (size_t)frame * a.in_stride + n0;
        float* __restrict__ of = out + ((size_t)frame * a.out_stride + a.out_lo) * C + ((long long)n0 - a.in_lo) * C;
        {
            const f32x4 sc = *reinterpret_cast<const f32x4*>(act.scale + 4 * pl.q);
            const f32x4 sh = *reinterpret_cast<const f32x4*>(act.shift + 4 * pl.q);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int r = rl0 + 2 * k + pl.h;
                const size_t off = (in_row0 + (unsigned)(r < rows_here ? r : rows_here - 1)) * C + 4u * pl.q;
                f32x4 v = pz[k] * sc + sh;
                if (act.a.p > 0.f) v *= keep_scale4(act.a.seed, (unsigned long long)off, act.a.p, act.a.inv_keep);
                if (act.a.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                v += pr[k];
                const int rf = n0 + (r < rows_here ? r : rows_here - 1);
                if (rf < act.h_skip_lo || rf >= act.h_skip_hi)
                    *reinterpret_cast<f32x4*>(act.h + off) = v;         // (a missing row: the last row's value once more)
                *reinterpret_cast<f32x4*>(&s_a[r * LDA + 4 * pl.q]) = v;
            }
        }
        __syncthreads();
        if (tile + (int)gridDim.x < n_tiles) issue(tile + gridDim.x);
        f32x4v acc0[4], acc1[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) acc0[b] = acc1[b] = f32x4v{0.f, 0.f, 0.f, 0.f};
        mfma16_pair<false>(s_a, 0, lane, wreg0, acc0[0], acc0[1]);
        mfma16_pair<false>(s_a, 0, lane, wreg1, acc1[0], acc1[1]);
        if (rows_here > 32) {
            mfma16_pair<false>(s_a, 32, lane, wreg0, acc0[2], acc0[3]);
            mfma16_pair<false>(s_a, 32, lane, wreg1, acc1[2], acc1[3]);
        }
        __syncthreads();
        {
            const int j = lane & 15, q4 = lane >> 4;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                f32x4v o0, o1;
                o0.x = acc0[b].x + bv0.x; o0.y = acc0[b].y + bv0.y; o0.z = acc0[b].z + bv0.z; o0.w = acc0[b].w + bv0.w;
                o1.x = acc1[b].x + bv1.x; o1.y = acc1[b].y + bv1.y; o1.z = acc1[b].z + bv1.z; o1.w = acc1[b].w + bv1.w;
                *reinterpret_cast<f32x4v*>(&s_a[(16 * b + j) * LDA + 32 * wave + 4 * q4]) = o0;
                *reinterpret_cast<f32x4v*>(&s_a[(16 * b + j) * LDA + 32 * wave + 16 + 4 * q4]) = o1;
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int r = rl0 + 2 * k + pl.h;
            if (r >= lo && r < hi)
                *reinterpret_cast<f32x4*>(of + (long long)r * C + 4 * pl.q) = *reinterpret_cast<const f32x4*>(&s_a[r * LDA + 4 * pl.q]);
        }
        {
            const int c = tid & 127;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int r0 = 16 * ((tid >> 7) + 2 * u);
#pragma unroll 4
                for (int r = r0; r < r0 + 16; ++r) {
                    const float v = (r >= lo && r < hi) ? s_a[r * LDA + c] : 0.f;
                    cs[u] += v; cq[u] += v * v;
                }
            }
        }
        __syncthreads();
    }
    float* red = s_a;                                       // [4 row groups][2][128]
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int g = (tid >> 7) + 2 * u;
        red[g * 256 + (tid & 127)] = cs[u];
        red[g * 256 + 128 + (tid & 127)] = cq[u];
    }
    __syncthreads();
    partial[(size_t)blockIdx.x * 256 + tid] = red[tid] + red[256 + tid] + red[512 + tid] + red[768 + tid];
}

// ---- hidden activation of a BN-ReLU-Dropout block, recomputed from z ------------------------------------------------
// (4 consecutive channels; idx = element index of the first, a multiple of 4)
__device__ inline f32x4 hidden_act4(const f32x4& z, const f32x4& scale, const f32x4& shift, const ClsDrop& d, unsigned long long idx) {
    f32x4 v = z * scale + shift;
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = fmaxf(v[u], 0.f);
    if (d.p > 0.f) v *= keep_scale4(d.seed, idx, d.p, d.inv_keep);
    return v;
}

// ---- forward, second layers: z2 = h1 W2^T + b2 (block diagonal: head = wave), column sums of z2 -----------------------
__global__ __launch_bounds__(CT_THREADS) void k_cls_mid_fwd(const float* __restrict__ z1, long long rows, const float* __restrict__ w2,
                                                            const float* __restrict__ b2, const ClsBn bn1, const ClsDrop d1_,
                                                            float* __restrict__ z2, float* __restrict__ partial) {
    const ClsDrop d1 = resolved(d1_);
    __shared__ __attribute__((aligned(16))) float s_h[TILE * LDA];
    __shared__ __attribute__((aligned(16))) float s_z[TILE * LDZ];
    const int tid = threadIdx.x, lane = tid & 63, head = wave_id();
    // MFMA 16x16x4 A operand: lane (o = l & 15, kq = l >> 4) holds W2[head][o][8 kq + t], t = 0..7
    float w2a[8];
    {
        const f32x4* p = reinterpret_cast<const f32x4*>(w2 + (size_t)(head * 16 + (lane & 15)) * 32 + 8 * (lane >> 4));
        const f32x4 q0 = p[0], q1 = p[1];
        w2a[0] = q0.x; w2a[1] = q0.y; w2a[2] = q0.z; w2a[3] = q0.w; w2a[4] = q1.x; w2a[5] = q1.y; w2a[6] = q1.z; w2a[7] = q1.w;
    }
    const f32x4 b2v = *reinterpret_cast<const f32x4*>(b2 + head * 16 + 4 * (lane >> 4));
    float cs = 0.f, cq = 0.f;                               // thread -> channel tid & 63, rows 16 (tid >> 6) ..
    const long long n_tiles = (rows + TILE - 1) / TILE;
    // the 8 row loads per thread of a tile are issued one tile ahead (behind the first barrier of the tile before): their round
    // trip runs under that tile's products and write-back instead of in front of this tile's
    f32x4 pz1[8];
    auto issue = [&](long long tile) {
        const long long row0 = tile * TILE;
        const int rows_here = (int)((rows - row0) < TILE ? (rows - row0) : TILE);
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int e = tid + CT_THREADS * it, r = e >> 5, c4 = (e & 31) * 4;
            pz1[it] = r < rows_here ? ldnt4(z1 + (size_t)(row0 + r) * H1 + c4) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    if ((long long)blockIdx.x < n_tiles) issue(blockIdx.x);
    for (long long tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const long long row0 = tile * TILE;
        const int rows_here = (int)((rows - row0) < TILE ? (rows - row0) : TILE);
        {
            const int c4 = (tid & 31) * 4;                   // (a thread always owns the same 4 channels)
            const f32x4 sc = *reinterpret_cast<const f32x4*>(bn1.scale + c4), sh = *reinterpret_cast<const f32x4*>(bn1.shift + c4);
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int r = (tid + CT_THREADS * it) >> 5;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (r < rows_here) v = hidden_act4(pz1[it], sc, sh, d1, (unsigned long long)(row0 + r) * H1 + c4);
                *reinterpret_cast<f32x4*>(&s_h[r * LDA + c4]) = v;
            }
        }
        __syncthreads();
        if (tile + gridDim.x < n_tiles) issue(tile + gridDim.x);
        const int j = lane & 15, kq = lane >> 4;
#pragma unroll
        for (int b4 = 0; b4 < 4; ++b4) {
            const f32x4* hp = reinterpret_cast<const f32x4*>(&s_h[(16 * b4 + j) * LDA + 32 * head + 8 * kq]);
            const f32x4 h0 = hp[0], h1v = hp[1];
            f32x4v z = {0.f, 0.f, 0.f, 0.f};
            z = __builtin_amdgcn_mfma_f32_16x16x4f32(w2a[0], h0.x, z, 0, 0, 0);
            z = __builtin_amdgcn_mfma_f32_16x16x4f32(w2a[1], h0.y, z, 0, 0, 0);
            z = __builtin_amdgcn_mfma_f32_16x16x4f32(w2a[2], h0.z, z, 0, 0, 0);
            z = __builtin_amdgcn_mfma_f32_16x16x4f32(w2a[3], h0.w, z, 0, 0, 0);
            z = __builtin_amdgcn_mfma_f32_16x16x4f32(w2a[4], h1v.x, z, 0, 0, 0);
            z = __builtin_amdgcn_mfma_f32_16x16x4f32(w2a[5], h1v.y, z, 0, 0, 0);
            z = __builtin_amdgcn_mfma_f32_16x16x4f32(w2a[6], h1v.z, z, 0, 0, 0);
            z = __builtin_amdgcn_mfma_f32_16x16x4f32(w2a[7], h1v.w, z, 0, 0, 0);
            // D: lane (row j, q = kq), reg e -> z2[16 b4 + j][16 head + 4 q + e]
            f32x4v o;
            o.x = z.x + b2v.x; o.y = z.y + b2v.y; o.z = z.z + b2v.z; o.w = z.w + b2v.w;
            *reinterpret_cast<f32x4v*>(&s_z[(16 * b4 + j) * LDZ + 16 * head + 4 * kq]) = o;
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int e = tid + CT_THREADS * it, r = e >> 4, c4 = (e & 15) * 4;
            if (r < rows_here)
                *reinterpret_cast<f32x4*>(z2 + (size_t)(row0 + r) * H2 + c4) = *reinterpret_cast<const f32x4*>(&s_z[r * LDZ + c4]);
        }
        {
            const int c = tid & 63, r0 = 16 * (tid >> 6);
#pragma unroll 4
            for (int r = r0; r < r0 + 16; ++r) {
                const float v = r < rows_here ? s_z[r * LDZ + c] : 0.f;
                cs += v; cq += v * v;
            }
        }
        __syncthreads();
    }
    float* red = s_z;                                       // [4 row groups][2][64]
    red[(tid >> 6) * 128 + (tid & 63)] = cs;
    red[(tid >> 6) * 128 + 64 + (tid & 63)] = cq;
    __syncthreads();
    if (tid < 128) partial[(size_t)blockIdx.x * 128 + tid] = red[tid] + red[128 + tid] + red[256 + tid] + red[384 + tid];
}

// ---- forward, third layers: one thread per (row, head) -----------------------------------------------------------------
__global__ __launch_bounds__(CT_THREADS) void k_cls_out_fwd(const float* __restrict__ z2, long long rows, const ClsBn bn2, const ClsDrop d2_,
                                                            const float* __restrict__ w3, const float* __restrict__ b3, int sigmoid,
                                                            float* __restrict__ logits) {
    const ClsDrop d2 = resolved(d2_);
    const long long n = rows * 4;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int head = (int)(i & 3);
        const long long row = i >> 2;
        const float* zp = z2 + (size_t)row * H2 + 16 * head;
        float y = b3[head];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 zz = *reinterpret_cast<const f32x4*>(zp + 4 * q);
            const f32x4 sc = *reinterpret_cast<const f32x4*>(bn2.scale + 16 * head + 4 * q);
            const f32x4 sh = *reinterpret_cast<const f32x4*>(bn2.shift + 16 * head + 4 * q);
            const f32x4 ww = *reinterpret_cast<const f32x4*>(w3 + 16 * head + 4 * q);
            const unsigned long long idx = (unsigned long long)row * H2 + 16 * head + 4 * q;
            const f32x4 hh = hidden_act4(zz, sc, sh, d2, idx);
            y += ww.x * hh.x + ww.y * hh.y + ww.z * hh.z + ww.w * hh.w;
        }
        if (sigmoid) y = 1.0f / (1.0f + __expf(-y));
        logits[i] = y;
    }
}

// ---- backward of the third layers: column sums ----------------------------------------------------------------------------
// thread = (row, head): g2[c] = dlogit * w3[c] * mask2[c];  sums over rows of g2, g2 * xhat2, dlogit * h2 (= dw3) and dlogit (= db3).
// partial layout per block: [3][64] + [4]
constexpr int OUT_SUMS = 3 * H2 + 4;
__global__ __launch_bounds__(CT_THREADS) void k_cls_out_bwd_sums(const float* __restrict__ dlogits, const float* __restrict__ z2, long long rows,
                                                                 const ClsBn bn2, const ClsDrop d2_, const float* __restrict__ w3,
                                                                 float* __restrict__ partial) {
    const ClsDrop d2 = resolved(d2_);
    __shared__ float red[4][CT_THREADS / 4][17];            // [head][thread of the head][channel]
    const int tid = threadIdx.x, head = tid & 3;
    float mean[16], istd[16], scl[16], sft[16], w3v[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        const int ch = 16 * head + c;
        mean[c] = bn2.mean[ch]; istd[c] = bn2.invstd[ch]; scl[c] = bn2.scale[ch]; sft[c] = bn2.shift[ch]; w3v[c] = w3[ch];
    }
    float sg[16], sgx[16], sw[16], sb = 0.f;
#pragma unroll
    for (int c = 0; c < 16; ++c) sg[c] = sgx[c] = sw[c] = 0.f;
    const long long n = rows * 4;
    for (long long i = (long long)blockIdx.x * blockDim.x + tid; i < n; i += (long long)gridDim.x * blockDim.x) {
        const long long row = i >> 2;                       // (i & 3) == head: blockDim and gridDim * blockDim are multiples of 4
        const float dl = dlogits[i];
        const float* zp = z2 + (size_t)row * H2 + 16 * head;
        sb += dl;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 zq = *reinterpret_cast<const f32x4*>(zp + 4 * q);
            const f32x4 kq = d2.p > 0.f ? keep_scale4(d2.seed, (unsigned long long)row * H2 + 16 * head + 4 * q, d2.p, d2.inv_keep)
                                        : f32x4{1.f, 1.f, 1.f, 1.f};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int c = 4 * q + u;
                const float zz = zq[u];
                const float xh = (zz - mean[c]) * istd[c];
                const float v = zz * scl[c] + sft[c];
                const float k = kq[u];
                const float m = v > 0.f ? k : 0.f;
                const float g = dl * w3v[c] * m;
                sg[c] += g; sgx[c] += g * xh; sw[c] += dl * v * m;
            }
        }
    }
    // block reduction over the 64 threads of each head, in a fixed order
    float* out = partial + (size_t)blockIdx.x * OUT_SUMS;
#pragma unroll 1
    for (int q = 0; q < 3; ++q) {
        __syncthreads();
#pragma unroll
        for (int c = 0; c < 16; ++c) red[head][tid >> 2][c] = q == 0 ? sg[c] : (q == 1 ? sgx[c] : sw[c]);
        __syncthreads();
        if (tid < H2) {
            const int hd = tid >> 4, c = tid & 15;
            float s = 0.f;
            for (int t = 0; t < CT_THREADS / 4; ++t) s += red[hd][t][c];
            out[q * H2 + tid] = s;
        }
    }
    __syncthreads();
    red[head][tid >> 2][16] = sb;
    __syncthreads();
    if (tid < 4) {
        float s = 0.f;
        for (int t = 0; t < CT_THREADS / 4; ++t) s += red[tid][t][16];
        out[3 * H2 + tid] = s;
    }
}

// ---- backward of the second layers ------------------------------------------------------------------------------------------
// per 64-row tile: dz2 = gamma2 invstd2 (g2 - mean(g2) - xhat2 mean(g2 xhat2));  dW2[head] += dz2^T h1 (K = rows, MFMA);
// dh1 = dz2 W2 (MFMA), written as whole rows.  tot = the reduced sums of k_cls_out_bwd_sums (double).
// (3 waves per SIMD = the 3 workgroups per CU the 768-block grid counts on: 150 VGPRs, no spills; left to itself the
//  compiler took 182 and the third of the grid that was not resident ran as a tail: 1.06 -> 0.71 ms at B = 32)
#ifndef MID_BWD_WGS
#define MID_BWD_WGS 2               // workgroups per CU (register budget 256 / 168 VGPRs for 2 / 3)
#endif
// MASKED: what is written is g1 = dh1 * mask1 (the masked gradient the first layers' BatchNorm backward starts from -- this kernel
// forms it anyway for its sums), so that k_cls_first_bwd need not regenerate the dropout / ReLU mask per element.
template <bool MASKED>
__global__ __launch_bounds__(CT_THREADS, MID_BWD_WGS) void k_cls_mid_bwd(const float* __restrict__ dlogits, const float* __restrict__ z2,
                                                            const float* __restrict__ z1, long long rows, const ClsBn bn1, const ClsDrop d1_,
                                                            const ClsBn bn2, const ClsDrop d2_, const float* __restrict__ w2,
                                                            const float* __restrict__ w3, const double* __restrict__ tot,
                                                            float* __restrict__ dh1, float* __restrict__ partial_dw2,
                                                            float* __restrict__ partial_bn1) {
    const ClsDrop d1 = resolved(d1_), d2 = resolved(d2_);
    __shared__ __attribute__((aligned(16))) float s_h[TILE * LDA];      // h1 tile, then the dh1 tile
    __shared__ __attribute__((aligned(16))) float s_dz[TILE * LDZ];
    __shared__ float s_c[3][H2];                                          // a2 = gamma2 invstd2, mean g2, mean g2 xhat2
    const int tid = threadIdx.x, lane = tid & 63, head = wave_id();
    if (tid < H2) {
        const double inv_n = 1.0 / (double)rows;
        s_c[0][tid] = bn2.gamma[tid] * bn2.invstd[tid];
        s_c[1][tid] = (float)(tot[tid] * inv_n);
        s_c[2][tid] = (float)(tot[H2 + tid] * inv_n);
    }
    // dh1 = dz2 W2: MFMA A operand lane (m = l & 15, kq = l >> 4) holds W2[head][4 t + kq][16 ib + m], t = 0..3, ib = 0, 1
    float wt[2][4];
#pragma unroll
    for (int ib = 0; ib < 2; ++ib)
#pragma unroll
        for (int t = 0; t < 4; ++t) wt[ib][t] = w2[(size_t)(head * 16 + 4 * t + (lane >> 4)) * 32 + 16 * ib + (lane & 15)];
    f32x4v dw[2] = {f32x4v{0.f, 0.f, 0.f, 0.f}, f32x4v{0.f, 0.f, 0.f, 0.f}};      // dW2[head][4 q + e][16 ib + i]
    // Sums of the FIRST layers' BatchNorm backward, taken here where dh1 is produced (g1 = dh1 * mask1: sum g1, sum g1 xhat1 --
    // what a separate pass over dh1 and z1 computed before): a thread always owns the same 4 channels, c4 = 4 (tid & 31)
    f32x4 sg1 = {0.f, 0.f, 0.f, 0.f}, sx1 = sg1;
    const int c4s = (tid & 31) * 4;
    const long long n_tiles = (rows + TILE - 1) / TILE;
    // The row loads of a tile (z1: 8, z2: 4 float4 per thread, 4 dlogits) are issued one tile ahead, right behind the first
    // barrier of the tile before: their round trip runs under that tile's products instead of in front of this one's.
    f32x4 pz1[8], pz2[4], cz1[8];
    float pdl[4];
    auto issue = [&](long long tile) {
        const long long row0 = tile * TILE;
        const int rows_here = (int)((rows - row0) < TILE ? (rows - row0) : TILE);
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int e = tid + CT_THREADS * it, r = e >> 5, c4 = (e & 31) * 4;
            pz1[it] = r < rows_here ? *reinterpret_cast<const f32x4*>(z1 + (size_t)(row0 + r) * H1 + c4) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int e = tid + CT_THREADS * it, r = e >> 4, c4 = (e & 15) * 4;
            const bool ok = r < rows_here;
            pz2[it] = ok ? *reinterpret_cast<const f32x4*>(z2 + (size_t)(row0 + r) * H2 + c4) : f32x4{0.f, 0.f, 0.f, 0.f};
            pdl[it] = ok ? dlogits[(row0 + r) * 4 + (c4 >> 4)] : 0.f;
        }
    };
    if ((long long)blockIdx.x < n_tiles) issue(blockIdx.x);
    __syncthreads();
    for (long long tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const long long row0 = tile * TILE;
        const int rows_here = (int)((rows - row0) < TILE ? (rows - row0) : TILE);
        unsigned keep1 = 0u;                           // dropout keep flags of this thread's 32 elements of the tile (bit 4 it + u)
        {   // h1 tile from the prefetched z1 (rows >= rows_here are zero)
            int c4o = c4s;
            asm volatile("" : "+v"(c4o));
            const f32x4 sc = *reinterpret_cast<const f32x4*>(bn1.scale + c4o), sh = *reinterpret_cast<const f32x4*>(bn1.shift + c4o);
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int e = tid + CT_THREADS * it, r = e >> 5, c4 = (e & 31) * 4;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (r < rows_here) {                   // hidden_act4 with the dropout keep flags kept (4 bits per float4) for the end of the tile
                    v = pz1[it] * sc + sh;
#pragma unroll
                    for (int u = 0; u < 4; ++u) v[u] = fmaxf(v[u], 0.f);
                    if (d1.p > 0.f) {
#ifdef EG_ABL_HASH_MID        // (timing-only)
                        const f32x4 kk = f32x4{d1.inv_keep, 0.f, d1.inv_keep, 0.f};
#else
                        const f32x4 kk = keep_scale4(d1.seed, (unsigned long long)(row0 + r) * H1 + c4, d1.p, d1.inv_keep);
#endif
                        v *= kk;
#pragma unroll
                        for (int u = 0; u < 4; ++u) keep1 |= (kk[u] != 0.f ? 1u : 0u) << (4 * it + u);
                    }
                }
                *reinterpret_cast<f32x4*>(&s_h[r * LDA + c4]) = v;
                cz1[it] = pz1[it];                     // this tile's z1 stays in registers for the BatchNorm sums at the end of the tile
            }
        }
#pragma unroll
        for (int it = 0; it < 4; ++it) {                                 // dz2 tile: rows >= rows_here are zero
            const int e = tid + CT_THREADS * it, r = e >> 4, c4 = (e & 15) * 4;
            f32x4 o = {0.f, 0.f, 0.f, 0.f};
            if (r < rows_here) {
                const long long row = row0 + r;
                const float dl = pdl[it];
                const f32x4 zz = pz2[it];
                const f32x4 mn = *reinterpret_cast<const f32x4*>(bn2.mean + c4), is = *reinterpret_cast<const f32x4*>(bn2.invstd + c4);
                const f32x4 sc = *reinterpret_cast<const f32x4*>(bn2.scale + c4), sh = *reinterpret_cast<const f32x4*>(bn2.shift + c4);
                const f32x4 ww = *reinterpret_cast<const f32x4*>(w3 + c4);
#ifdef EG_ABL_HASH_MID
                const f32x4 kk = f32x4{d2.inv_keep, 0.f, d2.inv_keep, 0.f}; (void)row;
#else
                const f32x4 kk = d2.p > 0.f ? keep_scale4(d2.seed, (unsigned long long)row * H2 + c4, d2.p, d2.inv_keep)
                                            : f32x4{1.f, 1.f, 1.f, 1.f};
#endif
                float ov[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int ch = c4 + u;
                    const float xh = (zz[u] - mn[u]) * is[u];
                    const float v = zz[u] * sc[u] + sh[u];
                    const float k = kk[u];
                    const float g = v > 0.f ? dl * ww[u] * k : 0.f;
                    ov[u] = s_c[0][ch] * (g - s_c[1][ch] - xh * s_c[2][ch]);
                }
                o = f32x4{ov[0], ov[1], ov[2], ov[3]};
            }
            *reinterpret_cast<f32x4*>(&s_dz[r * LDZ + c4]) = o;
        }
        __syncthreads();
        if (tile + gridDim.x < n_tiles) issue(tile + gridDim.x);
        const int i16 = lane & 15, kq = lane >> 4;
        // dW2[head] += dz2[:, head]^T h1[:, head]:  A[o][k] = dz2[4 s + k][16 head + o],  B[k][i] = h1[4 s + k][32 head + 16 ib + i]
#pragma unroll 4
        for (int s = 0; s < 16; ++s) {
            const float av = s_dz[(4 * s + kq) * LDZ + 16 * head + i16];
            const float b0 = s_h[(4 * s + kq) * LDA + 32 * head + i16];
            const float b1 = s_h[(4 * s + kq) * LDA + 32 * head + 16 + i16];
            dw[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b0, dw[0], 0, 0, 0);
            dw[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b1, dw[1], 0, 0, 0);
        }
        __syncthreads();                                                  // every wave is done with h1: the tile becomes dh1
        // dh1[r][32 head + 16 ib + m] = sum_o dz2[r][16 head + o] W2[head][o][16 ib + m]:  B[k][n] = dz2[16 b4 + n][16 head + 4 t + k]
#pragma unroll
        for (int b4 = 0; b4 < 4; ++b4) {
            f32x4v acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float bv = s_dz[(16 * b4 + i16) * LDZ + 16 * head + 4 * t + kq];
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wt[0][t], bv, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wt[1][t], bv, acc1, 0, 0, 0);
            }
            // D: lane (row n = i16, q = kq), reg e -> dh1[16 b4 + n][32 head + 16 ib + 4 q + e]
            *reinterpret_cast<f32x4v*>(&s_h[(16 * b4 + i16) * LDA + 32 * head + 4 * kq]) = acc0;
            *reinterpret_cast<f32x4v*>(&s_h[(16 * b4 + i16) * LDA + 32 * head + 16 + 4 * kq]) = acc1;
        }
        __syncthreads();
        // (this thread's per-channel constants, fetched per tile -- L1 hits -- so that they are not live across the products above)
        int c4o = c4s;
        asm volatile("" : "+v"(c4o));
        const f32x4 mn1 = *reinterpret_cast<const f32x4*>(bn1.mean + c4o), is1 = *reinterpret_cast<const f32x4*>(bn1.invstd + c4o);
        const f32x4 sc1 = *reinterpret_cast<const f32x4*>(bn1.scale + c4o), sh1 = *reinterpret_cast<const f32x4*>(bn1.shift + c4o);
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int e = tid + CT_THREADS * it, r = e >> 5, c4 = (e & 31) * 4;
            if (r < rows_here) {
                const f32x4 dv = *reinterpret_cast<const f32x4*>(&s_h[r * LDA + c4]);
                if (!MASKED) *reinterpret_cast<f32x4*>(dh1 + (size_t)(row0 + r) * H1 + c4) = dv;
                // z1 of this element again for the mask and xhat: from the registers it arrived in (re-read from memory it was
                // 1.18 GB of HBM traffic per launch at batch 32 -- the tile had left the L2 by then, profiles/r04_train_pmc.json)
                const size_t off = (size_t)(row0 + r) * H1 + c4;
                const f32x4 zz = cz1[it];
                f32x4 kk = {1.f, 1.f, 1.f, 1.f};       // (the flags of the tile's first phase: no second hash per element)
                if (d1.p > 0.f) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) kk[u] = ((keep1 >> (4 * it + u)) & 1u) ? d1.inv_keep : 0.f;
                }
                f32x4 gv;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float v = zz[u] * sc1[u] + sh1[u];                      // (hidden_act4's expression: the same mask)
                    const float g = v > 0.f ? dv[u] * kk[u] : 0.f;
                    gv[u] = g;
                    sg1[u] += g;
                    sx1[u] += g * ((zz[u] - mn1[u]) * is1[u]);
                }
                if (MASKED) *reinterpret_cast<f32x4*>(dh1 + off) = gv;
            }
        }
        __syncthreads();
    }
    {   // per-block partial [2][128]: the 8 threads that share a channel group are added in a fixed order
        float* red = s_h;                                                     // [8][2][128]
        const int rg = tid >> 5;
#pragma unroll
        for (int u = 0; u < 4; ++u) { red[(rg * 2 + 0) * H1 + c4s + u] = sg1[u]; red[(rg * 2 + 1) * H1 + c4s + u] = sx1[u]; }
        __syncthreads();
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) t += red[k * 2 * H1 + tid];
        partial_bn1[(size_t)blockIdx.x * 2 * H1 + tid] = t;                    // tid < 256 = 2 * 128
    }
    // per-block dW2 partial [4][16][32]: lane (i = l & 15, q), reg e -> [head][4 q + e][16 ib + i]
    float* p = partial_dw2 + (size_t)blockIdx.x * (4 * 16 * 32) + head * 512;
#pragma unroll
    for (int ib = 0; ib < 2; ++ib)
#pragma unroll
        for (int e = 0; e < 4; ++e) p[(4 * (lane >> 4) + e) * 32 + 16 * ib + (lane & 15)] = dw[ib][e];
}

// ---- backward of the FIRST layers, one kernel: dz1 = bn1'(dh1 mask1) formed on load, never written; both products on it:
//          dW1 += dz1^T h[valid rows]      (K = rows)          waves 0..3: 32 output channels x 128 inputs each
//          dh[valid rows] = dz1 W1         (K = 128)           waves 4..7: 32 columns of dh each, two 32-row blocks
// (before: an apply + dW pass that wrote dz1, and a plain-rows product that read it back: 2.4 GB of traffic and one launch more
//  at batch 32).  64-row tiles, 512 threads, one workgroup per CU; the 12 row loads per thread of the next tile are issued
// before the products of the current one.  tot = sum g1, sum g1 xhat1 (double [2][128]) from k_cls_mid_bwd's partials.
constexpr int FB_BLOCKS = 256;
struct FirstBwdArgs {
    const float *dh1, *z1, *h, *W1;
    const double* tot;
    long long rows;
    RowMap xm;
    ClsDrop d1;
    float* dh;
    float* partial_dw1;
    int masked;                 // dh1 holds g1 = dh1 * mask1 already (k_cls_mid_bwd<true>)
    // SUMS: the BatchNorm-backward sums of the GNN layer in front of the heads (sum g, sum g xhat over the rows dh is written to,
    // g = dh * that layer's dropout / ReLU mask), taken in the tile's store phase where dh passes through registers anyway:
    // the layer's own sums pass (a read of dh and z: 2.4 GB at batch 32) then only has the filtered-out rows left
    const float *lz, *lmean, *linvstd, *lgamma, *lbeta;      // the layer's z [batch * stride, 128] and BatchNorm vectors
    const float *lscale, *lshift;                            // its forward scale / shift (hrec)
    int hrec;                                                // `h` holds the layer's RESIDUAL rows (NULL: none) and the h tile is recomputed as
                                                             // act(z) + residual with the forward's expression: h itself was never written
    ActArgs la;                                              // its (relu, p, seed)
    float* partial_lsums;                                    // [blocks][2][128]
    unsigned bytes_c, bytes_m;                               // sizes of the compact [rows,128] / unfiltered [batch * stride,128] arrays
    int direct;                                              // both below 2 GB: buffer accesses off whole-array descriptors
};

#ifndef FB_ABL
#define FB_ABL 0            // timing-only ablations (bits): 1 no weight-gradient product, 2 no dh product; direct form also: 4 no dz1 arithmetic,
#endif                      // 8 no dh store, 16 no LDS tile writes, 32 no row loads, 64 no layer sums
// The two roles run separate instantiations of the tile loop (so that neither carries the other's persistent registers: the
// [32 x 128] accumulators of the weight gradient / the W1 slice); both execute the same two workgroup barriers per tile.
template <bool GEMM2, bool SUMS>
__device__ inline void first_bwd_role_flat(const FirstBwdArgs& a, float* s_g, float* s_x, float* s_o, const float* s_c, int wave) {
    const int tid = threadIdx.x, lane_k = tid & 63;
    const int c4 = (tid & 31) * 4;
    float wreg[GEMM2 ? 64 : 1];                   // waves 4..7: W1[64 kh + s][32 (wave - 4) + i] (the transposed slice)
    f32x16 acc[GEMM2 ? 1 : 4];                    // waves 0..3: dW1[32 wave + m][32 jb + n]
    if constexpr (GEMM2) load_w_slice(a.W1, wave - 4, lane_k, 1, wreg);
    else {
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[b][e] = 0.f;
    }
    const long long rows = a.rows;
    const long long n_tiles = (rows + TILE - 1) / TILE;
    f32x4 pd[4], pz[4], px[4];
    f32x4 pl[SUMS ? 4 : 1], cl[SUMS ? 4 : 1];     // SUMS: the layer's z rows, prefetched with the others / of the current tile
    f32x4 lsg = {0.f, 0.f, 0.f, 0.f}, lsx = lsg;
    // Row of the unfiltered array behind compact row r of tile t: the tile's first row is decoded once (a 32-bit division on the
    // scalar unit), a row of the tile lies in that frame or the next one (n_valid >= 64 is checked on the host).  A 64-bit
    // division per row -- map_row -- cost more instructions than the rest of the tile's address arithmetic together.
    auto tile_map = [&](long long t, int& f0, int& in0) {
        const unsigned r0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(t * TILE));
        f0 = (int)(r0 / (unsigned)a.xm.n_valid);
        in0 = (int)(r0 - (unsigned)f0 * (unsigned)a.xm.n_valid);
    };
    auto mapped = [&](int f0, int in0, int rl) -> unsigned {
        const int in = in0 + rl;
        const bool next = in >= a.xm.n_valid;
        return (unsigned)((f0 + (next ? 1 : 0)) * a.xm.stride + a.xm.lo + (next ? in - a.xm.n_valid : in));
    };
    auto issue = [&](long long t) {
        int f0, in0;
        tile_map(t, f0, in0);
        const long long r0 = t * TILE + (tid >> 5);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const long long r = r0 + 16 * q;
            pd[q] = pz[q] = px[q] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (r < rows) {
                const unsigned off = (unsigned)r * (unsigned)C + (unsigned)c4;
                pd[q] = *reinterpret_cast<const f32x4*>(a.dh1 + off);
                pz[q] = *reinterpret_cast<const f32x4*>(a.z1 + off);
                const unsigned hoff = mapped(f0, in0, (tid >> 5) + 16 * q) * (unsigned)C + (unsigned)c4;
                px[q] = *reinterpret_cast<const f32x4*>(a.h + hoff);
                if (SUMS) pl[q] = *reinterpret_cast<const f32x4*>(a.lz + hoff);
            }
        }
    };
    if ((long long)blockIdx.x < n_tiles) issue(blockIdx.x);
    for (long long t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        int lane = lane_k;
        asm volatile("" : "+v"(lane));
        const long long r0 = t * TILE;
        {
            const f32x4 mn = *reinterpret_cast<const f32x4*>(s_c + 0 * H1 + c4), is = *reinterpret_cast<const f32x4*>(s_c + 1 * H1 + c4);
            const f32x4 ga = *reinterpret_cast<const f32x4*>(s_c + 2 * H1 + c4), be = *reinterpret_cast<const f32x4*>(s_c + 3 * H1 + c4);
            const f32x4 mg = *reinterpret_cast<const f32x4*>(s_c + 4 * H1 + c4), mgx = *reinterpret_cast<const f32x4*>(s_c + 5 * H1 + c4);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int rl = (tid >> 5) + 16 * q;
                const long long r = r0 + rl;
                f32x4 vg = {0.f, 0.f, 0.f, 0.f};
                if (r < rows) {
                    f32x4 d = pd[q];
                    if (a.masked) {
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const float xh = (pz[q][u] - mn[u]) * is[u];
                            vg[u] = ga[u] * is[u] * (d[u] - mg[u] - xh * mgx[u]);
                        }
                    } else {
                        if (a.d1.p > 0.f) d *= keep_scale4(a.d1.seed, (unsigned long long)r * C + c4, a.d1.p, a.d1.inv_keep);
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const float xh = (pz[q][u] - mn[u]) * is[u];
                            const float v = xh * ga[u] + be[u];
                            const float g = v > 0.f ? d[u] : 0.f;
                            vg[u] = ga[u] * is[u] * (g - mg[u] - xh * mgx[u]);
                        }
                    }
                }
                *reinterpret_cast<f32x4*>(&s_g[rl * LDA + c4]) = vg;
                *reinterpret_cast<f32x4*>(&s_x[rl * LDA + c4]) = px[q];
                if (SUMS) cl[q] = pl[q];
            }
        }
        __syncthreads();                          // tile in LDS; s_o of the tile before has been stored
        if (t + gridDim.x < n_tiles) issue(t + gridDim.x);
        if constexpr (!GEMM2) {
            const int i = lane & 31, kh = lane >> 5;
            if (!(FB_ABL & 1))
#pragma unroll 4
            for (int s = 0; s < TILE / 2; ++s) {
                const int r = 2 * s + kh;
                const float av = s_g[r * LDA + 32 * wave + i];
#pragma unroll
                for (int jb = 0; jb < 4; ++jb)
                    acc[jb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, s_x[r * LDA + 32 * jb + i], acc[jb], 0, 0, 0);
            }
        } else {
            const int j = lane & 31, hh = lane >> 5, w = wave - 4;
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) {
                f32x16 o;
#pragma unroll
                for (int e = 0; e < 16; ++e) o[e] = 0.f;
                if (!(FB_ABL & 2)) mfma_rowblock(s_g, 32 * rb, lane, wreg, o);
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<f32x4*>(&s_o[(32 * rb + j) * LDA + 32 * w + 4 * hh + 8 * g]) = f32x4{o[4 * g], o[4 * g + 1], o[4 * g + 2], o[4 * g + 3]};
            }
        }
        __syncthreads();                          // products done: s_g / s_x may be refilled, s_o is complete
        {
            int f0, in0;
            tile_map(t, f0, in0);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int rl = (tid >> 5) + 16 * q;
                const long long r = r0 + rl;
                if (r < rows) {
                    const unsigned hoff = mapped(f0, in0, rl) * (unsigned)C + (unsigned)c4;
                    const f32x4 dv = *reinterpret_cast<const f32x4*>(&s_o[rl * LDA + c4]);
                    *reinterpret_cast<f32x4*>(a.dh + hoff) = dv;
                    if (SUMS) {                   // k_bn_bwd_partial's expression per element (train.hip), on the row just written
                        const f32x4 lm = *reinterpret_cast<const f32x4*>(s_c + 6 * H1 + c4), li = *reinterpret_cast<const f32x4*>(s_c + 7 * H1 + c4);
                        const f32x4 lg = *reinterpret_cast<const f32x4*>(s_c + 8 * H1 + c4), lb = *reinterpret_cast<const f32x4*>(s_c + 9 * H1 + c4);
                        f32x4 g = dv;
                        if (a.la.p > 0.f) g *= keep_scale4(a.la.seed, (unsigned long long)hoff, a.la.p, a.la.inv_keep);
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const float xh = (cl[q][u] - lm[u]) * li[u];
                            const float v = xh * lg[u] + lb[u];
                            float ge = g[u];
                            if (a.la.relu) ge = v > 0.f ? ge : 0.f;
                            lsg[u] += ge;
                            lsx[u] += ge * xh;
                        }
                    }
                }
            }
        }
    }
    if constexpr (SUMS) {                         // the 16 threads that share a channel group (both roles), in a fixed order
        __syncthreads();
        float* red = s_g;                         // [16][2][128]
        const int rg = tid >> 5;
#pragma unroll
        for (int u = 0; u < 4; ++u) { red[(rg * 2 + 0) * H1 + c4 + u] = lsg[u]; red[(rg * 2 + 1) * H1 + c4 + u] = lsx[u]; }
        __syncthreads();
        if (tid < 2 * H1) {
            float t = 0.f;
#pragma unroll
            for (int k = 0; k < 16; ++k) t += red[k * 2 * H1 + tid];
            a.partial_lsums[(size_t)blockIdx.x * 2 * H1 + tid] = t;
        }
        __syncthreads();
    }
    if constexpr (!GEMM2) {
        const int i = lane_k & 31, kh = lane_k >> 5;
        float* p = a.partial_dw1 + (size_t)blockIdx.x * C * C;
#pragma unroll
        for (int jb = 0; jb < 4; ++jb)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = (e & 3) + 8 * (e >> 2) + 4 * kh;
                p[(size_t)(32 * wave + m) * C + 32 * jb + i] = acc[jb][e];
            }
    }
}

// DIRECT form (the arrays below 2 GB each -- batch 32 at 224/7 has 1.2 GB -- else the flat-address form above).
// Every vector instruction of a tile serialises with the SIMD's MFMAs (running the two roles half a period apart, so
// that each SIMD always had one wave in its chain and one wave on the tile's other work, changed nothing:
// profiles/r05_heads_bwd_stagger_ab.txt), so what the tile costs beside its 256 MFMAs per SIMD is its instruction COUNT.  Here the
// 16 row loads and 4 row stores of a tile are buffer accesses off whole-array descriptors: the lane's part of the offset is one
// VGPR per row group for the lifetime of the workgroup, the tile's part an SGPR; rows past the end read as zero and their stores
// are dropped (no per-row bounds test, no zero-initialised prefetch registers), no 64-bit address arithmetic; the one tile per
// frame that straddles a frame boundary adds the gap to the offsets of its later rows.  The layer's z rows (SUMS) are loaded
// row group by row group in the store phase, each right behind the last use of the current tile's (no second register set).
// Same arithmetic per element, same accumulation order: the same bits as the flat form.
template <bool GEMM2, bool SUMS>
__device__ inline void first_bwd_role_direct(const FirstBwdArgs& a, float* s_g, float* s_x, float* s_o, const float* s_c, int wave) {
    const int tid = threadIdx.x, lane_k = tid & 63;
    const int c4 = (tid & 31) * 4;
    float wreg[GEMM2 ? 64 : 1];                   // waves 4..7: W1[64 kh + s][32 (wave - 4) + i] (the transposed slice)
    f32x16 acc[GEMM2 ? 1 : 4];                    // waves 0..3: dW1[32 wave + m][32 jb + n]
    if constexpr (GEMM2) load_w_slice(a.W1, wave - 4, lane_k, 1, wreg);
    else {
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[b][e] = 0.f;
    }
    const int n_tiles = (int)((a.rows + TILE - 1) / TILE);
    const int rows = (int)a.rows;                 // (< 2^22 rows: the arrays are below 2 GB)
    f32x4 pd[4], pz[4], px[4];
    f32x4 pl[SUMS ? 4 : 1];
    f32x4 lsg = {0.f, 0.f, 0.f, 0.f}, lsx = lsg;
    const __amdgpu_buffer_rsrc_t rs_d = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dh1), 0, (int)a.bytes_c, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_z = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.z1), 0, (int)a.bytes_c, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_h = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.h), 0, a.h ? (int)a.bytes_m : 0, 0x00020000);   // (hrec without a residual: every row reads as zero)
    const __amdgpu_buffer_rsrc_t rs_l = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(SUMS ? a.lz : a.h), 0, (int)a.bytes_m, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_o = __builtin_amdgcn_make_buffer_rsrc(a.dh, 0, (int)a.bytes_m, 0x00020000);
    const int rl0 = tid >> 5;                     // this thread's rows of a tile: rl0 + 16 q
    const int voff0 = rl0 * (C * 4) + c4 * 4;     // ... their byte offsets: voff0 + q * QB
    constexpr int QB = 16 * C * 4;
    const int gap = (a.xm.stride - a.xm.n_valid) * (C * 4);
    // tile t: compact rows 64 t ..; its first row is row in0 of frame f0's kept rows (a 32-bit division on the scalar unit)
    struct Tile { int soff_c, soff_m, brk; };     // byte offsets of the tile's first row (compact / unfiltered); rows >= brk lie in the next frame
    auto tile_of = [&](int t) -> Tile {
        const unsigned r0 = (unsigned)__builtin_amdgcn_readfirstlane(t * TILE);
        const unsigned f0 = r0 / (unsigned)a.xm.n_valid;
        const int in0 = (int)(r0 - f0 * (unsigned)a.xm.n_valid);
        return Tile{(int)r0 * (C * 4), ((int)f0 * a.xm.stride + a.xm.lo + in0) * (C * 4), a.xm.n_valid - in0};
    };
    // offset of row group q in the unfiltered arrays (a tile that straddles a frame boundary -- one per frame -- skips the gap)
    auto vm = [&](const Tile& T, int q) -> int {
        const int v = voff0 + q * QB;
        return T.brk >= TILE ? v : (rl0 + 16 * q >= T.brk ? v + gap : v);
    };
    auto issue = [&](const Tile& T) {
        if (FB_ABL & 32) return;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            pd[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_d, voff0 + q * QB, T.soff_c, 0));
            pz[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_z, voff0 + q * QB, T.soff_c, 0));
            px[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_h, vm(T, q), T.soff_m, 0));
        }
    };
    if ((int)blockIdx.x < n_tiles) {
        const Tile T = tile_of(blockIdx.x);
        issue(T);
        if (SUMS) {
#pragma unroll
            for (int q = 0; q < 4; ++q) pl[SUMS ? q : 0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_l, vm(T, q), T.soff_m, 0));
        }
    }
#ifdef FB_STAMP              // (diagnostic build: cycles per phase, summed over the workgroup's tiles, printed by two waves of workgroup 0)
    unsigned long long st[7] = {0, 0, 0, 0, 0, 0, 0};
#define FB_T0 unsigned long long t_prev_ = __builtin_readcyclecounter();
#define FB_MARK(k) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); const unsigned long long now_ = __builtin_readcyclecounter(); st[k] += now_ - t_prev_; t_prev_ = now_; } while (0)
#define FB_MARK_NOWAIT(k) do { const unsigned long long now_ = __builtin_readcyclecounter(); st[k] += now_ - t_prev_; t_prev_ = now_; } while (0)
#else
#define FB_T0
#define FB_MARK(k) do {} while (0)
#define FB_MARK_NOWAIT(k) do {} while (0)
#endif
    for (int t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        int lane = lane_k;
        asm volatile("" : "+v"(lane));
        FB_T0
        FB_MARK(0);                               // wait for the prefetched rows
        const bool ragged = (t + 1) * TILE > rows;            // the array's last tile: rows past the end must not reach the sums
        unsigned lkeep = 0xFFFFu;              // the layer's dropout mask over this thread's 4 x 4 elements of the tile: hashed once, used
        if (SUMS && a.la.p > 0.f) {            // for the rebuilt h here and for the gate of the sums where dh leaves the tile
            const Tile Tc = tile_of(t);
            lkeep = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                lkeep |= keep_bits4(a.la.seed, (unsigned long long)(((unsigned)Tc.soff_m + (unsigned)vm(Tc, q)) >> 2), a.la.p) << (4 * q);
        }
        if (SUMS && a.hrec) {
            // the layer's output h = act(z) + residual was never written (k_act_lin128 h_skip_*): rebuilt here from the z rows this tile
            // holds for its sums anyway and the residual rows (prefetched in h's place), with the forward's own expression and mask
            const f32x4 lsc = *reinterpret_cast<const f32x4*>(s_c + 10 * H1 + c4), lsh = *reinterpret_cast<const f32x4*>(s_c + 11 * H1 + c4);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 v = pl[SUMS ? q : 0] * lsc + lsh;
                if (a.la.p > 0.f) v *= keep_scale_of_bits(lkeep >> (4 * q), a.la.inv_keep);
                if (a.la.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                px[q] = v + px[q];
            }
        }
        {
            const f32x4 mn = *reinterpret_cast<const f32x4*>(s_c + 0 * H1 + c4), is = *reinterpret_cast<const f32x4*>(s_c + 1 * H1 + c4);
            const f32x4 ga = *reinterpret_cast<const f32x4*>(s_c + 2 * H1 + c4), be = *reinterpret_cast<const f32x4*>(s_c + 3 * H1 + c4);
            const f32x4 mg = *reinterpret_cast<const f32x4*>(s_c + 4 * H1 + c4), mgx = *reinterpret_cast<const f32x4*>(s_c + 5 * H1 + c4);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int rl = rl0 + 16 * q;
                f32x4 vg;
                f32x4 d = pd[q];
                if (FB_ABL & 4) vg = d + pz[q];
                else if (a.masked) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const float xh = (pz[q][u] - mn[u]) * is[u];
                        vg[u] = ga[u] * is[u] * (d[u] - mg[u] - xh * mgx[u]);
                    }
                } else {
                    if (a.d1.p > 0.f) d *= keep_scale4(a.d1.seed, (unsigned long long)(t * TILE + rl) * C + c4, a.d1.p, a.d1.inv_keep);
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const float xh = (pz[q][u] - mn[u]) * is[u];
                        const float v = xh * ga[u] + be[u];
                        const float g = v > 0.f ? d[u] : 0.f;
                        vg[u] = ga[u] * is[u] * (g - mg[u] - xh * mgx[u]);
                    }
                }
                if (ragged && t * TILE + rl >= rows) vg = f32x4{0.f, 0.f, 0.f, 0.f};      // (as the flat form: a missing row is a zero row)
                if (!(FB_ABL & 16)) {
                    *reinterpret_cast<f32x4*>(&s_g[rl * LDA + c4]) = vg;
                    *reinterpret_cast<f32x4*>(&s_x[rl * LDA + c4]) = px[q];
                } else asm volatile("" :: "v"(vg.x), "v"(vg.y), "v"(vg.z), "v"(vg.w), "v"(px[q].x), "v"(px[q].y), "v"(px[q].z), "v"(px[q].w));
            }
        }
        __syncthreads();                          // tile in LDS; s_o of the tile before has been stored
        const bool more = t + (int)gridDim.x < n_tiles;      // (the loads in FRONT of this barrier instead: 2.40 -> 2.42 ms, profiles/r05_heads_bwd_ablations.txt)
        Tile Tn{0, 0, TILE};
        if (more) { Tn = tile_of(t + gridDim.x); issue(Tn); }
        FB_MARK_NOWAIT(3);                        // issue
        if constexpr (!GEMM2) {
            const int i = lane & 31, kh = lane >> 5;
            if (!(FB_ABL & 1))
#pragma unroll 4
            for (int s = 0; s < TILE / 2; ++s) {
                const int r = 2 * s + kh;
                const float av = s_g[r * LDA + 32 * wave + i];
#pragma unroll
                for (int jb = 0; jb < 4; ++jb)
                    acc[jb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, s_x[r * LDA + 32 * jb + i], acc[jb], 0, 0, 0);
            }
        } else {
            const int j = lane & 31, hh = lane >> 5, w = wave - 4;
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) {
                f32x16 o;
#pragma unroll
                for (int e = 0; e < 16; ++e) o[e] = 0.f;
                if (!(FB_ABL & 2)) mfma_rowblock(s_g, 32 * rb, lane, wreg, o);
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<f32x4*>(&s_o[(32 * rb + j) * LDA + 32 * w + 4 * hh + 8 * g]) = f32x4{o[4 * g], o[4 * g + 1], o[4 * g + 2], o[4 * g + 3]};
            }
        }
        FB_MARK_NOWAIT(4);                        // products
        __syncthreads();                          // products done: s_g / s_x may be refilled, s_o is complete
        FB_MARK_NOWAIT(5);                        // barrier 2
        {
            const Tile T = tile_of(t);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int rl = rl0 + 16 * q;
                const int vo = vm(T, q);
                const f32x4 dv = *reinterpret_cast<const f32x4*>(&s_o[rl * LDA + c4]);
                if (!(FB_ABL & 8)) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4v, dv), rs_o, vo, T.soff_m, 0);
                if (SUMS && !(FB_ABL & 64) && !(ragged && t * TILE + rl >= rows)) {     // k_bn_bwd_partial's expression per element (train.hip), on the row just written
                    const f32x4 lm = *reinterpret_cast<const f32x4*>(s_c + 6 * H1 + c4), li = *reinterpret_cast<const f32x4*>(s_c + 7 * H1 + c4);
                    const f32x4 lg = *reinterpret_cast<const f32x4*>(s_c + 8 * H1 + c4), lb = *reinterpret_cast<const f32x4*>(s_c + 9 * H1 + c4);
                    f32x4 g = dv;
#ifdef EG_ABL_HASH_FIRST      // (timing-only)
                    g *= a.la.inv_keep;
#else
                    if (a.la.p > 0.f) g *= keep_scale_of_bits(lkeep >> (4 * q), a.la.inv_keep);
#endif
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const float xh = (pl[SUMS ? q : 0][u] - lm[u]) * li[u];
                        const float v = xh * lg[u] + lb[u];
                        float ge = g[u];
                        if (a.la.relu) ge = v > 0.f ? ge : 0.f;
                        lsg[u] += ge;
                        lsx[u] += ge * xh;
                    }
                }
                asm volatile("s_nop 1" :: "v"(dv.x), "v"(dv.y), "v"(dv.z), "v"(dv.w) : "memory");       // (store data: DESIGN 5.26)
                if (SUMS && more) pl[SUMS ? q : 0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_l, vm(Tn, q), Tn.soff_m, 0));
            }
        }
        FB_MARK_NOWAIT(6);                        // store phase
    }
#ifdef FB_STAMP
    if (blockIdx.x == 7 && (tid & 63) == 0 && (wave == 0 || wave == 4 || wave == 3))
        printf("wave %d tiles %d  wait-loads %llu build %llu bar1 %llu issue %llu products %llu bar2 %llu store %llu  (cycles per tile)\n", wave,
               (n_tiles - 7 + (int)gridDim.x - 1) / (int)gridDim.x, st[0] * (unsigned long long)gridDim.x / n_tiles, st[1] * (unsigned long long)gridDim.x / n_tiles,
               st[2] * (unsigned long long)gridDim.x / n_tiles, st[3] * (unsigned long long)gridDim.x / n_tiles, st[4] * (unsigned long long)gridDim.x / n_tiles,
               st[5] * (unsigned long long)gridDim.x / n_tiles, st[6] * (unsigned long long)gridDim.x / n_tiles);
#endif
    if constexpr (SUMS) {                         // the 16 threads that share a channel group (both roles), in a fixed order
        __syncthreads();
        float* red = s_g;                         // [16][2][128]
        const int rg = tid >> 5;
#pragma unroll
        for (int u = 0; u < 4; ++u) { red[(rg * 2 + 0) * H1 + c4 + u] = lsg[u]; red[(rg * 2 + 1) * H1 + c4 + u] = lsx[u]; }
        __syncthreads();
        if (tid < 2 * H1) {
            float t = 0.f;
#pragma unroll
            for (int k = 0; k < 16; ++k) t += red[k * 2 * H1 + tid];
            a.partial_lsums[(size_t)blockIdx.x * 2 * H1 + tid] = t;
        }
        __syncthreads();
    }
    if constexpr (!GEMM2) {
        const int i = lane_k & 31, kh = lane_k >> 5;
        float* p = a.partial_dw1 + (size_t)blockIdx.x * C * C;
#pragma unroll
        for (int jb = 0; jb < 4; ++jb)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = (e & 3) + 8 * (e >> 2) + 4 * kh;
                p[(size_t)(32 * wave + m) * C + 32 * jb + i] = acc[jb][e];
            }
    }
}

template <bool SUMS, bool DIRECT>
__global__ __launch_bounds__(512) void k_cls_first_bwd(const FirstBwdArgs a_, const ClsBn bn1, const float* __restrict__ beta1) {
    FirstBwdArgs a = a_;
    a.d1 = resolved(a_.d1);
    if (SUMS) a.la = resolved(a_.la);
    extern __shared__ __attribute__((aligned(16))) float fb_smem[];
    float* s_g = fb_smem;                         // [64][LDA]  dz1 tile
    float* s_x = fb_smem + TILE * LDA;            // [64][LDA]  h tile (valid rows)
    float* s_o = fb_smem + 2 * TILE * LDA;        // [64][LDA]  dh tile on its way out
    float* s_c = fb_smem + 3 * TILE * LDA;        // [6 | 12][128]   mean, invstd, gamma, beta, mean g, mean g xhat (+ SUMS: the layer's mean, invstd, gamma, beta, scale, shift)
    const int tid = threadIdx.x, wave = wave_id();
    if (tid < H1) {
        const double inv_n = 1.0 / (double)a.rows;
        s_c[0 * H1 + tid] = bn1.mean[tid];
        s_c[1 * H1 + tid] = bn1.invstd[tid];
        s_c[2 * H1 + tid] = bn1.gamma[tid];
        s_c[3 * H1 + tid] = beta1[tid];
        s_c[4 * H1 + tid] = (float)(a.tot[tid] * inv_n);
        s_c[5 * H1 + tid] = (float)(a.tot[H1 + tid] * inv_n);
        if (SUMS) {
            s_c[6 * H1 + tid] = a.lmean[tid];
            s_c[7 * H1 + tid] = a.linvstd[tid];
            s_c[8 * H1 + tid] = a.lgamma[tid];
            s_c[9 * H1 + tid] = a.lbeta[tid];
            s_c[10 * H1 + tid] = a.hrec ? a.lscale[tid] : 1.f;
            s_c[11 * H1 + tid] = a.hrec ? a.lshift[tid] : 0.f;
        }
    }
    __syncthreads();
    if constexpr (DIRECT) {
        if (wave < 4) first_bwd_role_direct<false, SUMS>(a, s_g, s_x, s_o, s_c, wave);
        else first_bwd_role_direct<true, SUMS>(a, s_g, s_x, s_o, s_c, wave);
    } else {
        if (wave < 4) first_bwd_role_flat<false, SUMS>(a, s_g, s_x, s_o, s_c, wave);
        else first_bwd_role_flat<true, SUMS>(a, s_g, s_x, s_o, s_c, wave);
    }
}

__global__ void k_zero_rows(float* __restrict__ x, int batch, int stride, int lo, int n_valid) {
    // rows [0, lo) and [lo + n_valid, stride) of every frame (the rows the node-type filter drops: their gradient is zero)
    const int per = stride - n_valid;
    const long long n4 = (long long)batch * per * (C / 4);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        const long long rr = i / (C / 4);
        const int f = (int)(rr / per), k = (int)(rr - (long long)f * per);
        const int row = k < lo ? k : n_valid + k;
        *reinterpret_cast<f32x4*>(x + ((size_t)f * stride + row) * C + (i % (C / 4)) * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
    }
}

// Behind the mid kernel, ONE launch (it was four): both of its column reductions (k_reduce_f32_partials' arithmetic: dW2 -> grads; the
// first layers' BatchNorm-backward sums -> tot_bn1 for the first-layers kernel and dgamma1 / dbeta1 -> grads), the packing of the
// third layers' totals into grads (k_cls_grads_final's) and the zeroing of dh's rows outside the heads' filter (k_zero_rows').
// Workgroups: 64 (dW2, 2048 columns) + 8 (256 columns) + 1.
struct MidFinal {
    const float *partial2, *partial_bn1;
    int gb;
    const double* tot_out;
    double* tot_bn1;
    float* grads;
    float* dh;                      // nullable: nothing to zero
    int batch, stride, lo, n_valid;
};
__global__ __launch_bounds__(RED_F32_THREADS) void k_cls_mid_final(const MidFinal a) {
    __shared__ double red[RED_F32_THREADS];
    const int t = threadIdx.x, sl = t >> 5;
    float* dw2 = a.grads + 128 * 128 + 3 * 128;
    float* tail = dw2 + 4 * 16 * 32;                        // db2[64], dgamma2[64], dbeta2[64], dw3[64], db3[4]
    constexpr int NB2 = 4 * 16 * 32 / 32, NB1 = 2 * H1 / 32;
    if (blockIdx.x == NB2 + NB1) {
        if (t < H2) {
            tail[t] = 0.f;                                      // db2: a bias in front of a train-mode BatchNorm
            tail[H2 + t] = (float)a.tot_out[H2 + t];            // dgamma2 = sum g2 xhat2
            tail[2 * H2 + t] = (float)a.tot_out[t];             // dbeta2 = sum g2
            tail[3 * H2 + t] = (float)a.tot_out[2 * H2 + t];    // dw3
        }
        if (t < 4) tail[4 * H2 + t] = (float)a.tot_out[3 * H2 + t];
        if (t < 128) a.grads[128 * 128 + t] = 0.f;              // db1
        if (a.dh) {
            // rows [0, lo) and [lo + n_valid, stride) of every frame (the rows the node-type filter drops: their gradient is zero)
            const int per = a.stride - a.n_valid;
            const long long n4 = (long long)a.batch * per * (C / 4);
            for (long long i = t; i < n4; i += RED_F32_THREADS) {
                const long long rr = i / (C / 4);
                const int f = (int)(rr / per), k = (int)(rr - (long long)f * per);
                const int row = k < a.lo ? k : a.n_valid + k;
                *reinterpret_cast<f32x4*>(a.dh + ((size_t)f * a.stride + row) * C + (i % (C / 4)) * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
        return;
    }
    const bool second = blockIdx.x >= NB2;
    const float* __restrict__ partial = second ? a.partial_bn1 : a.partial2;
    const int n = second ? 2 * H1 : 4 * 16 * 32;
    const int col = (second ? blockIdx.x - NB2 : blockIdx.x) * 32 + (t & 31);
    const double s = strided_sum(partial + col, n, sl, a.gb, RED_F32_THREADS / 32);
    red[t] = s;
    __syncthreads();
#pragma unroll
    for (int st = RED_F32_THREADS / 64; st > 0; st >>= 1) {
        if (sl < st) red[t] += red[t + 32 * st];
        __syncthreads();
    }
    if (sl != 0) return;
    if (!second) { dw2[col] = (float)red[t]; return; }
    a.tot_bn1[col] = red[t];
    if (col < H1) a.grads[128 * 128 + 256 + col] = (float)red[t];           // dbeta1 = sum g1
    else a.grads[128 * 128 + 128 + col - H1] = (float)red[t];               // dgamma1 = sum g1 xhat1
}

static int grid_for(long long work_items, int per_block, int cap) {
    long long b = (work_items + per_block - 1) / per_block;
    if (b > cap) b = cap;
    return (int)(b < 1 ? 1 : b);
}

}  // namespace eg

using namespace eg;

// internal launchers of train.hip
int eg_launch_bn_bwd(const float* dy, const float* z, long long rows, const float* mean, const float* invstd, const float* gamma,
                     const float* beta, int relu, float dropout_p, unsigned long long seed, void* workspace, float* dz,
                     float* dgamma, float* dbeta, const float* x, const eg::RowMap* xmap, float* dw, hipStream_t stream,
                     const double* presum = nullptr, const eg::RowMap* presum_rows = nullptr, int presum_frames = 0,
                     const float* presum_taps = nullptr, float* db_zero = nullptr);

extern "C" {

// workspace layout (bytes): [0, 8 MB) float partial slabs; then 4096 doubles of totals; eg_bn_act_bwd / dweight use the
// region from offset CLS_WS_SHARED on (eg_workspace_bytes() bytes)
static const size_t CLS_WS_PARTIAL = (size_t)CT_MAX_BLOCKS * 2048 * sizeof(float);
static const size_t CLS_WS_TOTALS = 4096 * sizeof(double);
static const size_t CLS_WS_SHARED = CLS_WS_PARTIAL + CLS_WS_TOTALS;

size_t eg_classifier_train_workspace_bytes(void) { return CLS_WS_SHARED + eg_workspace_bytes(); }

// h != NULL: the rows exist; otherwise act describes how they come to be (and where they are written)
static int classifier_train_fwd(const float* h, const LinAct* act, int batch, int64_t n_per_frame, int64_t row_lo, int64_t n_valid,
                                const eg_cls_train_params* P, void* workspace, float* z1, float* z2, float* bn, int sigmoid,
                                float* logits, eg_stream_t stream_) {
    if (P) { if (int rc_ = eg_epoch_required(P->p1 > P->p2 ? P->p1 : P->p2)) return rc_; }
    if (act) { if (int rc_ = eg_epoch_required(act->a.p)) return rc_; }
    if (!P || !workspace || !z1 || !z2 || !bn || !logits) return set_error(EG_ERR_ARG, "NULL argument");
    if (batch < 1 || n_per_frame < 1 || row_lo < 0 || n_valid < 1 || row_lo + n_valid > n_per_frame)
        return set_error(EG_ERR_ARG, "bad row range");
    if (n_per_frame * (int64_t)batch >= (1ll << 31)) return set_error(EG_ERR_ARG, "batch * nodes exceeds int32");
    if (!(P->p1 >= 0.f && P->p1 < 1.f && P->p2 >= 0.f && P->p2 < 1.f)) return set_error(EG_ERR_ARG, "dropout p must be in [0, 1)");
    hipStream_t stream = (hipStream_t)stream_;
    const long long rows = (long long)batch * n_valid;
    float* partial = (float*)workspace;
    double* totals = (double*)((char*)workspace + CLS_WS_PARTIAL);
    // ---- first layers + BatchNorm1d(128) statistics
    LinMapDims d{};
    d.n_valid = (int)n_valid; d.tiles_per_frame = (int)(((act ? n_per_frame : n_valid) + TILE - 1) / TILE); d.batch = batch;
    d.transpose_w = 0;
    d.in_stride = (int)n_per_frame; d.in_lo = (int)row_lo; d.out_stride = (int)n_valid; d.out_lo = 0;
    const long long n_tiles = (long long)d.tiles_per_frame * batch;
    if (n_tiles >= (1ll << 31)) return set_error(EG_ERR_ARG, "too many row tiles");
    const int g1 = (int)(n_tiles < LIN_GRID ? n_tiles : LIN_GRID);
    // (round 4's 8-wave form k_lin128_map<true, true> lost every A/B against the 4-wave k_act_lin128 and is no longer reachable)
    if (act) hipLaunchKernelGGL(k_act_lin128, dim3(g1), dim3(256), 0, stream, P->w1, P->b1, z1, partial, d, *act);
    else hipLaunchKernelGGL((k_lin128_map<true, false>), dim3(g1), dim3(512), 0, stream, h, P->w1, P->b1, z1, partial, d, LinAct{});
    BnFinalize f1{totals, rows, H1, P->gamma1, P->beta1, P->eps1, P->momentum1, P->running_mean1, P->running_var1,
                  bn + 0 * H1, bn + 1 * H1, bn + 2 * H1, bn + 3 * H1};
    hipLaunchKernelGGL(k_bn_reduce_finalize, dim3(H1 / 16), dim3(RED_F32_THREADS), 0, stream, (const float*)partial, g1, f1);
    // ---- second layers + BatchNorm1d(64) statistics
    float* bn2p = bn + 4 * H1;
    const ClsBn bn1{bn + 0 * H1, bn + 1 * H1, bn + 2 * H1, bn + 3 * H1, P->gamma1};
    const ClsDrop d1{P->p1, P->p1 > 0.f ? 1.0f / (1.0f - P->p1) : 1.0f, P->seed1, eg_epoch_ptr()};
    const int g2 = grid_for(rows, TILE, 768);
    hipLaunchKernelGGL(k_cls_mid_fwd, dim3(g2), dim3(CT_THREADS), 0, stream, z1, rows, P->w2, P->b2, bn1, d1, z2, partial);
    BnFinalize f2{totals + 256, rows, H2, P->gamma2, P->beta2, P->eps2, P->momentum2, P->running_mean2, P->running_var2,
                  bn2p + 0 * H2, bn2p + 1 * H2, bn2p + 2 * H2, bn2p + 3 * H2};
    hipLaunchKernelGGL(k_bn_reduce_finalize, dim3(H2 / 16), dim3(RED_F32_THREADS), 0, stream, (const float*)partial, g2, f2);
    // ---- third layers
    const ClsBn bn2{bn2p + 0 * H2, bn2p + 1 * H2, bn2p + 2 * H2, bn2p + 3 * H2, P->gamma2};
    const ClsDrop d2{P->p2, P->p2 > 0.f ? 1.0f / (1.0f - P->p2) : 1.0f, P->seed2, eg_epoch_ptr()};
    hipLaunchKernelGGL(k_cls_out_fwd, dim3(grid_for(rows * 4, CT_THREADS, 65536)), dim3(CT_THREADS), 0, stream, z2, rows, bn2, d2,
                       P->w3, P->b3, sigmoid, logits);
    EG_HIP_TRY(hipGetLastError());
    return EG_OK;
}

int eg_classifier_train_fwd(const float* h, int batch, int64_t n_per_frame, int64_t row_lo, int64_t n_valid,
                            const eg_cls_train_params* P, void* workspace, float* z1, float* z2, float* bn, int sigmoid,
                            float* logits, eg_stream_t stream) {
    if (!h) return set_error(EG_ERR_ARG, "NULL argument");
    return classifier_train_fwd(h, nullptr, batch, n_per_frame, row_lo, n_valid, P, workspace, z1, z2, bn, sigmoid, logits, stream);
}

int eg_classifier_train_fwd_act(const float* z, const float* layer_bn, const float* residual, int relu, float dropout_p, uint64_t seed,
                                float* h, int batch, int64_t n_per_frame, int64_t row_lo, int64_t n_valid,
                                const eg_cls_train_params* P, void* workspace, float* z1, float* z2, float* bn, int sigmoid,
                                float* logits, int h_sparse, eg_stream_t stream) {
    if (!z || !layer_bn || !h) return set_error(EG_ERR_ARG, "NULL argument");
    if (dropout_p < 0.f || dropout_p >= 1.f) return set_error(EG_ERR_ARG, "dropout_p must be in [0, 1)");
    if (h == z || h == residual) return set_error(EG_ERR_ARG, "h must not alias z or the residual");
    LinAct act{};
    act.z = z; act.scale = layer_bn + 2 * C; act.shift = layer_bn + 3 * C; act.residual = residual; act.h = h;
    act.a.rows = (long long)batch * n_per_frame; act.a.relu = relu; act.a.p = dropout_p;
    act.a.inv_keep = dropout_p > 0.f ? 1.0f / (1.0f - dropout_p) : 1.0f; act.a.seed = seed; act.a.epoch = eg_epoch_ptr();
    if (h_sparse) { act.h_skip_lo = (int)row_lo; act.h_skip_hi = (int)(row_lo + n_valid); }
    return classifier_train_fwd(nullptr, &act, batch, n_per_frame, row_lo, n_valid, P, workspace, z1, z2, bn, sigmoid, logits, stream);
}

struct LayerSumsReq {               // eg_classifier_bwd_sums: the layer in front of the heads
    const float *z, *bn, *gamma, *beta;
    int relu;
    float p;
    uint64_t seed;
    double* sums;                   // out [2][128]
    const float* residual;          // recompute_h: the layer's residual rows (NULL: none)
    int recompute_h;                // h was not written (eg_classifier_train_fwd_act h_sparse): the first-layers kernel rebuilds its tile
};

static bool first_bwd_covers(const float* dh, int batch, int64_t n_per_frame, int64_t n_valid) {
    const long long rows = (long long)batch * n_valid;
    return dh && n_valid >= TILE && (long long)batch * n_per_frame * C < (1ll << 32) && rows * C < (1ll << 32);
}

static int classifier_bwd(const float* dlogits, const float* h, int batch, int64_t n_per_frame, int64_t row_lo, int64_t n_valid,
                          const eg_cls_train_params* P, const float* z1, const float* z2, const float* bn, void* workspace,
                          float* dh1_scratch, float* dh, float* grads, const LayerSumsReq* ls, eg_stream_t stream_) {
    if (P) { if (int rc_ = eg_epoch_required(P->p1 > P->p2 ? P->p1 : P->p2)) return rc_; }
    if (ls) { if (int rc_ = eg_epoch_required(ls->p)) return rc_; }
    const bool hrec = ls && ls->recompute_h;
    if (!dlogits || (!h && !hrec) || !P || !z1 || !z2 || !bn || !workspace || !dh1_scratch || !grads)
        return set_error(EG_ERR_ARG, "NULL argument");
    if (batch < 1 || n_per_frame < 1 || row_lo < 0 || n_valid < 1 || row_lo + n_valid > n_per_frame)
        return set_error(EG_ERR_ARG, "bad row range");
    if (ls && !first_bwd_covers(dh, batch, n_per_frame, n_valid))
        return set_error(EG_ERR_UNSUPPORTED, "the layer's sums come out of the fused first-layers kernel: dh wanted, n_valid >= 64, < 2^32 elements");
    // (decided before anything is launched: the recomputed h tile exists in the buffer-descriptor form of the kernel only)
    if (hrec && !((long long)batch * n_valid * (C * 4) < (1ll << 31) - (1ll << 20) && (long long)batch * n_per_frame * (C * 4) < (1ll << 31) - (1ll << 20)))
        return set_error(EG_ERR_UNSUPPORTED, "recompute_h needs arrays below 2 GB");
    hipStream_t stream = (hipStream_t)stream_;
    const long long rows = (long long)batch * n_valid;
    float* partial = (float*)workspace;
    double* totals = (double*)((char*)workspace + CLS_WS_PARTIAL);
    void* shared = (char*)workspace + CLS_WS_SHARED;
    const float* bn2p = bn + 4 * H1;
    const ClsBn bn1{bn + 0 * H1, bn + 1 * H1, bn + 2 * H1, bn + 3 * H1, P->gamma1};
    const ClsBn bn2{bn2p + 0 * H2, bn2p + 1 * H2, bn2p + 2 * H2, bn2p + 3 * H2, P->gamma2};
    const ClsDrop d1{P->p1, P->p1 > 0.f ? 1.0f / (1.0f - P->p1) : 1.0f, P->seed1, eg_epoch_ptr()};
    const ClsDrop d2{P->p2, P->p2 > 0.f ? 1.0f / (1.0f - P->p2) : 1.0f, P->seed2, eg_epoch_ptr()};
    // ---- third layers: sums
    const int ga = grid_for(rows * 4, CT_THREADS * 8, CT_MAX_BLOCKS);
    hipLaunchKernelGGL(k_cls_out_bwd_sums, dim3(ga), dim3(CT_THREADS), 0, stream, dlogits, z2, rows, bn2, d2, P->w3, partial);
    hipLaunchKernelGGL(k_reduce_f32_partials, dim3((OUT_SUMS + 31) / 32), dim3(RED_F32_THREADS), 0, stream, partial, ga, OUT_SUMS, totals);
    // ---- second layers: dz2, dW2, dh1 -- and the sums of the first layers' BatchNorm backward over g1 = dh1 * mask1
    const int gb = grid_for(rows, TILE, 256 * MID_BWD_WGS);
    float* partial2 = partial + (size_t)CT_MAX_BLOCKS * OUT_SUMS;
    float* partial_bn1 = partial2 + (size_t)768 * (4 * 16 * 32);               // [gb][2][128]
    double* tot_bn1 = totals + 256 + 4 * 16 * 32;
    const RowMap xm{(int)n_valid, (int)n_per_frame, (int)row_lo};
    const bool fused = first_bwd_covers(dh, batch, n_per_frame, n_valid);
    const bool masked = fused;                        // the masked gradient is handed over (the unfused route below applies the mask itself)
    hipLaunchKernelGGL(masked ? k_cls_mid_bwd<true> : k_cls_mid_bwd<false>, dim3(gb), dim3(CT_THREADS), 0, stream, dlogits, z2, z1, rows,
                       bn1, d1, bn2, d2, P->w2, P->w3, totals, dh1_scratch, partial2, partial_bn1);
    {
        const MidFinal mf{partial2, partial_bn1, gb, totals, tot_bn1, grads, (fused && n_valid < n_per_frame) ? dh : nullptr, batch, (int)n_per_frame,
                          (int)row_lo, (int)n_valid};
        hipLaunchKernelGGL(k_cls_mid_final, dim3(4 * 16 * 32 / 32 + 2 * H1 / 32 + 1), dim3(RED_F32_THREADS), 0, stream, mf);
    }
    EG_HIP_TRY(hipGetLastError());
    // ---- first layers: dz1 formed on the fly, dW1 = dz1^T h[valid rows] and dh[valid rows] = dz1 W1 in ONE kernel
    if (fused) {
        {
            static std::atomic<bool> attr_set[64];
            int dev = 0;
            EG_HIP_TRY(hipGetDevice(&dev));
            if (dev < 0 || dev >= 64 || !attr_set[dev].load(std::memory_order_acquire)) {
                EG_HIP_TRY(hipFuncSetAttribute((const void*)k_cls_first_bwd<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                EG_HIP_TRY(hipFuncSetAttribute((const void*)k_cls_first_bwd<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                EG_HIP_TRY(hipFuncSetAttribute((const void*)k_cls_first_bwd<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                EG_HIP_TRY(hipFuncSetAttribute((const void*)k_cls_first_bwd<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                if (dev >= 0 && dev < 64) attr_set[dev].store(true, std::memory_order_release);
            }
        }
        const long long nt = (rows + TILE - 1) / TILE;
        const int nf = (int)(nt < FB_BLOCKS ? nt : FB_BLOCKS);
        float* slabs = (float*)((char*)shared + eg_workspace_bytes() - (size_t)FB_BLOCKS * C * C * sizeof(float));
        const size_t lds = (size_t)(3 * TILE * LDA + 12 * H1) * sizeof(float);
        FirstBwdArgs fa{dh1_scratch, z1, h, P->w1, (const double*)tot_bn1, rows, xm, d1, dh, slabs, masked ? 1 : 0};
        {
            const long long bc = rows * (C * 4), bm = (long long)batch * n_per_frame * (C * 4);
            fa.direct = bc < (1ll << 31) - (1ll << 20) && bm < (1ll << 31) - (1ll << 20);   // (32-bit offsets incl. a tile's overhang)
            fa.bytes_c = fa.direct ? (unsigned)bc : 0u;
            fa.bytes_m = fa.direct ? (unsigned)bm : 0u;
        }
        if (ls && ls->recompute_h) {
            if (!fa.direct) return set_error(EG_ERR_UNSUPPORTED, "recompute_h needs arrays below 2 GB (the buffer-descriptor form of the kernel)");
            fa.h = ls->residual; fa.hrec = 1; fa.lscale = ls->bn + 2 * C; fa.lshift = ls->bn + 3 * C;
        }
        if (ls) {
            fa.lz = ls->z; fa.lmean = ls->bn; fa.linvstd = ls->bn + C; fa.lgamma = ls->gamma; fa.lbeta = ls->beta;
            fa.la.rows = (long long)batch * n_per_frame; fa.la.relu = ls->relu; fa.la.p = ls->p;
            fa.la.inv_keep = ls->p > 0.f ? 1.0f / (1.0f - ls->p) : 1.0f; fa.la.seed = ls->seed; fa.la.epoch = eg_epoch_ptr();
            fa.partial_lsums = partial_bn1;                                       // (reduced into tot_bn1 already)
            hipLaunchKernelGGL((fa.direct ? k_cls_first_bwd<true, true> : k_cls_first_bwd<true, false>), dim3(nf), dim3(512), lds, stream, fa, bn1, P->beta1);
            // dW1 from the slabs and the layer's sums from the column partials: one launch
            hipLaunchKernelGGL(k_dweight_final, dim3(C * C / 32 + 2 * H1 / 32), dim3(RED_F32_THREADS), 0, stream, (const float*)slabs, nf, grads,
                               ExtraReduce{(const float*)partial_bn1, nf, 2 * H1, ls->sums});
        } else {
            hipLaunchKernelGGL((fa.direct ? k_cls_first_bwd<false, true> : k_cls_first_bwd<false, false>), dim3(nf), dim3(512), lds, stream, fa, bn1, P->beta1);
            hipLaunchKernelGGL(k_dweight_final, dim3(C * C / 32), dim3(RED_F32_THREADS), 0, stream, (const float*)slabs, nf, grads, ExtraReduce{});
        }
        EG_HIP_TRY(hipGetLastError());
        return EG_OK;
    }
    // dh not wanted: dW1 alone out of
    // the apply + dW pass, which then writes no dz1
    float* dgamma1 = grads + 128 * 128 + 128;
    float* dbeta1 = dgamma1 + 128;
    // dh wanted but the fused kernel does not cover the shape (fewer than 64 valid rows per frame, or 2^32 elements and more):
    // dz1 goes through dh's own rows -- dh [batch * n_per_frame, 128] has room for the [rows, 128] array -- and the plain-rows
    // product then runs from a copy of it in dh1_scratch (dh1 is dead once dz1 exists)
    int rc = eg_launch_bn_bwd(dh1_scratch, z1, rows, bn + 0 * H1, bn + 1 * H1, P->gamma1, P->beta1, 1, P->p1, P->seed1, shared,
                              dh, dgamma1, dbeta1, h, &xm, grads, stream);
    if (rc != EG_OK || !dh) return rc;
    EG_HIP_TRY(hipMemcpyAsync(dh1_scratch, dh, sizeof(float) * (size_t)rows * C, hipMemcpyDeviceToDevice, stream));
    hipLaunchKernelGGL(k_zero_rows, dim3(64), dim3(256), 0, stream, dh, batch, (int)n_per_frame, (int)row_lo, (int)n_valid);
    LinMapDims d{};
    d.n_valid = (int)n_valid; d.tiles_per_frame = (int)((n_valid + TILE - 1) / TILE); d.batch = batch; d.transpose_w = 1;
    d.in_stride = (int)n_valid; d.in_lo = 0; d.out_stride = (int)n_per_frame; d.out_lo = (int)row_lo;
    const long long n_tiles = (long long)d.tiles_per_frame * batch;
    hipLaunchKernelGGL((k_lin128_map<false, false>), dim3((unsigned)(n_tiles < LIN_GRID ? n_tiles : LIN_GRID)), dim3(512), 0, stream, dh1_scratch,
                       P->w1, (const float*)nullptr, dh, (float*)nullptr, d, LinAct{});
    EG_HIP_TRY(hipGetLastError());
    return EG_OK;
}

int eg_classifier_bwd(const float* dlogits, const float* h, int batch, int64_t n_per_frame, int64_t row_lo, int64_t n_valid,
                      const eg_cls_train_params* P, const float* z1, const float* z2, const float* bn, void* workspace,
                      float* dh1_scratch, float* dh, float* grads, eg_stream_t stream) {
    return classifier_bwd(dlogits, h, batch, n_per_frame, row_lo, n_valid, P, z1, z2, bn, workspace, dh1_scratch, dh, grads, nullptr, stream);
}

int eg_classifier_bwd_sums(const float* dlogits, const float* h, int batch, int64_t n_per_frame, int64_t row_lo, int64_t n_valid,
                           const eg_cls_train_params* P, const float* z1, const float* z2, const float* bn, void* workspace,
                           float* dh1_scratch, float* dh, float* grads, const float* layer_z, const float* layer_bn,
                           const float* layer_gamma, const float* layer_beta, int layer_relu, float layer_dropout_p,
                           uint64_t layer_seed, double* layer_sums, const float* layer_residual, int recompute_h, eg_stream_t stream) {
    if (!layer_z || !layer_bn || !layer_gamma || !layer_beta || !layer_sums) return set_error(EG_ERR_ARG, "NULL argument");
    if (layer_dropout_p < 0.f || layer_dropout_p >= 1.f) return set_error(EG_ERR_ARG, "dropout_p must be in [0, 1)");
    if (!recompute_h && !h) return set_error(EG_ERR_ARG, "h is NULL and recompute_h is not set");
    const LayerSumsReq ls{layer_z, layer_bn, layer_gamma, layer_beta, layer_relu, layer_dropout_p, layer_seed, layer_sums, layer_residual,
                          recompute_h ? 1 : 0};
    return classifier_bwd(dlogits, h, batch, n_per_frame, row_lo, n_valid, P, z1, z2, bn, workspace, dh1_scratch, dh, grads, &ls, stream);
}

}  // extern "C"
