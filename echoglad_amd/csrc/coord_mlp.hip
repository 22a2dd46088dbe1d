// Coordinate-graph landmark update for gfx950 (reference src/core/models.py:438-453):
//     shape_feats[(f,j)] = coords[f,:,:] - coords[f,j,:]                                   (8 numbers per landmark)
//     delta = node_coordinate_mlp[i](cat(lm, shape_feats))        Linear(136,32)-BN-ReLU-Drop-Linear(32,16)-BN-ReLU-Drop-Linear(16,2)
//     coords <- clamp(coords + delta, 0, frame - 1)
// on the 4 landmark rows of every frame: R = 4 * batch rows in all (128 at batch 32).  That is far too little work for the
// chip -- what matters is launch count: the reference's module list is ~15 launches forward and ~35 backward per GNN layer
// (three of them rocBLAS GEMMs with 2..32 output columns).  Here the forward is ONE single-workgroup kernel and the backward
// another; BatchNorm uses batch statistics over the R rows (train) or the running ones (eval), Dropout is the same
// counter-based mask as everywhere else (train_common.h), hidden activations are recomputed from the saved pre-BatchNorm
// z1 / z2.  All reductions run in a fixed order inside the workgroup: bitwise reproducible.
#include "coord_common.h"

namespace eg {

#ifndef EG_CM_ABL            // timing-only ablations of k_coord_update_fwd_small (tools/coord_kernel_time.py): bit 0 no z1 product, 1 no
#define EG_CM_ABL 0          // statistics, 2 no weight load, 3 no input staging, 4 no hidden activations / z2
#endif
constexpr int CM_IN = 136, CM_H1 = 32, CM_H2 = 16, CM_OUT = 2;
constexpr int CM_THREADS = 1024, CM_WAVES = CM_THREADS / 64;
constexpr int CM_TILE = 64;                     // rows staged in LDS at a time
constexpr int CM_LDI = 140;                     // LDS row stride of an input tile (16-byte aligned rows)
constexpr int CM_LDW1 = CM_IN + 1, CM_LDW2 = CM_H1 + 1, CM_LDH = CM_H1 + 1, CM_LDZ2 = CM_H2 + 1;
constexpr int CM_SCR = 56;                      // backward scratch floats per row: dz2[16] | dz1[32] | d shape_feats[8]
// gradient vector layout (floats)
constexpr int CG_DW1 = 0, CG_DB1 = CG_DW1 + CM_H1 * CM_IN, CG_DG1 = CG_DB1 + CM_H1, CG_DBE1 = CG_DG1 + CM_H1;
constexpr int CG_DW2 = CG_DBE1 + CM_H1, CG_DB2 = CG_DW2 + CM_H2 * CM_H1, CG_DG2 = CG_DB2 + CM_H2, CG_DBE2 = CG_DG2 + CM_H2;
constexpr int CG_DW3 = CG_DBE2 + CM_H2, CG_DB3 = CG_DW3 + CM_OUT * CM_H2, CG_TOTAL = CG_DB3 + CM_OUT;
static_assert(CG_TOTAL == EG_COORD_MLP_GRADS_FLOATS, "header and kernel disagree on the gradient layout");

struct CoordMlpW {
    const float *w1, *b1, *gamma1, *beta1, *w2, *b2, *gamma2, *beta2, *w3, *b3;
    float p1, ik1, p2, ik2;
    unsigned long long seed1, seed2;
    const unsigned long long* epoch;                // (train_common.h: seed + epoch is what the kernels hash with)
};

struct CoordMlpFwd {
    const float *lm, *coords;                   // rows of 128 (frame f, landmark j at lm + f * lm_stride + j * 128), [R,2]
    long long lm_stride;                        // floats between two frames' landmark rows (4 * 128: a packed [R,128] array)
    float* lm_copy;                             // nullable: packed [R,128] copy of the landmark rows (kept for the backward when the
                                                // rows themselves -- coordinate rows inside [B*N,128] -- are overwritten afterwards)
    int rows, train;
    CoordMlpW w;
    float *rm1, *rv1, *rm2, *rv2;
    float eps1, eps2, mom1, mom2, cmax;
    float *z1, *z2, *bn, *pre, *newc;           // [R,32], [R,16], [96], [R,2], [R,2]
    float* newc2;                               // nullable: a second copy of the new coordinates (one for the caller to hand out, one to keep)
};

struct CoordMlpBwd {
    const float *dnew, *dnew2, *lm, *coords, *pre, *z1, *z2, *bn;     // d new_coords = dnew + dnew2 (either may be NULL)
    long long lm_stride, dlm_stride;            // floats between two frames' rows of lm / dlm (4 * 128: packed)
    int dlm_acc;                                // dlm += instead of dlm =
    int rows;
    CoordMlpW w;
    float cmax;
    float *scratch, *dlm, *dcoords, *grads;     // [R,56], [R,128] | NULL, [R,2] | NULL, [CG_TOTAL]
};

// input column i of row r = (frame f, landmark j): the landmark's features, then the offsets to the frame's 4 landmarks
__device__ inline float mlp_in(const float* __restrict__ lm, long long lm_stride, const float* __restrict__ coords, int r, int i) {
    if (i < C) return lm[(size_t)(r >> 2) * lm_stride + (r & 3) * C + i];
    const int k = (i - C) >> 1, d = i & 1;
    return coords[((r & ~3) + k) * 2 + d] - coords[r * 2 + d];
}

// rows row0 .. row0 + 63 of cat(lm, shape_feats) -> s_in[64][CM_LDI] (rows past the end: zeros).  Every global load is issued before
// the first LDS store: a loop that loads, stores and goes round again is one dependent memory round trip per iteration (8.5 of them
// here: 6 us of a 16-us kernel).
__device__ inline void stage_inputs(const float* __restrict__ lm, long long lm_stride, const float* __restrict__ coords, int row0, int R,
                                    float* s_in, float* __restrict__ lm_copy = nullptr) {
    const int t = threadIdx.x;
    if ((((uintptr_t)lm | (uintptr_t)lm_copy) & 15) == 0 && (lm_stride & 3) == 0) {
        // landmark features: 64 rows x 32 float4, two per thread (rows r and r + 32)
        const int q = t & 31;
        f32x4 v[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int row = row0 + (t >> 5) + 32 * k;
            v[k] = row < R ? *reinterpret_cast<const f32x4*>(lm + (size_t)(row >> 2) * lm_stride + (row & 3) * C + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        // offsets to the frame's 4 landmarks: 64 rows x 8, thread t < 512 -> (row, k, d)
        float sf = 0.f;
        const int sr = t >> 3, sk = (t & 7) >> 1, sd = t & 1, srow = row0 + sr;
        if (t < CM_TILE * 8 && srow < R) sf = coords[((srow & ~3) + sk) * 2 + sd] - coords[srow * 2 + sd];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int r = (t >> 5) + 32 * k, row = row0 + r;
            *reinterpret_cast<f32x4*>(&s_in[r * CM_LDI + 4 * q]) = v[k];
            if (lm_copy && row < R) *reinterpret_cast<f32x4*>(lm_copy + (size_t)row * C + 4 * q) = v[k];
        }
        if (t < CM_TILE * 8) s_in[sr * CM_LDI + C + (t & 7)] = sf;
        return;
    }
    for (int e = t; e < CM_TILE * CM_IN; e += CM_THREADS) {
        const int r = e / CM_IN, i = e - r * CM_IN, row = row0 + r;
        const float v = row < R ? mlp_in(lm, lm_stride, coords, row, i) : 0.f;
        s_in[r * CM_LDI + i] = v;
        if (lm_copy && i < C && row < R) lm_copy[(size_t)row * C + i] = v;
    }
}

__device__ inline float hidden_act(float z, float scale, float shift, unsigned long long seed, unsigned long long idx, float p,
                                   float ik, float* keep_out) {
    const float y = __builtin_fmaf(z, scale, shift);       // (explicit: the small and the general kernels must agree bit for bit -- frame f of a
                                                            //  batch of 32 runs the one, the same frame alone the other -- whatever the compiler would contract)
    const float k = p > 0.f ? keep_scale(seed, idx, p, ik) : 1.0f;
    *keep_out = y > 0.f ? k : 0.f;              // d act / d y
    return y > 0.f ? y * k : 0.f;
}

__device__ inline void load_weights(const CoordMlpW& w, float* s_w1, float* s_w2, float* s_w3) {
    const int t = threadIdx.x;
    if ((((uintptr_t)w.w1 | (uintptr_t)w.w2) & 15) == 0) {
        // all loads first (w1: 1088 float4 -- one per thread and a second one for 64 of them; w2: 128; w3: 32 floats), then the stores
        static_assert(CM_H1 * CM_IN / 4 == CM_THREADS + 64 && CM_IN % 4 == 0 && CM_H2 * CM_H1 / 4 == 128, "thread -> float4 assignment");
        const f32x4 a0 = reinterpret_cast<const f32x4*>(w.w1)[t];
        const f32x4 a1 = t < 64 ? reinterpret_cast<const f32x4*>(w.w1)[CM_THREADS + t] : f32x4{0.f, 0.f, 0.f, 0.f};
        const f32x4 b0 = t < 128 ? reinterpret_cast<const f32x4*>(w.w2)[t] : f32x4{0.f, 0.f, 0.f, 0.f};
        const float c0 = t < CM_OUT * CM_H2 ? w.w3[t] : 0.f;
        {
            const int i = 4 * t, o = i / CM_IN, k = i - o * CM_IN;
            float* d = s_w1 + o * CM_LDW1 + k;
            d[0] = a0.x; d[1] = a0.y; d[2] = a0.z; d[3] = a0.w;
        }
        if (t < 64) {
            const int i = 4 * (CM_THREADS + t), o = i / CM_IN, k = i - o * CM_IN;
            float* d = s_w1 + o * CM_LDW1 + k;
            d[0] = a1.x; d[1] = a1.y; d[2] = a1.z; d[3] = a1.w;
        }
        if (t < 128) {
            const int i = 4 * t, o = i / CM_H1, k = i - o * CM_H1;
            float* d = s_w2 + o * CM_LDW2 + k;
            d[0] = b0.x; d[1] = b0.y; d[2] = b0.z; d[3] = b0.w;
        }
        if (t < CM_OUT * CM_H2) s_w3[t] = c0;
        return;
    }
    for (int i = t; i < CM_H1 * CM_IN; i += CM_THREADS) s_w1[(i / CM_IN) * CM_LDW1 + i % CM_IN] = w.w1[i];
    for (int i = t; i < CM_H2 * CM_H1; i += CM_THREADS) s_w2[(i / CM_H1) * CM_LDW2 + i % CM_H1] = w.w2[i];
    if (t < CM_OUT * CM_H2) s_w3[t] = w.w3[t];
}

// s_red[PARTS][W] -> sum for column c, fixed order (threads < W call this after a barrier)
template <int W>
__device__ inline float reduce_parts(const float* s_red, int c) {
    float t = 0.f;
#pragma unroll 8
    for (int p = 0; p < CM_THREADS / W; ++p) t += s_red[p * W + c];
    return t;
}

// sum of v over the whole workgroup (fixed order: lanes by xor-shuffle, then waves 0..15), returned to every thread
__device__ inline float block_sum(float v, float* s_part) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    __syncthreads();                            // s_part may still be read from a previous call
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < CM_WAVES; ++w) t += s_part[w];
    return t;
}

// mean and biased variance of every column of z [rows][W] (two passes, fixed order) -> s_mean, s_var
template <int W>
__device__ inline void column_stats(const float* __restrict__ z, int rows, float* s_red, float* s_mean, float* s_var) {
    constexpr int PARTS = CM_THREADS / W;
    const int c = threadIdx.x % W, part = threadIdx.x / W;
    float s = 0.f;
    for (int r = part; r < rows; r += PARTS) s += z[r * W + c];
    s_red[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x < W) s_mean[c] = reduce_parts<W>(s_red, c) / (float)rows;
    __syncthreads();
    const float m = s_mean[c];
    float q = 0.f;
    for (int r = part; r < rows; r += PARTS) {
        const float d = z[r * W + c] - m;
        q += d * d;
    }
    s_red[threadIdx.x] = q;
    __syncthreads();
    if (threadIdx.x < W) s_var[c] = reduce_parts<W>(s_red, c) / (float)rows;
    __syncthreads();
}

__device__ inline CoordMlpW resolved(const CoordMlpW& w) {
    CoordMlpW r = w;
    const unsigned long long e = epoch_now(w.epoch);
    r.seed1 = w.seed1 + e;
    r.seed2 = w.seed2 + e;
    return r;
}

// small parameters in LDS: b1 g1 be1 m1 v1 (32 each) | b2 g2 be2 m2 v2 (16 each) | b3 (2)
constexpr int SP_B1 = 0, SP_G1 = 32, SP_BE1 = 64, SP_M1 = 96, SP_V1 = 128, SP_B2 = 160, SP_G2 = 176, SP_BE2 = 192, SP_M2 = 208, SP_V2 = 224,
              SP_B3 = 240, SP_TOTAL = 242;

__device__ inline void load_small_params(const CoordMlpW& w, const float* rm1, const float* rv1, const float* rm2, const float* rv2, float* s_par) {
    const int t = threadIdx.x;
    if (t < 32) { s_par[SP_B1 + t] = w.b1[t]; s_par[SP_G1 + t] = w.gamma1[t]; s_par[SP_BE1 + t] = w.beta1[t]; }
    else if (t < 64) { const int c = t - 32; s_par[SP_M1 + c] = rm1 ? rm1[c] : 0.f; s_par[SP_V1 + c] = rv1 ? rv1[c] : 1.f; }
    else if (t < 80) { const int c = t - 64; s_par[SP_B2 + c] = w.b2[c]; s_par[SP_G2 + c] = w.gamma2[c]; s_par[SP_BE2 + c] = w.beta2[c]; }
    else if (t < 96) { const int c = t - 80; s_par[SP_M2 + c] = rm2 ? rm2[c] : 0.f; s_par[SP_V2 + c] = rv2 ? rv2[c] : 1.f; }
    else if (t < 98) s_par[SP_B3 + t - 96] = w.b3[t - 96];
}

// bn_setup with z in LDS and the parameters in s_par
template <int W>
__device__ inline void bn_setup_lds(const float* s_z, int rows, int train, const float* s_gamma, const float* s_beta, const float* s_rm,
                                    const float* s_rv, float eps, float mom, float* rm, float* rv, float* bn_mean, float* bn_inv,
                                    float* s_red, float* s_mean, float* s_var, float* s_scale, float* s_shift) {
    if (train && !(EG_CM_ABL & 2)) column_stats<W>(s_z, rows, s_red, s_mean, s_var);
    if (threadIdx.x < W) {
        const int c = threadIdx.x;
        float mean, var;
        if (train) {
            mean = s_mean[c];
            var = s_var[c];
            if (rm && mom >= 0.f) {
                const float unbiased = rows > 1 ? var * (float)rows / (float)(rows - 1) : var;
                rm[c] = (1.0f - mom) * s_rm[c] + mom * mean;
                rv[c] = (1.0f - mom) * s_rv[c] + mom * unbiased;
            }
        } else {
            mean = s_rm[c];
            var = s_rv[c];
        }
        const float inv = 1.0f / sqrtf(var + eps);
        bn_mean[c] = mean;
        bn_inv[c] = inv;
        const float sc = s_gamma[c] * inv;
        s_scale[c] = sc;
        s_shift[c] = __builtin_fmaf(-mean, sc, s_beta[c]);
    }
    __syncthreads();
}

// Any number of rows, 64 at a time.  Up to CM_KEEP rows (batch 32 -- BASELINE configs[3]'s batch per GPU) z1 / z2 also stay in LDS for the
// phases behind them (the statistics, the next product): reading them back from global memory between the phases was four dependent
// round trips; beyond, they come back from global memory.  The small parameters sit in LDS from the top in either case.
constexpr int CM_KEEP = 128;
__global__ __launch_bounds__(CM_THREADS) void k_coord_mlp_fwd(const CoordMlpFwd a_) {
    CoordMlpFwd a = a_;
    a.w = resolved(a_.w);
    __shared__ __attribute__((aligned(16))) float s_in[CM_TILE * CM_LDI];       // input tile, later the h1 tile
    __shared__ float s_w1[CM_H1 * CM_LDW1], s_w2[CM_H2 * CM_LDW2], s_w3[CM_OUT * CM_H2], s_par[SP_TOTAL];
    __shared__ float s_z1k[CM_KEEP * CM_H1], s_z2k[CM_KEEP * CM_H2];
    __shared__ float s_red[CM_THREADS], s_mean[CM_H1], s_var[CM_H1], s_sc1[CM_H1], s_sh1[CM_H1], s_sc2[CM_H2], s_sh2[CM_H2];
    const int t = threadIdx.x, R = a.rows;
    const bool keep = R <= CM_KEEP;
    const float* z1src = keep ? s_z1k : a.z1;          // (generic pointers: LDS or global)
    const float* z2src = keep ? s_z2k : a.z2;
    const float* __restrict__ lm = a.lm;
    const float* __restrict__ coords = a.coords;
    load_weights(a.w, s_w1, s_w2, s_w3);
    load_small_params(a.w, a.rm1, a.rv1, a.rm2, a.rv2, s_par);
    // ---- z1 = in W1^T + b1, 64 rows at a time: thread -> (row t >> 5 and +32, output channel t & 31)
    for (int row0 = 0; row0 < R; row0 += CM_TILE) {
        __syncthreads();
        stage_inputs(lm, a.lm_stride, coords, row0, R, s_in, a.lm_copy);
        __syncthreads();
        const int o = t & 31;
        const float b = s_par[SP_B1 + o];
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int r = (t >> 5) + 32 * half;
            if (row0 + r >= R) continue;
            float acc = b;
#pragma unroll 8
            for (int i = 0; i < CM_IN; ++i) acc = __builtin_fmaf(s_w1[o * CM_LDW1 + i], s_in[r * CM_LDI + i], acc);
            a.z1[(row0 + r) * CM_H1 + o] = acc;
            if (keep) s_z1k[(row0 + r) * CM_H1 + o] = acc;
        }
    }
    __syncthreads();
    bn_setup_lds<CM_H1>(z1src, R, a.train, s_par + SP_G1, s_par + SP_BE1, s_par + SP_M1, s_par + SP_V1, a.eps1, a.mom1, a.rm1, a.rv1, a.bn,
                        a.bn + CM_H1, s_red, s_mean, s_var, s_sc1, s_sh1);
    // ---- z2 = h1 W2^T + b2: h1 tile [64][33] in LDS, thread -> (row t >> 4, output channel t & 15)
    float* s_h1 = s_in;
    for (int row0 = 0; row0 < R; row0 += CM_TILE) {
        __syncthreads();
        for (int e = t; e < CM_TILE * CM_H1; e += CM_THREADS) {
            const int r = e >> 5, i = e & 31, row = row0 + r;
            float k, h = 0.f;
            if (row < R)
                h = hidden_act(z1src[row * CM_H1 + i], s_sc1[i], s_sh1[i], a.w.seed1, (unsigned long long)row * CM_H1 + i, a.w.p1,
                               a.w.ik1, &k);
            s_h1[r * CM_LDH + i] = h;
        }
        __syncthreads();
        const int r = t >> 4, o = t & 15;
        if (row0 + r < R) {
            float acc = s_par[SP_B2 + o];
#pragma unroll
            for (int i = 0; i < CM_H1; ++i) acc = __builtin_fmaf(s_w2[o * CM_LDW2 + i], s_h1[r * CM_LDH + i], acc);
            a.z2[(row0 + r) * CM_H2 + o] = acc;
            if (keep) s_z2k[(row0 + r) * CM_H2 + o] = acc;
        }
    }
    __syncthreads();
    bn_setup_lds<CM_H2>(z2src, R, a.train, s_par + SP_G2, s_par + SP_BE2, s_par + SP_M2, s_par + SP_V2, a.eps2, a.mom2, a.rm2, a.rv2,
                        a.bn + 2 * CM_H1, a.bn + 2 * CM_H1 + CM_H2, s_red, s_mean, s_var, s_sc2, s_sh2);
    // ---- delta = h2 W3^T + b3; coords <- clamp(coords + delta)
    for (int e = t; e < R * CM_OUT; e += CM_THREADS) {
        const int r = e >> 1, d = e & 1;
        float acc = s_par[SP_B3 + d];
#pragma unroll
        for (int i = 0; i < CM_H2; ++i) {
            float k;
            const float h = hidden_act(z2src[r * CM_H2 + i], s_sc2[i], s_sh2[i], a.w.seed2, (unsigned long long)r * CM_H2 + i,
                                       a.w.p2, a.w.ik2, &k);
            acc = __builtin_fmaf(s_w3[d * CM_H2 + i], h, acc);
        }
        const float pre = coords[e] + acc;
        if (a.pre) a.pre[e] = pre;
        const float nc = fminf(fmaxf(pre, 0.f), a.cmax);
        a.newc[e] = nc;
        if (a.newc2) a.newc2[e] = nc;
    }
}

// ---- up to 64 rows (batch <= 16): the whole update in LDS -----------------------------------------------------------------
// The general kernel above round-trips z1 / z2 through global memory between its phases and fetches its small parameters where it
// needs them: ~15 dependent global round trips, 20 us for a kernel with 70 KB of work -- and at batch 1, where a whole training step
// is ~1 ms, there are three of them forward and three backward.  With R <= 64 everything fits one tile: every global load is issued
// at the top, z1 / z2 stay in LDS (and go out to global once, for the backward), and the resampling of the coordinate rows
// (k_bilinear4_fwd, a launch of its own otherwise) runs at the end on the coordinates still in LDS.  Same arithmetic, same order of
// additions as the general kernel.
struct CoordSample {            // resampling behind the MLP (h == NULL: none)
    const float* h;             // [batch * n_per_frame, 128]
    float* out;                 // row 0 of frame 0's sample rows; frame stride row_stride floats
    long long n_per_frame, main_base, row_stride;
    int frame;
};

__global__ __launch_bounds__(CM_THREADS) void k_coord_update_fwd_small(const CoordMlpFwd a_, const CoordSample sp) {
    CoordMlpFwd a = a_;
    __shared__ __attribute__((aligned(16))) float s_in[CM_TILE * CM_LDI];
    __shared__ float s_w1[CM_H1 * CM_LDW1], s_w2[CM_H2 * CM_LDW2], s_w3[CM_OUT * CM_H2], s_par[SP_TOTAL];
    __shared__ float s_z1[CM_TILE * CM_H1], s_z2[CM_TILE * CM_H2], s_h1[CM_TILE * CM_LDH], s_newc[CM_TILE * 2];
    __shared__ float s_red[CM_THREADS], s_mean[CM_H1], s_var[CM_H1], s_sc1[CM_H1], s_sh1[CM_H1], s_sc2[CM_H2], s_sh2[CM_H2];
    const int t = threadIdx.x, R = a.rows;
    const float* __restrict__ lm = a.lm;
    const float* __restrict__ coords = a.coords;
    // ---- every global load of the kernel
    const unsigned long long e = epoch_now(a.w.epoch);
    if (!(EG_CM_ABL & 4)) load_weights(a.w, s_w1, s_w2, s_w3);
    load_small_params(a.w, a.rm1, a.rv1, a.rm2, a.rv2, s_par);
    if (!(EG_CM_ABL & 8)) stage_inputs(lm, a.lm_stride, coords, 0, R, s_in, a.lm_copy);
    const float c_own = t < R * CM_OUT ? coords[t] : 0.f;
    a.w.seed1 += e;
    a.w.seed2 += e;
    __syncthreads();
    // ---- z1 = in W1^T + b1: thread -> (row t >> 5 and +32, output channel t & 31)
    {
        const int o = t & 31;
        const float b = s_par[SP_B1 + o];
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int r = (t >> 5) + 32 * half;
            if (r >= R) continue;                       // (rows past the end are nobody's operand: at batch 1 two waves of 16 have work)
            float acc = b;
            if (!(EG_CM_ABL & 1))
#pragma unroll 8
            for (int i = 0; i < CM_IN; ++i) acc = __builtin_fmaf(s_w1[o * CM_LDW1 + i], s_in[r * CM_LDI + i], acc);
            s_z1[r * CM_H1 + o] = acc;
            a.z1[r * CM_H1 + o] = acc;
        }
    }
    __syncthreads();
    bn_setup_lds<CM_H1>(s_z1, R, a.train, s_par + SP_G1, s_par + SP_BE1, s_par + SP_M1, s_par + SP_V1, a.eps1, a.mom1, a.rm1, a.rv1, a.bn,
                        a.bn + CM_H1, s_red, s_mean, s_var, s_sc1, s_sh1);
    // ---- z2 = h1 W2^T + b2: thread -> (row t >> 4, output channel t & 15)
    for (int el = t; el < R * CM_H1; el += CM_THREADS) {
        const int r = el >> 5, i = el & 31;
        float k;
        s_h1[r * CM_LDH + i] = hidden_act(s_z1[r * CM_H1 + i], s_sc1[i], s_sh1[i], a.w.seed1, (unsigned long long)r * CM_H1 + i, a.w.p1, a.w.ik1, &k);
    }
    __syncthreads();
    if ((t >> 4) < R) {
        const int r = t >> 4, o = t & 15;
        float acc = s_par[SP_B2 + o];
#pragma unroll
        for (int i = 0; i < CM_H1; ++i) acc = __builtin_fmaf(s_w2[o * CM_LDW2 + i], s_h1[r * CM_LDH + i], acc);
        s_z2[r * CM_H2 + o] = acc;
        a.z2[r * CM_H2 + o] = acc;
    }
    __syncthreads();
    bn_setup_lds<CM_H2>(s_z2, R, a.train, s_par + SP_G2, s_par + SP_BE2, s_par + SP_M2, s_par + SP_V2, a.eps2, a.mom2, a.rm2, a.rv2,
                        a.bn + 2 * CM_H1, a.bn + 2 * CM_H1 + CM_H2, s_red, s_mean, s_var, s_sc2, s_sh2);
    // ---- delta = h2 W3^T + b3; coords <- clamp(coords + delta)
    if (t < R * CM_OUT) {
        const int r = t >> 1, d = t & 1;
        float acc = s_par[SP_B3 + d];
#pragma unroll
        for (int i = 0; i < CM_H2; ++i) {
            float k;
            const float h = hidden_act(s_z2[r * CM_H2 + i], s_sc2[i], s_sh2[i], a.w.seed2, (unsigned long long)r * CM_H2 + i, a.w.p2, a.w.ik2, &k);
            acc = __builtin_fmaf(s_w3[d * CM_H2 + i], h, acc);
        }
        const float pre = c_own + acc;
        if (a.pre) a.pre[t] = pre;
        const float nc = fminf(fmaxf(pre, 0.f), a.cmax);
        a.newc[t] = nc;
        if (a.newc2) a.newc2[t] = nc;
        s_newc[t] = nc;
    }
    // ---- the coordinate rows resampled at the new positions (models.py:455-473): one wave per point
    if (sp.h) {
        __syncthreads();
        const int lane = t & 63;
        for (int p = t >> 6; p < R; p += CM_WAVES) {
            const int f = p >> 2;
            const f32x2 acc = bilinear_sample(sp.h + ((size_t)f * sp.n_per_frame + sp.main_base) * C + 2 * lane, s_newc[2 * p], s_newc[2 * p + 1], sp.frame);
            *reinterpret_cast<f32x2*>(sp.out + (size_t)f * sp.row_stride + (size_t)(p & 3) * C + 2 * lane) = acc;
        }
    }
}

// backward of the resampling, run in FRONT of the MLP's backward by the small form (h == NULL: the samples were not used)
struct CoordResampleBwd {
    const float* h;             // the tensor the samples were taken from
    const float* newc;          // [R,2] the positions
    float* dx;                  // the gradient tensor: its coordinate rows hold the samples' gradient (read here, before dlm overwrites
                                // them), its main-grid rows receive the taps
    long long n_per_frame, main_base, coord_base;
    int frame;
    TapSums ts;                 // ts.z != NULL: the additions' own BatchNorm-backward sums (coord_common.h)
};

// SMALL (R <= 64, batch <= 16): one tile -- z1 / z2 / the inputs are loaded once at the top, the scratch lives in LDS, the resampling's
// backward (k_bilinear4_bwd + the addition of its d coords, two launches of their own otherwise) runs first, one wave per frame.
// The general form round-trips the scratch through global memory between its phases: ~12 dependent round trips, 21 us at batch 1.
template <bool SMALL>
__global__ __launch_bounds__(CM_THREADS) void k_coord_mlp_bwd(const CoordMlpBwd a_, const CoordResampleBwd rb) {
    CoordMlpBwd a = a_;
    a.w = resolved(a_.w);
    __shared__ __attribute__((aligned(16))) float s_in[CM_TILE * CM_LDI];       // input tile; before that h1 | keep1 | xhat1 tiles
    __shared__ float s_w1[CM_H1 * CM_LDW1], s_w2[CM_H2 * CM_LDW2], s_w3[CM_OUT * CM_H2];
    __shared__ float s_dz[CM_TILE * CM_LDH];                                    // dz2 tile [64][17], later dz1 tile [64][33]
    __shared__ float s_red[4][CM_THREADS], s_part[CM_WAVES];
    __shared__ float s_gm[CM_H1], s_gxm[CM_H1], s_sc1[CM_H1], s_sh1[CM_H1], s_sc2[CM_H2], s_sh2[CM_H2];
    __shared__ float s_bn[2 * CM_H1 + 2 * CM_H2], s_g1[CM_H1], s_g2[CM_H2];    // mean1 | invstd1 | mean2 | invstd2, gamma1, gamma2
    __shared__ float s_z1l[SMALL ? CM_TILE * CM_H1 : 1], s_z2l[SMALL ? CM_TILE * CM_H2 : 1], s_scr[SMALL ? CM_TILE * CM_SCR : CM_KEEP * CM_SCR];
    __shared__ float s_hkx[SMALL ? 3 * CM_TILE * CM_LDH : 1], s_dd[SMALL ? CM_TILE * 2 : 1];
    const int t = threadIdx.x, R = a.rows;
    const float* __restrict__ lm = a.lm;
    const float* __restrict__ coords = a.coords;
    const float* __restrict__ z1 = SMALL ? s_z1l : a.z1;
    const float* __restrict__ z2 = SMALL ? s_z2l : a.z2;
    // (the general form: up to CM_KEEP rows -- batch 32 -- the scratch lives in LDS as well; a generic pointer then)
    float* const scratch = (SMALL || R <= CM_KEEP) ? s_scr : a.scratch;
    const float* mean1 = s_bn;
    const float* inv1 = s_bn + CM_H1;
    const float* mean2 = s_bn + 2 * CM_H1;
    const float* inv2 = s_bn + 2 * CM_H1 + CM_H2;
    // ---- every global load the first phases need
    load_weights(a.w, s_w1, s_w2, s_w3);
    if (t < CM_H1) {
        const float g = a.w.gamma1[t], m = a.bn[t], iv = a.bn[CM_H1 + t], sc = g * iv;
        s_g1[t] = g; s_bn[t] = m; s_bn[CM_H1 + t] = iv;
        s_sc1[t] = sc;
        s_sh1[t] = a.w.beta1[t] - m * sc;
    } else if (t < CM_H1 + CM_H2) {
        const int c = t - CM_H1;
        const float g = a.w.gamma2[c], m = a.bn[2 * CM_H1 + c], iv = a.bn[2 * CM_H1 + CM_H2 + c], sc = g * iv;
        s_g2[c] = g; s_bn[2 * CM_H1 + c] = m; s_bn[2 * CM_H1 + CM_H2 + c] = iv;
        s_sc2[c] = sc;
        s_sh2[c] = a.w.beta2[c] - m * sc;
    }
    if (SMALL) {
        {
            const float za = t < R * CM_H1 ? a.z1[t] : 0.f, zb = t + CM_THREADS < R * CM_H1 ? a.z1[t + CM_THREADS] : 0.f;
            const float zc = t < R * CM_H2 ? a.z2[t] : 0.f;
            s_z1l[t] = za; s_z1l[t + CM_THREADS] = zb; s_z2l[t] = zc;
        }
        stage_inputs(lm, a.lm_stride, coords, 0, R, s_in);
        float dcur = 0.f, pre = 0.f;
        if (t < R * 2) { dcur = a.dnew ? a.dnew[t] : 0.f; pre = a.pre[t]; }
        if (rb.h) {
            // the resampling's backward: wave = frame (batch <= 16 = the workgroup's waves)
            const int lane = t & 63, f = __builtin_amdgcn_readfirstlane(t >> 6);
            if (f * 4 < R) {
                const size_t fbase = ((size_t)f * rb.n_per_frame + rb.main_base) * C;
                const float* dout_f = rb.dx + ((size_t)f * rb.n_per_frame + rb.coord_base) * C;
                float gh[4], gw[4];
                f32x2 ts1 = {0.f, 0.f}, ts2 = ts1;
                if (rb.ts.z) {
                    const unsigned long long tseed = rb.ts.seed + epoch_now(rb.ts.epoch);
                    bilinear_bwd_frame4<true>(dout_f, rb.h, rb.newc + 8 * f, rb.dx, fbase, rb.frame, rb.ts, tseed, lane, gh, gw, ts1, ts2);
                    *reinterpret_cast<f32x2*>(rb.ts.out + (size_t)f * 2 * C + 2 * lane) = ts1;
                    *reinterpret_cast<f32x2*>(rb.ts.out + (size_t)f * 2 * C + C + 2 * lane) = ts2;
                } else {
                    bilinear_bwd_frame4<false>(dout_f, rb.h, rb.newc + 8 * f, rb.dx, fbase, rb.frame, rb.ts, 0ull, lane, gh, gw, ts1, ts2);
                }
                if (lane < 8) {
                    float vsel = 0.f;
#pragma unroll
                    for (int q = 0; q < 4; ++q) { vsel = lane == 2 * q ? gh[q] : vsel; vsel = lane == 2 * q + 1 ? gw[q] : vsel; }
                    s_dd[8 * f + lane] = vsel;
                }
            }
            __syncthreads();
            if (t < R * 2) dcur += s_dd[t];
            __syncthreads();
        }
        if (t < R * 2) s_dd[t] = (pre >= 0.f && pre <= a.cmax) ? dcur : 0.f;      // gradient of the clamp (torch.clamp: passes where min <= x <= max)
    }
    __syncthreads();
    auto dd = [&](int r, int d) -> float {
        if (SMALL) return s_dd[r * 2 + d];
        const float pre = a.pre[r * 2 + d];
        const float g = (a.dnew ? a.dnew[r * 2 + d] : 0.f) + (a.dnew2 ? a.dnew2[r * 2 + d] : 0.f);
        return (pre >= 0.f && pre <= a.cmax) ? g : 0.f;
    };
    // ---- output layer: g2 = (dd W3) mask2, its BatchNorm sums, dW3, db3
    {
        const int c = t & 15, part = t >> 4;
        float sg = 0.f, sgx = 0.f, dw0 = 0.f, dw1 = 0.f;
        for (int r = part; r < R; r += CM_THREADS / CM_H2) {
            const float z = z2[r * CM_H2 + c], d0 = dd(r, 0), d1 = dd(r, 1);
            float k;
            const float h = hidden_act(z, s_sc2[c], s_sh2[c], a.w.seed2, (unsigned long long)r * CM_H2 + c, a.w.p2, a.w.ik2, &k);
            const float g = (d0 * s_w3[c] + d1 * s_w3[CM_H2 + c]) * k;
            sg += g;
            sgx += g * ((z - mean2[c]) * inv2[c]);
            dw0 += d0 * h;
            dw1 += d1 * h;
        }
        s_red[0][t] = sg; s_red[1][t] = sgx; s_red[2][t] = dw0; s_red[3][t] = dw1;
        __syncthreads();
        if (t < CM_H2) {
            const float g = reduce_parts<CM_H2>(s_red[0], t), gx = reduce_parts<CM_H2>(s_red[1], t);
            s_gm[t] = g / (float)R;
            s_gxm[t] = gx / (float)R;
            a.grads[CG_DG2 + t] = gx;
            a.grads[CG_DBE2 + t] = g;
            a.grads[CG_DB2 + t] = 0.f;                       // a bias in front of a train-mode BatchNorm
        } else if (t >= 64 && t < 64 + CM_H2) {
            a.grads[CG_DW3 + t - 64] = reduce_parts<CM_H2>(s_red[2], t - 64);
        } else if (t >= 128 && t < 128 + CM_H2) {
            a.grads[CG_DW3 + CM_H2 + t - 128] = reduce_parts<CM_H2>(s_red[3], t - 128);
        }
        float v = 0.f;                                       // db3[d] = sum_r dd(r, d): element parity == thread parity
        for (int e = t; e < R * 2; e += CM_THREADS) v += dd(e >> 1, e & 1);
        const float db0 = block_sum((t & 1) ? 0.f : v, s_part);
        const float db1 = block_sum((t & 1) ? v : 0.f, s_part);
        if (t == 0) { a.grads[CG_DB3] = db0; a.grads[CG_DB3 + 1] = db1; }
        // dz2 = gamma2 invstd2 (g2 - mean g2 - xhat2 mean(g2 xhat2))
        for (int e = t; e < R * CM_H2; e += CM_THREADS) {
            const int r = e >> 4, cc = e & 15;
            const float z = z2[e];
            float k;
            hidden_act(z, s_sc2[cc], s_sh2[cc], a.w.seed2, (unsigned long long)e, a.w.p2, a.w.ik2, &k);
            const float g = (dd(r, 0) * s_w3[cc] + dd(r, 1) * s_w3[CM_H2 + cc]) * k;
            const float xh = (z - mean2[cc]) * inv2[cc];
            scratch[r * CM_SCR + cc] = s_g2[cc] * inv2[cc] * (g - s_gm[cc] - xh * s_gxm[cc]);
        }
        __syncthreads();
    }
    // ---- hidden layer 2, 64 rows at a time: dW2 = dz2^T h1;  g1 = (dz2 W2) mask1 (-> scratch) and its BatchNorm sums
    {
        float* s_h1 = SMALL ? s_hkx : s_in;                                  // (SMALL: s_in keeps the input tile staged at the top)
        float* s_k1 = s_h1 + CM_TILE * CM_LDH;
        float* s_x1 = s_h1 + 2 * CM_TILE * CM_LDH;
        const int wo = (t & 511) >> 5, wi = t & 31, whalf = t >> 9;         // dW2[wo][wi], rows 32 * whalf .. + 32 of the tile
        const int c = t & 31, part = t >> 5;                                 // g1 column c, rows part and part + 32 of the tile
        float dw2 = 0.f, sg = 0.f, sgx = 0.f;
        for (int row0 = 0; row0 < R; row0 += CM_TILE) {
            __syncthreads();
            for (int e = t; e < CM_TILE * CM_H1; e += CM_THREADS) {
                const int r = e >> 5, i = e & 31, row = row0 + r;
                float k = 0.f, h = 0.f, xh = 0.f;
                if (row < R) {
                    const float z = z1[row * CM_H1 + i];
                    h = hidden_act(z, s_sc1[i], s_sh1[i], a.w.seed1, (unsigned long long)row * CM_H1 + i, a.w.p1, a.w.ik1, &k);
                    xh = (z - mean1[i]) * inv1[i];
                }
                s_h1[r * CM_LDH + i] = h;
                s_k1[r * CM_LDH + i] = k;
                s_x1[r * CM_LDH + i] = xh;
            }
            {
                const int r = t >> 4, o = t & 15;
                s_dz[r * CM_LDZ2 + o] = row0 + r < R ? scratch[(row0 + r) * CM_SCR + o] : 0.f;
            }
            __syncthreads();
            const int rows_here = R - row0 < CM_TILE ? R - row0 : CM_TILE;      // (rows past the end are zeros in every operand: skipped)
            const int r_hi = 32 * whalf + 32 < rows_here ? 32 * whalf + 32 : rows_here;
#pragma unroll 8
            for (int r = 32 * whalf; r < r_hi; ++r) dw2 += s_dz[r * CM_LDZ2 + wo] * s_h1[r * CM_LDH + wi];
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int r = part + 32 * half;
                if (r >= rows_here) continue;
                float gp = 0.f;
#pragma unroll
                for (int o = 0; o < CM_H2; ++o) gp += s_dz[r * CM_LDZ2 + o] * s_w2[o * CM_LDW2 + c];
                const float g = gp * s_k1[r * CM_LDH + c];
                sg += g;
                sgx += g * s_x1[r * CM_LDH + c];
                scratch[(row0 + r) * CM_SCR + CM_H2 + c] = g;
            }
        }
        s_red[0][t] = sg; s_red[1][t] = sgx; s_red[2][t] = dw2;
        __syncthreads();
        if (t < CM_H1) {
            const float g = reduce_parts<CM_H1>(s_red[0], t), gx = reduce_parts<CM_H1>(s_red[1], t);
            s_gm[t] = g / (float)R;
            s_gxm[t] = gx / (float)R;
            a.grads[CG_DG1 + t] = gx;
            a.grads[CG_DBE1 + t] = g;
            a.grads[CG_DB1 + t] = 0.f;
        }
        if (t < 512) a.grads[CG_DW2 + t] = s_red[2][t] + s_red[2][t + 512];
        __syncthreads();
        // dz1 = gamma1 invstd1 (g1 - mean g1 - xhat1 mean(g1 xhat1)), in place in the scratch
        for (int e = t; e < R * CM_H1; e += CM_THREADS) {
            const int r = e >> 5, cc = e & 31;
            const float xh = (z1[e] - mean1[cc]) * inv1[cc];
            float* p = scratch + r * CM_SCR + CM_H2 + cc;
            *p = s_g1[cc] * inv1[cc] * (*p - s_gm[cc] - xh * s_gxm[cc]);
        }
    }
    // ---- first layer, 64 rows at a time: dW1 = dz1^T in;  d in = dz1 W1 -> d lm, d shape_feats
    {
        // dW1 work item = (o, 4 consecutive inputs): 32 * 34 = 1088 items; threads 0..63 take a second one
        const int it0 = t, it1 = CM_THREADS + t;
        const bool two = it1 < CM_H1 * (CM_IN / 4);
        const int o0 = it0 / 34, q0 = it0 - 34 * o0, o1 = two ? it1 / 34 : 0, q1 = two ? it1 - 34 * o1 : 0;
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        for (int row0 = 0; row0 < R; row0 += CM_TILE) {
            __syncthreads();
            if (!SMALL) stage_inputs(lm, a.lm_stride, coords, row0, R, s_in);
            for (int e = t; e < CM_TILE * CM_H1; e += CM_THREADS) {
                const int r = e >> 5, i = e & 31;
                s_dz[r * CM_LDH + i] = row0 + r < R ? scratch[(row0 + r) * CM_SCR + CM_H2 + i] : 0.f;
            }
            __syncthreads();
            const int rows_here = R - row0 < CM_TILE ? R - row0 : CM_TILE;
#pragma unroll 4
            for (int r = 0; r < rows_here; ++r) {
                acc0 += s_dz[r * CM_LDH + o0] * *reinterpret_cast<const f32x4*>(&s_in[r * CM_LDI + 4 * q0]);
                if (two) acc1 += s_dz[r * CM_LDH + o1] * *reinterpret_cast<const f32x4*>(&s_in[r * CM_LDI + 4 * q1]);
            }
            if (a.dlm) {                                  // thread -> column t & 127, rows (t >> 7) + 8 j
                const int i = t & 127;
                for (int r = t >> 7; r < CM_TILE && row0 + r < R; r += 8) {
                    float acc = 0.f;
#pragma unroll
                    for (int o = 0; o < CM_H1; ++o) acc += s_dz[r * CM_LDH + o] * s_w1[o * CM_LDW1 + i];
                    float* dp = a.dlm + (size_t)((row0 + r) >> 2) * a.dlm_stride + ((row0 + r) & 3) * C + i;
                    *dp = a.dlm_acc ? *dp + acc : acc;
                }
            }
            if (t < rows_here * 8) {
                const int r = t >> 3, i = C + (t & 7);
                float acc = 0.f;
#pragma unroll
                for (int o = 0; o < CM_H1; ++o) acc += s_dz[r * CM_LDH + o] * s_w1[o * CM_LDW1 + i];
                scratch[(row0 + r) * CM_SCR + CM_H2 + CM_H1 + (t & 7)] = acc;
            }
        }
        *reinterpret_cast<f32x4*>(a.grads + CG_DW1 + o0 * CM_IN + 4 * q0) = acc0;
        if (two) *reinterpret_cast<f32x4*>(a.grads + CG_DW1 + o1 * CM_IN + 4 * q1) = acc1;
        __syncthreads();
    }
    // ---- d coords: through the clamp directly, as "the other landmark" k = m of all 4 rows of the frame (+), and as
    //      "self" of row m for all k (-)
    if (a.dcoords) {
        const float* dsf = scratch + CM_H2 + CM_H1;
        for (int e = t; e < R * 2; e += CM_THREADS) {
            const int r = e >> 1, d = e & 1, f4 = r & ~3, m = r & 3;
            float acc = dd(r, d);
            for (int j = 0; j < 4; ++j) acc += dsf[(f4 + j) * CM_SCR + 2 * m + d];
            for (int k = 0; k < 4; ++k) acc -= dsf[r * CM_SCR + 2 * k + d];
            a.dcoords[e] = acc;
        }
    }
}

static CoordMlpW weights_of(const eg_cls_train_params* P, bool train) {
    CoordMlpW w{P->w1, P->b1, P->gamma1, P->beta1, P->w2, P->b2, P->gamma2, P->beta2, P->w3, P->b3};
    w.p1 = train ? P->p1 : 0.f;
    w.p2 = train ? P->p2 : 0.f;
    w.ik1 = w.p1 > 0.f ? 1.0f / (1.0f - w.p1) : 1.0f;
    w.ik2 = w.p2 > 0.f ? 1.0f / (1.0f - w.p2) : 1.0f;
    w.seed1 = P->seed1;
    w.seed2 = P->seed2;
    w.epoch = eg_epoch_ptr();
    return w;
}

static bool params_ok(const eg_cls_train_params* P) {
    return P && P->w1 && P->b1 && P->gamma1 && P->beta1 && P->w2 && P->b2 && P->gamma2 && P->beta2 && P->w3 && P->b3;
}

}  // namespace eg

using namespace eg;

extern "C" {

static int coord_mlp_fwd_rows(const float* lm, int64_t lm_frame_stride, float* lm_copy, const float* coords, int batch,
                              const eg_cls_train_params* P, int train, int frame, float* z1, float* z2, float* bn, float* pre,
                              float* new_coords, float* new_coords2, eg_stream_t stream) {
    if (!lm || !coords || !z1 || !z2 || !bn || !new_coords || !params_ok(P)) return set_error(EG_ERR_ARG, "NULL argument");
    if (batch < 1 || frame < 1 || batch > (1 << 20) || lm_frame_stride < 4 * C) return set_error(EG_ERR_ARG, "bad batch / frame / stride");
    if (!train && (!P->running_mean1 || !P->running_var1 || !P->running_mean2 || !P->running_var2))
        return set_error(EG_ERR_ARG, "eval mode needs the running statistics");
    if (!(P->p1 >= 0.f && P->p1 < 1.f && P->p2 >= 0.f && P->p2 < 1.f)) return set_error(EG_ERR_ARG, "dropout p must be in [0, 1)");
    if (train) { if (int rc_ = eg_epoch_required(P->p1 > P->p2 ? P->p1 : P->p2)) return rc_; }
    CoordMlpFwd a{};
    a.lm = lm; a.lm_stride = lm_frame_stride; a.lm_copy = lm_copy; a.coords = coords; a.rows = 4 * batch; a.train = train ? 1 : 0;
    a.w = weights_of(P, train != 0);
    a.rm1 = P->running_mean1; a.rv1 = P->running_var1; a.rm2 = P->running_mean2; a.rv2 = P->running_var2;
    a.eps1 = P->eps1; a.eps2 = P->eps2; a.mom1 = P->momentum1; a.mom2 = P->momentum2; a.cmax = (float)(frame - 1);
    a.z1 = z1; a.z2 = z2; a.bn = bn; a.pre = pre; a.newc = new_coords; a.newc2 = new_coords2;
    if (a.rows <= CM_TILE) hipLaunchKernelGGL(k_coord_update_fwd_small, dim3(1), dim3(CM_THREADS), 0, (hipStream_t)stream, a, CoordSample{});
    else hipLaunchKernelGGL(k_coord_mlp_fwd, dim3(1), dim3(CM_THREADS), 0, (hipStream_t)stream, a);
    EG_HIP_TRY(hipGetLastError());
    return EG_OK;
}

int eg_coord_mlp_fwd_rows(const float* lm, int64_t lm_frame_stride, float* lm_copy, const float* coords, int batch,
                          const eg_cls_train_params* P, int train, int frame, float* z1, float* z2, float* bn, float* pre,
                          float* new_coords, eg_stream_t stream) {
    return coord_mlp_fwd_rows(lm, lm_frame_stride, lm_copy, coords, batch, P, train, frame, z1, z2, bn, pre, new_coords, nullptr, stream);
}

int eg_coord_update_fwd(float* h, int64_t n_per_frame, int64_t coord_base, int64_t main_base, const float* coords, int batch,
                        const eg_cls_train_params* P, int train, int frame, int resample, float* lm_copy, float* z1, float* z2, float* bn,
                        float* pre, float* new_coords, float* new_coords2, eg_stream_t stream) {
    if (!h) return set_error(EG_ERR_ARG, "NULL argument");
    if (batch < 1 || frame < 1 || coord_base < 0 || coord_base + 4 > n_per_frame || main_base < 0 || main_base + (int64_t)frame * frame > n_per_frame ||
        (coord_base < main_base + (int64_t)frame * frame && coord_base + 4 > main_base))
        return set_error(EG_ERR_ARG, "bad batch / frame / row ranges");
    float* rows = h + (size_t)coord_base * C;
    if (!resample || 4 * batch > CM_TILE) {
        if (int rc = coord_mlp_fwd_rows(rows, n_per_frame * C, lm_copy, coords, batch, P, train, frame, z1, z2, bn, pre, new_coords, new_coords2, stream)) return rc;
        if (!resample) return EG_OK;
        return eg_bilinear4_fwd_rows(h, new_coords, batch, 4, n_per_frame, main_base, frame, rows, n_per_frame * C, stream);
    }
    // (argument checks of the MLP: the same as eg_coord_mlp_fwd_rows)
    if (!coords || !z1 || !z2 || !bn || !new_coords || !params_ok(P)) return set_error(EG_ERR_ARG, "NULL argument");
    if (!train && (!P->running_mean1 || !P->running_var1 || !P->running_mean2 || !P->running_var2))
        return set_error(EG_ERR_ARG, "eval mode needs the running statistics");
    if (!(P->p1 >= 0.f && P->p1 < 1.f && P->p2 >= 0.f && P->p2 < 1.f)) return set_error(EG_ERR_ARG, "dropout p must be in [0, 1)");
    if (train) { if (int rc_ = eg_epoch_required(P->p1 > P->p2 ? P->p1 : P->p2)) return rc_; }
    CoordMlpFwd a{};
    a.lm = rows; a.lm_stride = n_per_frame * C; a.lm_copy = lm_copy; a.coords = coords; a.rows = 4 * batch; a.train = train ? 1 : 0;
    a.w = weights_of(P, train != 0);
    a.rm1 = P->running_mean1; a.rv1 = P->running_var1; a.rm2 = P->running_mean2; a.rv2 = P->running_var2;
    a.eps1 = P->eps1; a.eps2 = P->eps2; a.mom1 = P->momentum1; a.mom2 = P->momentum2; a.cmax = (float)(frame - 1);
    a.z1 = z1; a.z2 = z2; a.bn = bn; a.pre = pre; a.newc = new_coords; a.newc2 = new_coords2;
    const CoordSample sp{h, rows, (long long)n_per_frame, (long long)main_base, (long long)n_per_frame * C, frame};
    hipLaunchKernelGGL(k_coord_update_fwd_small, dim3(1), dim3(CM_THREADS), 0, (hipStream_t)stream, a, sp);
    EG_HIP_TRY(hipGetLastError());
    return EG_OK;
}

int eg_coord_mlp_fwd(const float* lm, const float* coords, int batch, const eg_cls_train_params* P, int train, int frame,
                     float* z1, float* z2, float* bn, float* pre, float* new_coords, eg_stream_t stream) {
    return eg_coord_mlp_fwd_rows(lm, 4 * C, nullptr, coords, batch, P, train, frame, z1, z2, bn, pre, new_coords, stream);
}

int eg_coord_mlp_bwd_rows(const float* dnew_coords, const float* lm, const float* coords, int batch, const eg_cls_train_params* P,
                          int frame, const float* z1, const float* z2, const float* bn, const float* pre, float* scratch, float* dlm,
                          int64_t dlm_frame_stride, int dlm_accumulate, float* dcoords, float* grads, eg_stream_t stream) {
    if (!dnew_coords || !lm || !coords || !z1 || !z2 || !bn || !pre || !scratch || !grads || !params_ok(P))
        return set_error(EG_ERR_ARG, "NULL argument");
    if (batch < 1 || frame < 1 || batch > (1 << 20) || dlm_frame_stride < 4 * C) return set_error(EG_ERR_ARG, "bad batch / frame / stride");
    if (int rc_ = eg_epoch_required(P->p1 > P->p2 ? P->p1 : P->p2)) return rc_;
    CoordMlpBwd a{};
    a.dnew = dnew_coords; a.lm = lm; a.lm_stride = 4 * C; a.coords = coords; a.pre = pre; a.z1 = z1; a.z2 = z2; a.bn = bn; a.rows = 4 * batch;
    a.w = weights_of(P, true);
    a.cmax = (float)(frame - 1);
    a.scratch = scratch; a.dlm = dlm; a.dlm_stride = dlm_frame_stride; a.dlm_acc = dlm_accumulate ? 1 : 0; a.dcoords = dcoords; a.grads = grads;
    if (a.rows <= CM_TILE) hipLaunchKernelGGL(k_coord_mlp_bwd<true>, dim3(1), dim3(CM_THREADS), 0, (hipStream_t)stream, a, CoordResampleBwd{});
    else hipLaunchKernelGGL(k_coord_mlp_bwd<false>, dim3(1), dim3(CM_THREADS), 0, (hipStream_t)stream, a, CoordResampleBwd{});
    EG_HIP_TRY(hipGetLastError());
    return EG_OK;
}

int eg_coord_update_bwd(float* dx, int64_t n_per_frame, int64_t coord_base, int64_t main_base, const float* h, const float* new_coords,
                        const float* dnew_coords, const float* lm, const float* coords, int batch, const eg_cls_train_params* P, int frame,
                        const float* z1, const float* z2, const float* bn, const float* pre, float* scratch, float* dbil,
                        const eg_lower_sums* lower, float* tap_sums, float* dcoords, float* grads, eg_stream_t stream) {
    if (!dx || !h || !new_coords || !lm || !coords || !z1 || !z2 || !bn || !pre || !scratch || !dbil || !grads || !params_ok(P))
        return set_error(EG_ERR_ARG, "NULL argument");
    if (batch < 1 || frame < 1 || batch > (1 << 20) || coord_base < 0 || coord_base + 4 > n_per_frame || main_base < 0 ||
        main_base + (int64_t)frame * frame > n_per_frame || (coord_base < main_base + (int64_t)frame * frame && coord_base + 4 > main_base))
        return set_error(EG_ERR_ARG, "bad batch / frame / row ranges");
    if (lower && (!lower->z || !lower->bn || !tap_sums)) return set_error(EG_ERR_ARG, "NULL argument (lower sums)");
    if (lower && (lower->dropout_p < 0.f || lower->dropout_p >= 1.f)) return set_error(EG_ERR_ARG, "dropout_p must be in [0, 1)");
    float* rows = dx + (size_t)coord_base * C;
    if (4 * batch > CM_TILE) {
        // three launches: the resampling's backward (d coords -> dbil), then the MLP's with d new_coords = dnew_coords + dbil
        int rc = lower ? eg_bilinear4_bwd_rows_sums(rows, n_per_frame * C, h, new_coords, batch, 4, n_per_frame, main_base, frame, dx, dbil, lower, tap_sums, stream)
                       : eg_bilinear4_bwd_rows(rows, n_per_frame * C, h, new_coords, batch, 4, n_per_frame, main_base, frame, dx, dbil, stream);
        if (rc) return rc;
        if (int rc_ = eg_epoch_required(P->p1 > P->p2 ? P->p1 : P->p2)) return rc_;
        CoordMlpBwd a{};
        a.dnew = dnew_coords; a.dnew2 = dbil; a.lm = lm; a.lm_stride = 4 * C; a.coords = coords; a.pre = pre; a.z1 = z1; a.z2 = z2; a.bn = bn; a.rows = 4 * batch;
        a.w = weights_of(P, true);
        a.cmax = (float)(frame - 1);
        a.scratch = scratch; a.dlm = rows; a.dlm_stride = n_per_frame * C; a.dlm_acc = 0; a.dcoords = dcoords; a.grads = grads;
        hipLaunchKernelGGL(k_coord_mlp_bwd<false>, dim3(1), dim3(CM_THREADS), 0, (hipStream_t)stream, a, CoordResampleBwd{});
        EG_HIP_TRY(hipGetLastError());
        return EG_OK;
    }
    float pmax = P->p1 > P->p2 ? P->p1 : P->p2;
    if (lower && lower->dropout_p > pmax) pmax = lower->dropout_p;
    if (int rc_ = eg_epoch_required(pmax)) return rc_;
    CoordMlpBwd a{};
    a.dnew = dnew_coords; a.lm = lm; a.lm_stride = 4 * C; a.coords = coords; a.pre = pre; a.z1 = z1; a.z2 = z2; a.bn = bn; a.rows = 4 * batch;
    a.w = weights_of(P, true);
    a.cmax = (float)(frame - 1);
    a.scratch = scratch; a.dlm = rows; a.dlm_stride = n_per_frame * C; a.dlm_acc = 0; a.dcoords = dcoords; a.grads = grads;
    CoordResampleBwd rb{h, new_coords, dx, (long long)n_per_frame, (long long)main_base, (long long)coord_base, frame, TapSums{}};
    if (lower)
        rb.ts = TapSums{lower->z, lower->bn, lower->relu, lower->dropout_p, lower->dropout_p > 0.f ? 1.0f / (1.0f - lower->dropout_p) : 1.0f,
                        (unsigned long long)lower->seed, eg_epoch_ptr(), tap_sums};
    hipLaunchKernelGGL(k_coord_mlp_bwd<true>, dim3(1), dim3(CM_THREADS), 0, (hipStream_t)stream, a, rb);
    EG_HIP_TRY(hipGetLastError());
    return EG_OK;
}

int eg_coord_mlp_bwd(const float* dnew_coords, const float* lm, const float* coords, int batch, const eg_cls_train_params* P,
                     int frame, const float* z1, const float* z2, const float* bn, const float* pre, float* scratch, float* dlm,
                     float* dcoords, float* grads, eg_stream_t stream) {
    return eg_coord_mlp_bwd_rows(dnew_coords, lm, coords, batch, P, frame, z1, z2, bn, pre, scratch, dlm, 4 * C, 0, dcoords, grads, stream);
}

}  // extern "C"
