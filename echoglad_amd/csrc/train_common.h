// Helpers shared by the train-mode kernels (train.hip, cls_train.hip): counter-based dropout mask, deterministic
// two-stage column reductions, train-mode BatchNorm finalisation.
#pragma once
#include "tile.h"

namespace eg {

// ---- counter-based dropout mask: keep iff hash(seed, element) >= p ---------------------------------
__device__ inline float keep_scale(unsigned long long seed, unsigned long long idx, float p, float inv_keep) {
    unsigned long long z = seed + idx * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    const float u = (float)(unsigned)(z >> 40) * (1.0f / 16777216.0f);     // 24 random bits -> [0,1)
    return u >= p ? inv_keep : 0.0f;
}

// Compact row r of a per-frame row filter -> row of the unfiltered [batch * stride, .] array:
//   (r / n_valid) * stride + lo + r % n_valid        (models.py:485: the node-type filter keeps a contiguous range per frame)
struct RowMap {
    int n_valid, stride, lo;
};
__device__ inline long long map_row(const RowMap& m, long long r) {
    const long long f = r / m.n_valid;
    return f * m.stride + m.lo + (r - f * m.n_valid);
}

// stage 2 of every column reduction: partial float [nblocks][n] -> totals double [n], fixed order (bitwise reproducible)
__global__ void k_reduce_f32_partials(const float* __restrict__ partial, int nblocks, int n, double* __restrict__ totals);

// Train-mode BatchNorm1d from column totals: totals[0..cc) = sum, totals[cc..2cc) = sum of squares over `rows` rows.
//   mean, invstd = 1/sqrt(var_biased + eps), scale = gamma * invstd, shift = beta - mean * scale,
//   running <- (1 - momentum) running + momentum * {mean, var_unbiased}   (momentum < 0 or NULL pointers: no update)
struct BnFinalize {
    const double* totals;
    long long rows;
    int cc;
    const float *gamma, *beta;
    float eps, momentum;
    float *running_mean, *running_var;
    float *mean, *invstd, *scale, *shift;
};
__global__ void k_bn_finalize(const BnFinalize a);

}  // namespace eg
