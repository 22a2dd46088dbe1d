// Helpers shared by the train-mode kernels (train.hip, cls_train.hip): counter-based dropout mask, deterministic
// two-stage column reductions, train-mode BatchNorm finalisation.
#pragma once
#include "tile.h"

namespace eg {

// ---- counter-based dropout mask -------------------------------------------------------------------------------------------
// One 64-bit hash (splitmix64 finaliser of seed + group index) serves the 4 consecutive elements 4g .. 4g + 3: element e keeps
// iff the 16-bit field (e & 3) of hash(e >> 2) is >= floor(65536 p).  A pure function of (seed, element index): forward and
// backward kernels regenerate the same mask, nothing is stored.  (The 64-bit multiplies are quarter-rate VALU work; one hash
// per element cost 0.7 ms of a 20 ms training step.)
__device__ inline unsigned long long mask_word(unsigned long long seed, unsigned long long group) {
#ifdef EG_ABL_HASH          // timing-only ablation: what would a free mask be worth?  (not a usable mask)
    return (seed ^ group) * 0x0001000100010001ull;
#endif
    unsigned long long z = seed + group * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__device__ inline float keep_field(unsigned long long word, int u, unsigned threshold, float inv_keep) {
    return ((unsigned)(word >> (16 * u)) & 0xFFFFu) >= threshold ? inv_keep : 0.0f;
}
__device__ inline unsigned drop_threshold(float p) { return (unsigned)(p * 65536.0f); }

__device__ inline float keep_scale(unsigned long long seed, unsigned long long idx, float p, float inv_keep) {
    return keep_field(mask_word(seed, idx >> 2), (int)(idx & 3), drop_threshold(p), inv_keep);
}
// the 4 elements idx .. idx + 3, idx a multiple of 4
__device__ inline f32x4 keep_scale4(unsigned long long seed, unsigned long long idx, float p, float inv_keep) {
    const unsigned long long w = mask_word(seed, idx >> 2);
    const unsigned thr = drop_threshold(p);
    return f32x4{keep_field(w, 0, thr, inv_keep), keep_field(w, 1, thr, inv_keep), keep_field(w, 2, thr, inv_keep),
                 keep_field(w, 3, thr, inv_keep)};
}
// the keep flags of the 4 elements idx .. idx + 3 as bits 0..3 (for a kernel that needs the same mask twice, far apart: 4 bits to
// hold instead of a second hash), and the factors back from them
__device__ inline unsigned keep_bits4(unsigned long long seed, unsigned long long idx, float p) {
    const unsigned long long w = mask_word(seed, idx >> 2);
    const unsigned thr = drop_threshold(p);
    unsigned b = 0;
#pragma unroll
    for (int u = 0; u < 4; ++u) b |= (((unsigned)(w >> (16 * u)) & 0xFFFFu) >= thr ? 1u : 0u) << u;
    return b;
}
__device__ inline f32x4 keep_scale_of_bits(unsigned bits, float inv_keep) {
    return f32x4{(bits & 1u) ? inv_keep : 0.0f, (bits & 2u) ? inv_keep : 0.0f, (bits & 4u) ? inv_keep : 0.0f, (bits & 8u) ? inv_keep : 0.0f};
}
// the 2 elements idx, idx + 1, idx even
__device__ inline f32x2 keep_scale2(unsigned long long seed, unsigned long long idx, float p, float inv_keep) {
    const unsigned long long w = mask_word(seed, idx >> 2);
    const unsigned thr = drop_threshold(p);
    const int u = (int)(idx & 2);
    return f32x2{keep_field(w, u, thr, inv_keep), keep_field(w, u + 1, thr, inv_keep)};
}

// The seed a kernel hashes with is  seed (the caller's, a kernel argument) + EPOCH (one 64-bit word per device, read from
// device memory at the top of every kernel that applies or regenerates a mask; 0 unless eg_dropout_epoch_* moved it).  A train
// step captured into a HIP graph freezes its kernel arguments -- the epoch word, bumped by a node of the graph itself, is what
// gives every replay fresh masks (echoglad_amd/engine.py GraphedTrainStep).  Host side: train.hip.
const unsigned long long* eg_epoch_ptr();            // this device's epoch word (allocated and zeroed at first use; NULL on failure)
// EG_OK, or EG_ERR_HIP with the error text set when a launch that draws a mask (p > 0) cannot get the epoch word: with a NULL
// pointer the kernels would hash with epoch 0 -- every replay of a captured step the same masks, and no error anywhere
int eg_epoch_required(float dropout_p);

// arguments of the BN affine + dropout + ReLU (+ residual) activation
struct ActArgs {
    long long rows;
    int relu;
    float p, inv_keep;
    unsigned long long seed;
    const unsigned long long* epoch;                 // (host: make_act / the launchers; device: resolved() at the top of the kernel)
};
__device__ inline unsigned long long epoch_now(const unsigned long long* epoch) { return epoch ? *epoch : 0ull; }
__device__ inline ActArgs resolved(const ActArgs& a) {
    ActArgs r = a;
    r.seed = a.seed + epoch_now(a.epoch);
    return r;
}

// ---- "last workgroup out" --------------------------------------------------------------------------------------------------------
// A kernel whose workgroups leave partial results for a small epilogue (a reduction of the partials, a finalisation) runs the epilogue
// itself, in whichever workgroup arrives last, instead of leaving it to a launch of its own: every node of a captured batch-1 training
// step costs ~4.5 us whatever it does.  Every workgroup calls this after its last global store of what the epilogue reads (all threads,
// uniformly); it returns true in exactly one workgroup, with the other workgroups' stores visible to all of its threads.  *ticket is 0
// before the launch and 0 again after it (atomicInc wraps at total - 1): no reset, no memset node.  The epilogue must read the
// partials in a fixed order (never "in arrival order"), so the result does not depend on which workgroup runs it.
__device__ inline bool last_workgroup_out(unsigned* ticket, unsigned total) {
    // ONE thread fences per workgroup: a device-scope release writes the XCD's L2 back (8 XCDs, an L2 each) -- with every thread of
    // 512 workgroups doing it a 10-us reduction took 124 us.  The barrier in front makes the workgroup's stores happen-before thread 0's
    // release (cumulative); so: worth it for a handful of workgroups, not for hundreds.
    __shared__ unsigned s_last_out;
    __syncthreads();
    if (threadIdx.x == 0 && threadIdx.y == 0 && threadIdx.z == 0) {
        __threadfence();                              // release, device scope
        const unsigned last = atomicInc(ticket, total - 1) == total - 1 ? 1u : 0u;
        if (last) __threadfence();                    // acquire: the others' stores
        s_last_out = last;
    }
    __syncthreads();
    return s_last_out != 0u;
}
// host side (train.hip): a zeroed device word per (device, stream, slot); allocated at first use -- never inside a stream capture, a
// warm-up step in front of the capture has been here (like eg_epoch_ptr).  NULL on failure.
constexpr int EG_TICKET_SLOTS = 16;
unsigned* eg_ticket_ptr(void* stream, int slot);

// non-temporal 16-B load of a streaming pass (every operand is read once)
__device__ inline f32x4 ldnt4(const float* p) {
    return f32x4{__builtin_nontemporal_load(p), __builtin_nontemporal_load(p + 1), __builtin_nontemporal_load(p + 2),
                 __builtin_nontemporal_load(p + 3)};
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

// sum of p[lo * stride], p[(lo + step) * stride], ... (indices < hi) in ascending order, fp64 -- the loads in batches of 8 in front of
// their adds (a plain `for (...) s += p[...]` with a run-time trip count is one dependent memory round trip per iteration: the small
// reduction kernels of a training step were 2 / 3 latency); the ORDER of the additions is the plain loop's, so are the bits
__device__ inline double strided_sum(const float* __restrict__ p, long long stride, int lo, int hi, int step) {
    double s = 0.0;
    int b = lo;
    for (; b + 7 * step < hi; b += 8 * step) {
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = p[(size_t)(b + k * step) * stride];
#pragma unroll
        for (int k = 0; k < 8; ++k) s += (double)v[k];
    }
    for (; b < hi; b += step) s += (double)p[(size_t)b * stride];
    return s;
}

// stage 2 of every column reduction: partial float [nblocks][n] -> totals double [n], fixed order (bitwise reproducible)
constexpr int RED_F32_THREADS = 1024;      // k_reduce_f32_partials, k_dweight_final: 32 columns x 32 slices of the partial list
__global__ void k_reduce_f32_partials(const float* __restrict__ partial, int nblocks, int n, double* __restrict__ totals);

// fixed-order sum of per-workgroup [128,128] float slabs -> dw (launch with 128 * 128 / 32 workgroups of RED_F32_THREADS = 1024 threads; train.hip)
// -- and, with `extra.partial` set and ceil(extra.n / 32) workgroups MORE, a k_reduce_f32_partials of the same producer's column
// partials in the same launch (every node of a captured batch-1 training step costs ~4.5 us whatever it does)
struct ExtraReduce {
    const float* partial;       // [nblocks][n]; NULL: none
    int nblocks, n;
    double* totals;             // [n]
};
__global__ void k_dweight_final(const float* __restrict__ partial, int nblocks, float* __restrict__ dw, const ExtraReduce extra);

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
// ... with the fixed-order reduction of the per-workgroup partials float [nblocks][2 * cc] in front of it, in ONE launch of
// cc / 16 workgroups (RED_F32_THREADS threads: 16 columns x {sum, sum of squares} x 32 slices of the partial list; the same
// summation order as k_reduce_f32_partials, so the same bits).  a.totals is not read.
__global__ void k_bn_reduce_finalize(const float* __restrict__ partial, int nblocks, const BnFinalize a);

}  // namespace eg
