// The optimizer update of the training step as ONE launch (gfx950).
// The reference trains with torch.optim.Adam (src/engine.py:60-72 via its optimizer builder).  On the GPU torch's fused form is a
// `_foreach_add_` on the step counts plus one multi-tensor launch per ~30 parameter tensors (its pointer table travels in the kernel
// arguments: 4 KB): 4 launches for the 82 tensors of the GNN stack, landmark MLPs and heads -- 34 us of a 0.9-ms captured batch-1 step,
// for 300 KB of parameters.  Here: one launch.  Up to ADAM_MAX_TENSORS (p, grad, exp_avg, exp_avg_sq, numel) entries travel in the
// kernel arguments (3.5 KB), workgroup b finds its (tensor, chunk) in one parallel step over the chunk counts, and the step counts
// -- steps[k] of one device array, one per tensor as in torch's state (a parameter without a gradient in some step falls behind the
// others) -- are advanced by the workgroup that leaves last (every workgroup has read its own by then).
// Arithmetic: torch's fused Adam, ADAM_MODE::ORIGINAL (aten/src/ATen/native/cuda/fused_adam_utils.cuh), amsgrad off:
//     g = grad (+ weight_decay * p);  m = m + (1 - b1) (g - m);  v = b2 v + (1 - b2) g g
//     p -= (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)          t = step + 1, bias corrections in fp64 like there
#include "train_common.h"

namespace eg {

constexpr int ADAM_MAX_TENSORS = 96;
constexpr int ADAM_THREADS = 256;
constexpr int ADAM_PER = 4;                           // elements per thread: all their loads are issued before the first use
constexpr int ADAM_CHUNK = ADAM_THREADS * ADAM_PER;   // elements per workgroup

struct AdamTable {
    float* p[ADAM_MAX_TENSORS];
    const float* g[ADAM_MAX_TENSORS];
    float* m[ADAM_MAX_TENSORS];
    float* v[ADAM_MAX_TENSORS];
    int n[ADAM_MAX_TENSORS];
    float* steps;                       // [count] device floats: the number of updates of each tensor so far
    int count;
};
static_assert(sizeof(AdamTable) <= 3584, "the table travels in the kernel arguments: 4 KB with the scalars and the ~256 hidden bytes the runtime appends");
static_assert(ADAM_MAX_TENSORS <= ADAM_THREADS, "one thread per tensor in the look-up");

struct AdamScalars {
    float lr, beta1, beta2, eps, weight_decay;
    int maximize;
    const float* lr_dev;            // nullable: the learning rate as a device scalar (a scheduler's changes reach a captured launch)
    unsigned* ticket;
};

__global__ __launch_bounds__(ADAM_THREADS) void k_adam_step(const AdamTable tb, const AdamScalars a) {
    // (tensor, chunk) of this workgroup: thread k sums the chunk counts in front of tensor k and looks whether this workgroup falls into
    // its range -- one parallel step over LDS (a scalar walk over the table is a chain of dependent constant-memory loads: 15 us)
    __shared__ int s_chunks[ADAM_MAX_TENSORS], s_ti, s_rest, s_last;
    const int t = threadIdx.x;
    if (t < tb.count) s_chunks[t] = (tb.n[t] + ADAM_CHUNK - 1) / ADAM_CHUNK;
    __syncthreads();
    if (t < tb.count) {
        int first = 0;
        for (int k = 0; k < t; ++k) first += s_chunks[k];
        if ((int)blockIdx.x >= first && (int)blockIdx.x < first + s_chunks[t]) { s_ti = t; s_rest = blockIdx.x - first; }
    }
    __syncthreads();
    const int ti = s_ti, rest = s_rest;
    const float t_now = tb.steps[ti] + 1.0f;
    const double bc1 = 1.0 - pow((double)a.beta1, (double)t_now), bc2 = 1.0 - pow((double)a.beta2, (double)t_now);
    const float lr = a.lr_dev ? *a.lr_dev : a.lr;
    const float step_size = (float)((double)lr / bc1), bc2_sqrt = (float)sqrt(bc2);
    // (a uniform run-time index into the kernel-argument segment: scalar loads)
    float* __restrict__ p = tb.p[ti]; const float* __restrict__ g = tb.g[ti]; float* __restrict__ m = tb.m[ti]; float* __restrict__ v = tb.v[ti];
    const int n = tb.n[ti];
    const int lo = rest * ADAM_CHUNK, hi = lo + ADAM_CHUNK < n ? lo + ADAM_CHUNK : n;
    const float w1 = 1.0f - a.beta1, w2 = 1.0f - a.beta2;
    float pv[ADAM_PER], gv[ADAM_PER], mv[ADAM_PER], vv[ADAM_PER];
#pragma unroll
    for (int k = 0; k < ADAM_PER; ++k) {
        const int i = lo + t + k * ADAM_THREADS, j = i < hi ? i : lo;
        pv[k] = p[j]; gv[k] = g[j]; mv[k] = m[j]; vv[k] = v[j];
    }
#pragma unroll
    for (int k = 0; k < ADAM_PER; ++k) {
        const int i = lo + t + k * ADAM_THREADS;
        if (i >= hi) continue;
        const float pi = pv[k];
        float gi = a.maximize ? -gv[k] : gv[k];
        if (a.weight_decay != 0.f) gi += pi * a.weight_decay;
        float mi = mv[k], vi = vv[k];
        mi = mi + w1 * (gi - mi);
        vi = a.beta2 * vi + w2 * gi * gi;
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + a.eps;
        p[i] = pi - step_size * mi / denom;
    }
    // the last workgroup out advances the step counts (no data is handed over, so no fence: every workgroup has read its count above)
    __syncthreads();
    if (t == 0) s_last = atomicInc(a.ticket, gridDim.x - 1) == gridDim.x - 1;
    __syncthreads();
    if (s_last && t < tb.count) tb.steps[t] += 1.0f;
}

}  // namespace eg

using namespace eg;

extern "C" {

int eg_adam_step(const eg_adam_tensor* tensors, int count, float* steps, float lr, const float* lr_device, float beta1, float beta2, float eps,
                 float weight_decay, int maximize, eg_stream_t stream) {
    if (!tensors || !steps || count < 0) return set_error(EG_ERR_ARG, "NULL argument");
    if (!(beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f) || !(eps >= 0.f)) return set_error(EG_ERR_ARG, "betas must be in [0, 1), eps >= 0");
    if (count > ADAM_MAX_TENSORS) return set_error(EG_ERR_UNSUPPORTED, "more than 96 tensors per call: split the list");
    if (count == 0) return EG_OK;
    AdamTable tb{};
    long long blocks = 0;
    for (int k = 0; k < count; ++k) {
        const eg_adam_tensor& t = tensors[k];
        if (!t.param || !t.grad || !t.exp_avg || !t.exp_avg_sq || t.numel < 1 || t.numel >= (1ll << 31)) return set_error(EG_ERR_ARG, "bad tensor entry");
        tb.p[k] = t.param; tb.g[k] = t.grad; tb.m[k] = t.exp_avg; tb.v[k] = t.exp_avg_sq; tb.n[k] = (int)t.numel;
        blocks += (t.numel + ADAM_CHUNK - 1) / ADAM_CHUNK;
    }
    tb.count = count;
    tb.steps = steps;
    if (blocks >= (1ll << 31)) return set_error(EG_ERR_ARG, "too many elements");
    unsigned* ticket = eg_ticket_ptr((void*)stream, 2);
    if (!ticket) return set_error(EG_ERR_HIP, "no device memory for a ticket word");
    const AdamScalars a{lr, beta1, beta2, eps, weight_decay, maximize ? 1 : 0, lr_device, ticket};
    hipLaunchKernelGGL(k_adam_step, dim3((unsigned)blocks), dim3(ADAM_THREADS), 0, (hipStream_t)stream, tb, a);
    EG_HIP_TRY(hipGetLastError());
    return EG_OK;
}

}  // extern "C"
