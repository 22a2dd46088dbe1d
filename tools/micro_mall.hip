// Micro-benchmark (diagnostic): does the tail of a large array a kernel has just written (or read) ascending stay in the
// memory-side cache (256 MB on MI355X), so that a consumer walking the array DESCENDING starts on cache hits?
// A 1.18 GB array (one [B*N,128] fp32 tensor of the batch-32 training step): producer = streaming write or read, ascending;
// consumer = streaming read, ascending or descending; the consumer's time and the time of its first / last quarter.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void k_write(f32x4* p, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) p[i] = f32x4{1.f, 2.f, 3.f, 4.f};
}
template <bool DESC>
__global__ void k_read(const f32x4* p, size_t lo, size_t hi, float* out) {
    f32x4 s = {0, 0, 0, 0};
    const size_t n = hi - lo, stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const size_t j = DESC ? hi - 1 - i : lo + i;
        const f32x4 v = __builtin_nontemporal_load(p + j);
        s += v;
    }
    if (s.x + s.y + s.z + s.w == 12345.f) out[0] = 1.f;
}

static float timed(void (*launch)(void*), void* ctx) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a); launch(ctx); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms;
}
struct Ctx { f32x4* p; size_t lo, hi; float* out; };
int main() {
    const size_t bytes = (size_t)32 * 72024 * 512, n4 = bytes / 16;
    f32x4 *p, *other; float* out;
    hipMalloc(&p, bytes); hipMalloc(&other, bytes); hipMalloc(&out, 4);
    hipMemset(other, 0, bytes);
    const int grid = 256 * 16, blk = 256;
    for (int producer = 0; producer < 2; ++producer)
        for (int desc = 0; desc < 2; ++desc)
            for (int part = 0; part < 3; ++part) {            // 0: whole array, 1: the quarter the consumer touches first, 2: the last quarter
                float best = 1e9f;
                for (int rep = 0; rep < 5; ++rep) {
                    hipLaunchKernelGGL(k_read<false>, dim3(grid), dim3(blk), 0, 0, other, (size_t)0, n4, out);    // evict
                    if (producer == 0) hipLaunchKernelGGL(k_write, dim3(grid), dim3(blk), 0, 0, p, n4);
                    else hipLaunchKernelGGL(k_read<false>, dim3(grid), dim3(blk), 0, 0, p, (size_t)0, n4, out);
                    hipDeviceSynchronize();
                    size_t lo = 0, hi = n4;
                    if (part == 1) { if (desc) lo = n4 - n4 / 8; else hi = n4 / 8; }
                    if (part == 2) { if (desc) hi = n4 / 8; else lo = n4 - n4 / 8; }
                    Ctx c{p, lo, hi, out};
                    const float ms = timed(desc ? [](void* v) { Ctx* c = (Ctx*)v; hipLaunchKernelGGL(k_read<true>, dim3(256 * 16), dim3(256), 0, 0, c->p, c->lo, c->hi, c->out); }
                                                : [](void* v) { Ctx* c = (Ctx*)v; hipLaunchKernelGGL(k_read<false>, dim3(256 * 16), dim3(256), 0, 0, c->p, c->lo, c->hi, c->out); }, &c);
                    if (ms < best) best = ms;
                }
                const double gb = (part == 0 ? bytes : bytes / 8) / 1e9;
                printf("producer %-5s ascending, consumer %-10s %-28s %7.3f ms  %6.2f TB/s\n", producer ? "read" : "write", desc ? "descending" : "ascending",
                       part == 0 ? "whole array (1.18 GB)" : part == 1 ? "the 1/8 it touches first" : "the 1/8 it touches last", best, gb / best);
            }
    return 0;
}
