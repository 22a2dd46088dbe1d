// Micro-benchmark (diagnostic): what bounds the row loads of one CU?
// One persistent workgroup per CU (256 of them) of W waves; every wave keeps K independent 1-KB wave loads (64 lanes x 16 B =
// two 512-B feature rows) in flight, consumes them with one add per register and goes on.  Sources:
//   stream  every workgroup walks its own slice of a 2.4-GB array once (HBM, no reuse)
//   mall    every workgroup re-reads a 1-MB slice (256 MB in all: memory-side cache, not L2)
//   l2      every workgroup re-reads a 64-KB slice (2 MB per XCD: L2-resident)
// Printed: bytes per clock per CU (s_memtime of the slowest workgroup) and the chip-wide rate, for W in {1,2,4,8} x K in
// {4,10,20,40}.  The producers of the shipped layer kernel are 4 waves x 20 loads = 80 KB in flight per CU.
//   hipcc --offload-arch=gfx950 -O3 -o tools/_bin/micro_loads tools/micro_loads.hip && tools/_bin/micro_loads
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int K>
__global__ __launch_bounds__(512) void k_loads(const float* __restrict__ x, size_t rows_per_wg, int passes, float* out,
                                               unsigned long long* cyc) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, waves = blockDim.x >> 6;
    const float* base = x + (size_t)blockIdx.x * rows_per_wg * 128;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const size_t chunks = rows_per_wg / 2;                    // 1-KB chunks (2 rows) of this workgroup's slice
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int p = 0; p < passes; ++p)
        for (size_t c0 = (size_t)wave * K; c0 + K <= chunks; c0 += (size_t)waves * K) {
            f32x4 v[K];
#pragma unroll
            for (int k = 0; k < K; ++k) v[k] = *reinterpret_cast<const f32x4*>(base + (c0 + k) * 256 + lane * 4);
#pragma unroll
            for (int k = 0; k < K; ++k) acc += v[k];
        }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int K>
void run(const float* x, size_t total_rows, int mode, int waves, float* out, unsigned long long* cyc, double clock_ratio) {
    const int grid = 256;
    // stream: the whole array once; mall: 1 MB per workgroup (256 MB in all: beyond the 8 x 4 MB of L2, inside the 256-MB
    // memory-side cache); l2: 64 KB per workgroup (2 MB per XCD: L2-resident)
    const size_t rows_per_wg = mode == 0 ? total_rows / grid : mode == 1 ? 2048 : 128;
    const int passes = mode == 0 ? 1 : mode == 1 ? 64 : 1024;
    const bool l2 = mode != 0;
    hipLaunchKernelGGL(k_loads<K>, dim3(grid), dim3(64 * waves), 0, 0, x, rows_per_wg, l2 ? 2 : 1, out, cyc);   // warm
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_loads<K>, dim3(grid), dim3(64 * waves), 0, 0, x, rows_per_wg, passes, out, cyc);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(grid);
    hipMemcpy(h.data(), cyc, grid * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    const unsigned long long worst = *std::max_element(h.begin(), h.end());
    const double bytes_wg = (double)(rows_per_wg / 2 / ((size_t)waves * K) * ((size_t)waves * K)) * 1024.0 * passes;
    // s_memtime counts at a fixed 100 MHz on this part; clock_ratio converts to shader clocks (measured by the caller)
    printf("  %-6s W=%d K=%2d (%3d KB in flight): %6.1f B/clk/CU  %6.2f TB/s\n", mode == 0 ? "stream" : mode == 1 ? "mall" : "l2", waves, K, waves * K,
           bytes_wg / ((double)worst * clock_ratio), bytes_wg * grid / (ms * 1e-3) / 1e12);
}

__global__ void k_clock(unsigned long long* o) {
    const unsigned long long m0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    float a = threadIdx.x;
    for (int i = 0; i < 4000000; ++i) a = a * 1.0000001f + 1e-9f;
    const unsigned long long m1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { o[0] = m1 - m0; o[1] = r1 - r0; o[2] = (unsigned long long)a; }
}

int main() {
    const size_t total_rows = (size_t)256 * 18432;                       // 2.4 GB of 512-B rows
    float *x, *out;
    unsigned long long* cyc;
    hipMalloc(&x, total_rows * 512);
    hipMalloc(&out, (size_t)256 * 512 * 4);
    hipMalloc(&cyc, 256 * 8 + 64);
    hipMemset(x, 0, total_rows * 512);
    // s_memtime unit vs the shader clock
    hipLaunchKernelGGL(k_clock, dim3(1), dim3(64), 0, 0, cyc);
    unsigned long long c[3];
    hipMemcpy(c, cyc, sizeof(c), hipMemcpyDeviceToHost);
    printf("s_memtime / s_memrealtime = %.3f (memrealtime is 100 MHz)\n", (double)c[0] / (double)c[1]);
    const double ratio = 1.0;   // B/clk below are per s_memtime tick; see the ratio above and the TB/s column for absolute numbers
    for (int mode = 0; mode < 3; ++mode)
        for (int waves : {1, 2, 4, 8}) {
            run<4>(x, total_rows, mode, waves, out, cyc, ratio);
            run<10>(x, total_rows, mode, waves, out, cyc, ratio);
            run<20>(x, total_rows, mode, waves, out, cyc, ratio);
            if (mode == 0) run<40>(x, total_rows, mode, waves, out, cyc, ratio);
        }
    return 0;
}
