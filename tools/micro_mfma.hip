// Micro-benchmark (diagnostic): cycles per v_mfma_f32_32x32x2_f32 for one wave per SIMD when the 128 MFMAs of a tile are
//   mode 0: two dependent chains of 64, one after the other (the shipped consumer)
//   mode 1: the same two chains interleaved instruction by instruction
// with the B operand (a) in registers, (b) read from LDS like tile.h's mfma_rowblock.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE, bool LDS>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* cyc, int iters) {
    __shared__ __attribute__((aligned(16))) float s_a[64 * 132];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 64 * 132; i += 256) s_a[i] = (float)((i * 2654435761u) >> 20) * 1e-4f;
    __syncthreads();
    float w[64];
    for (int i = 0; i < 64; ++i) w[i] = 1e-3f * (float)((lane * 64 + i) % 97);
    f32x16 acc0 = {0}, acc1 = {0};
    const f32x4* ap0 = reinterpret_cast<const f32x4*>(s_a + (lane & 31) * 132 + 64 * (lane >> 5));
    const f32x4* ap1 = reinterpret_cast<const f32x4*>(s_a + (32 + (lane & 31)) * 132 + 64 * (lane >> 5));
    float breg[64];
    for (int i = 0; i < 64; ++i) breg[i] = s_a[(lane * 7 + i) % (64 * 132)];
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                f32x4 b = LDS ? ap0[t] : f32x4{breg[4 * t], breg[4 * t + 1], breg[4 * t + 2], breg[4 * t + 3]};
#pragma unroll
                for (int e = 0; e < 4; ++e) acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(w[4 * t + e], b[e], acc0, 0, 0, 0);
            }
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                f32x4 b = LDS ? ap1[t] : f32x4{breg[4 * t], breg[4 * t + 1], breg[4 * t + 2], breg[4 * t + 3]};
#pragma unroll
                for (int e = 0; e < 4; ++e) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(w[4 * t + e], b[e], acc1, 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                f32x4 b0 = LDS ? ap0[t] : f32x4{breg[4 * t], breg[4 * t + 1], breg[4 * t + 2], breg[4 * t + 3]};
                f32x4 b1 = LDS ? ap1[t] : f32x4{breg[4 * t + 1], breg[4 * t], breg[4 * t + 3], breg[4 * t + 2]};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(w[4 * t + e], b0[e], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(w[4 * t + e], b1[e], acc1, 0, 0, 0);
                }
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE, bool LDS>
void run(const char* name) {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 8);
    const int iters = 200;
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((k<MODE, LDS>), dim3(256), dim3(256), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(256);
    hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
    double s = 0; for (auto v : h) s += (double)v;
    printf("%-44s %7.1f cycles per MFMA (ideal 64)\n", name, s / 256 / iters / 128);
    hipFree(out); hipFree(cyc);
}
int main() {
    run<0, false>("two chains in sequence, B in registers");
    run<1, false>("two chains interleaved, B in registers");
    run<0, true>("two chains in sequence, B from LDS");
    run<1, true>("two chains interleaved, B from LDS");
    return 0;
}
