// Micro-benchmark (diagnostic): does v_mfma_f32_32x32x2_f32 (fp32 in, 64 FLOP/clk/SIMD = the fp32 VALU rate) leave the SIMD's
// vector ALU to the OTHER wave of that SIMD?  One 512-thread workgroup per CU: waves 0-3 (one per SIMD) run a dependent MFMA
// chain, waves 4-7 (their SIMD partners) run independent v_fma_f32 / v_pk_fma_f32 streams.  Cycles per instruction of each
// role alone and together.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// mode bit 0: MFMA waves work; bit 1: VALU waves work; PK: packed fma; PRIO: s_setprio 3 on the VALU waves
template <bool PK, bool PRIO>
__global__ __launch_bounds__(512, 2) void k(float* out, unsigned long long* cyc, int iters, int mode) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    unsigned long long t0 = 0, t1 = 0;
    float s = 0.f;
    if (wave < 4) {
        f32x16 acc = {0};
        float a = 1e-3f * lane, b = 2e-3f * lane;
        __syncthreads();
        t0 = __builtin_amdgcn_s_memtime();
        if (mode & 1)
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int e = 0; e < 64; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
            }
        t1 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < 16; ++i) s += acc[i];
    } else {
        if (PRIO) __builtin_amdgcn_s_setprio(3);
        f32x2 v[8];
        for (int i = 0; i < 8; ++i) v[i] = f32x2{1e-3f * (lane + i), 2e-3f * (lane - i)};
        const f32x2 m = {1.0001f, 0.9999f}, c = {1e-6f, -1e-6f};
        __syncthreads();
        t0 = __builtin_amdgcn_s_memtime();
        if (mode & 2)
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int r = 0; r < 8; ++r)
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        if (PK) v[i] = v[i] * m + c;                     // v_pk_fma_f32
                        else { v[i].x = __builtin_fmaf(v[i].x, m.x, c.x); v[i].y = __builtin_fmaf(v[i].y, m.y, c.y); }
                    }
            }
        t1 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < 8; ++i) s += v[i].x + v[i].y;
    }
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <bool PK, bool PRIO>
void run(const char* name, int mode) {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8 * 8);
    const int iters = 400;
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((k<PK, PRIO>), dim3(256), dim3(512), 0, 0, out, cyc, iters, mode);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(256 * 8);
    hipMemcpy(h.data(), cyc, 256 * 8 * 8, hipMemcpyDeviceToHost);
    double m = 0, v = 0;
    for (int b = 0; b < 256; ++b) for (int w = 0; w < 8; ++w) (w < 4 ? m : v) += (double)h[b * 8 + w];
    const double n_valu = PK ? 64.0 : 128.0;            // vector instructions per iteration of a VALU wave
    printf("%-58s MFMA wave: %6.1f cyc / MFMA (ideal 64)   VALU wave: %6.2f cyc / instruction\n", name,
           (mode & 1) ? m / 1024 / iters / 64 : 0.0, (mode & 2) ? v / 1024 / iters / n_valu : 0.0);
    hipFree(out); hipFree(cyc);
}
int main() {
    run<false, false>("MFMA waves alone", 1);
    run<false, false>("v_fma_f32 waves alone", 2);
    run<true, false>("v_pk_fma_f32 waves alone", 2);
    run<false, false>("MFMA + v_fma_f32 partners", 3);
    run<false, true>("MFMA + v_fma_f32 partners (partners at s_setprio 3)", 3);
    run<true, false>("MFMA + v_pk_fma_f32 partners", 3);
    run<true, true>("MFMA + v_pk_fma_f32 partners (partners at s_setprio 3)", 3);
    return 0;
}
