// Micro-benchmark (diagnostic): a stalled MFMA holds the SIMD's issue stage -- the partner wave gets one instruction per
// MFMA of a dense chain (micro_coissue2).  Does idling the MFMA wave between its MFMAs (s_nop: the wave is not a
// candidate for issue) hand the issue stage to the partner without losing MFMA rate?
// One 512-thread workgroup per CU: waves 0-3 run a dependent v_mfma_f32_32x32x2_f32 chain with NOPS x `s_nop 7` (8 cycles
// each) after every MFMA; waves 4-7 (their SIMD partners) run v_fma_f32 / ds_read_b128 / global_load_dwordx4 streams.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
enum { P_FMA = 0, P_DSREAD = 1, P_GLOAD = 3, P_MIX = 6 };

template <int N> __device__ inline void nops() {
    if constexpr (N > 0) { asm volatile("s_nop 7"); nops<N - 1>(); }
}

template <int NOPS, int PK>
__global__ __launch_bounds__(512, 2) void k(float* out, const float* src, unsigned long long* cyc, int iters, int mode) {
    __shared__ float lds[512 * 4 + 64];
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
                for (int e = 0; e < 64; ++e) {
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    nops<NOPS>();
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        t1 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < 16; ++i) s += acc[i];
    } else {
        __builtin_amdgcn_s_setprio(3);
        float v[8];
        for (int i = 0; i < 8; ++i) v[i] = 1e-3f * (lane + i);
        f32x4 q[8];
        for (int i = 0; i < 8; ++i) q[i] = f32x4{v[i], v[i], v[i], v[i]};
        float* mine = lds + (wave - 4) * 64 * 4 + lane * 4;
        const float* gp = src + (size_t)blockIdx.x * 4096 + lane * 4;
        __syncthreads();
        t0 = __builtin_amdgcn_s_memtime();
        if (mode & 2)
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int rr = 0; rr < 8; ++rr)
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        if (PK == P_FMA) v[i] = __builtin_fmaf(v[i], 1.0001f, 1e-6f);
                        else if (PK == P_DSREAD) { q[i] = *reinterpret_cast<volatile f32x4*>(mine); }
                        else if (PK == P_GLOAD) { q[i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(gp + 256 * i)); asm volatile("" :: "v"(q[i])); }
                        else {      // a producer-like mix: 6 FMAs, 1 LDS read, 1 global load per 8
                            if (i < 6) v[i] = __builtin_fmaf(v[i], 1.0001f, 1e-6f);
                            else if (i == 6) q[i] = *reinterpret_cast<volatile f32x4*>(mine);
                            else { q[i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(gp + 256 * rr)); asm volatile("" :: "v"(q[i])); }
                        }
                    }
            }
        t1 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < 8; ++i) s += v[i] + q[i].x;
    }
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int NOPS, int PK>
void run(const char* name, int mode) {
    float *out, *src; unsigned long long* cyc;
    (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&cyc, 256 * 8 * 8); (void)hipMalloc(&src, 256 * 4096 * 4 + 65536);
    (void)hipMemset(src, 0, 256 * 4096 * 4 + 65536);
    const int iters = 200;
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((k<NOPS, PK>), dim3(256), dim3(512), 0, 0, out, src, cyc, iters, mode);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(256 * 8);
    (void)hipMemcpy(h.data(), cyc, 256 * 8 * 8, hipMemcpyDeviceToHost);
    double m = 0, v = 0;
    for (int b = 0; b < 256; ++b) for (int w = 0; w < 8; ++w) (w < 4 ? m : v) += (double)h[b * 8 + w];
    printf("%-44s nops %d x 8   MFMA wave: %6.1f cyc / MFMA (ideal 64)   partner: %7.2f cyc / instruction\n", name, NOPS,
           (mode & 1) ? m / 1024 / iters / 64.0 : 0.0, (mode & 2) ? v / 1024 / iters / 64.0 : 0.0);
    (void)hipFree(out); (void)hipFree(cyc); (void)hipFree(src);
}
template <int PK> void sweep(const char* name) {
    run<0, PK>(name, 2);
    run<0, PK>(name, 3); run<2, PK>(name, 3); run<4, PK>(name, 3); run<5, PK>(name, 3); run<6, PK>(name, 3); run<7, PK>(name, 3); run<8, PK>(name, 3);
}
int main() {
    run<0, P_FMA>("chain alone", 1); run<4, P_FMA>("chain alone", 1); run<6, P_FMA>("chain alone", 1); run<7, P_FMA>("chain alone", 1); run<8, P_FMA>("chain alone", 1);
    sweep<P_FMA>("v_fma_f32 partner");
    sweep<P_DSREAD>("ds_read_b128 partner");
    sweep<P_GLOAD>("global_load_dwordx4 partner");
    sweep<P_MIX>("mixed partner (6 fma, 1 ds, 1 load)");
    return 0;
}
