// Micro-benchmark (diagnostic): issue rate of the SIMD partner of an fp32 MFMA chain, by MFMA shape and by the partner's
// instruction class.  One 512-thread workgroup per CU: waves 0-3 (one per SIMD) run a dependent MFMA chain (32x32x2: 64
// cycles each; 16x16x4: 32 cycles each, same FLOP rate), waves 4-7 (their SIMD partners) run an independent stream of
// v_fma_f32 / ds_read_b128 / ds_write_b128 / global_load_dwordx4 (L1/L2 hits) / v_readlane_b32.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

enum { P_FMA = 0, P_DSREAD = 1, P_DSWRITE = 2, P_GLOAD = 3, P_READLANE = 4, P_SALU = 5, P_WAITCNT = 6, P_SNOP = 7, P_BRANCH = 8 };

template <int MF, int PK>
__global__ __launch_bounds__(512, 2) void k(float* out, const float* src, unsigned long long* cyc, int iters, int mode) {
    __shared__ float lds[512 * 4 + 64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    unsigned long long t0 = 0, t1 = 0;
    float s = 0.f;
    if (wave < 4) {
        f32x16 acc = {0}, accb = {0}, accc = {0}, accd = {0};
        f32x4 acc4[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
        float a = 1e-3f * lane, b = 2e-3f * lane;
        __syncthreads();
        t0 = __builtin_amdgcn_s_memtime();
        if (mode & 1)
            for (int it = 0; it < iters; ++it) {
                if (MF == 0) {
#pragma unroll
                    for (int e = 0; e < 64; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
                } else if (MF == 2) {
#pragma unroll
                    for (int e = 0; e < 32; ++e) {
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
                        accb = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, accb, 0, 0, 0);
                    }
                } else if (MF == 3) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
                        accb = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, accb, 0, 0, 0);
                        accc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, a, accc, 0, 0, 0);
                        accd = __builtin_amdgcn_mfma_f32_32x32x2f32(b, b, accd, 0, 0, 0);
                    }
                } else {            // two independent 16x16x4 chains interleaved (a dependent one alone would stall on its latency)
#pragma unroll
                    for (int e = 0; e < 64; ++e) {
                        acc4[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc4[0], 0, 0, 0);
                        acc4[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(b, a, acc4[1], 0, 0, 0);
                    }
                }
            }
        t1 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < 16; ++i) s += acc[i] + accb[i] + accc[i] + accd[i];
        s += acc4[0].x + acc4[1].y;
    } else {
        __builtin_amdgcn_s_setprio(3);
        float v[8];
        for (int i = 0; i < 8; ++i) v[i] = 1e-3f * (lane + i);
        f32x4 q[8];
        for (int i = 0; i < 8; ++i) q[i] = f32x4{v[i], v[i], v[i], v[i]};
        float* mine = lds + (wave - 4) * 64 * 4 + lane * 4;
        const float* gp = src + (size_t)blockIdx.x * 4096 + lane * 4;
        int r = 0;
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
                        else if (PK == P_DSWRITE) { *reinterpret_cast<volatile f32x4*>(mine) = q[i]; }
                        else if (PK == P_GLOAD) { q[i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(gp + 256 * i)); asm volatile("" :: "v"(q[i])); }
                        else if (PK == P_WAITCNT) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n s_waitcnt vmcnt(0) lgkmcnt(0)\n s_waitcnt vmcnt(0) lgkmcnt(0)\n s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); }
                        else if (PK == P_SNOP) { asm volatile("s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0" ::: "memory"); }
                        else if (PK == P_BRANCH) { asm volatile("s_cmp_eq_u32 %0, 12345\n s_cbranch_scc1 1f\n s_add_u32 %0, %0, 1\n1:\n s_sub_u32 %0, %0, 1" : "+s"(r) :: "scc"); }
                        else if (PK == P_SALU) { asm volatile("s_add_u32 %0, %0, 3\n s_add_u32 %0, %0, 5\n s_add_u32 %0, %0, 7\n s_add_u32 %0, %0, 9" : "+s"(r)); }
                        else { r += __builtin_amdgcn_readlane(lane + r, i); asm volatile("" : "+s"(r)); }
                    }
            }
        t1 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < 8; ++i) s += v[i] + q[i].x;
        s += r;
    }
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int MF, int PK>
void run(const char* name, int mode) {
    float *out, *src; unsigned long long* cyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8 * 8); hipMalloc(&src, 256 * 4096 * 4 + 65536);
    hipMemset(src, 0, 256 * 4096 * 4 + 65536);
    const int iters = 200;
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((k<MF, PK>), dim3(256), dim3(512), 0, 0, out, src, cyc, iters, mode);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(256 * 8);
    hipMemcpy(h.data(), cyc, 256 * 8 * 8, hipMemcpyDeviceToHost);
    double m = 0, v = 0;
    for (int b = 0; b < 256; ++b) for (int w = 0; w < 8; ++w) (w < 4 ? m : v) += (double)h[b * 8 + w];
    const double n_mfma = MF == 1 ? 128.0 : 64.0;
    printf("%-64s MFMA wave: %6.1f cyc / 4096 FLOP-per-lane-block (ideal 64)   partner: %7.2f cyc / instruction\n", name,
           (mode & 1) ? m / 1024 / iters / n_mfma * (MF == 1 ? 2.0 : 1.0) : 0.0, (mode & 2) ? v / 1024 / iters / 64.0 : 0.0);
    hipFree(out); hipFree(cyc); hipFree(src);
}
#define ROW(MF, PK, label) run<MF, PK>(label " alone", 2); run<MF, PK>("32x32x2 chain + " label, 3);
int main() {
    run<0, P_SALU>("4 x s_add_u32 alone (per group of 4)", 2);
    run<0, P_SALU>("one chain + 4 x s_add_u32 (per group of 4)", 3);
    run<0, P_READLANE>("v_readlane alone", 2);
    run<0, P_READLANE>("one chain + v_readlane", 3);
    run<0, P_WAITCNT>("4 x s_waitcnt (satisfied) alone (per group of 4)", 2);
    run<0, P_WAITCNT>("one chain + 4 x s_waitcnt (per group of 4)", 3);
    run<0, P_SNOP>("4 x s_nop 0 alone (per group of 4)", 2);
    run<0, P_SNOP>("one chain + 4 x s_nop 0 (per group of 4)", 3);
    run<0, P_BRANCH>("s_cmp + s_cbranch (not taken) + 2 SALU alone (per group of 4)", 2);
    run<0, P_BRANCH>("one chain + s_cmp + s_cbranch + 2 SALU (per group of 4)", 3);
    run<0, P_FMA>("v_fma_f32 alone", 2);
    run<0, P_FMA>("one chain + v_fma_f32", 3);
    return 0;
}
