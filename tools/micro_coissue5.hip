// Micro-benchmark (diagnostic, round 5): what hands the SIMD partner its issue slots beside an fp32 MFMA chain?
// micro_coissue4 found (a) the +64-cycle penalty per partner instruction is the same beside 32x32x2, 16x16x4 and 4x4x1 chains
// (slots follow time, not MFMA boundaries) and (b) a chain whose wave issues ONE vector instruction of its own between two MFMAs
// runs 13.6 cycles slower per MFMA while its partner issues ~9 instructions per MFMA instead of ~1.  This file maps (b):
// the MFMA wave's own instructions between consecutive dependent 32x32x2 MFMAs are exact inline-asm sequences
//   OWN_NONE            mfma
//   OWN_VFMA  x K       mfma ; K x v_fma_f32 (registers disjoint from the accumulators)
//   OWN_VNOP  x K       mfma ; K x v_nop
//   OWN_SNOP  x K       mfma ; K x s_nop 0
//   OWN_DSREAD x K      mfma ; K x ds_read_b128 (no wait)
//   OWN_VMOV  x K       mfma ; K x v_mov_b32
//   OWN_SLEEP           mfma ; s_sleep 1
//   OWN_SETPRIO         mfma ; s_setprio 0
// partner (s_setprio 3): v_fma_f32 | ds_read_b128 | global_load_dwordx4 (L2 hits) | v_pk_fma_f32 streams.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

enum { OWN_NONE = 0, OWN_VFMA, OWN_VNOP, OWN_SNOP, OWN_DSREAD, OWN_VMOV, OWN_SLEEP, OWN_SETPRIO };
enum { P_FMA = 0, P_DSREAD = 1, P_GLOAD = 2, P_PKFMA = 3 };

#define MF "v_mfma_f32_32x32x2_f32 %0, %1, %2, %0\n"

template <int OWN, int K>
__device__ __forceinline__ void mfma_step(f32x16& acc, float a, float b, float& o0, float& o1, f32x4& q, unsigned lp) {
    if constexpr (OWN == OWN_NONE) asm volatile(MF : "+v"(acc) : "v"(a), "v"(b));
    else if constexpr (OWN == OWN_VFMA) {
        if constexpr (K == 1) asm volatile(MF "v_fma_f32 %3, %3, %1, %2\n" : "+v"(acc), "+v"(a), "+v"(b), "+v"(o0));
        else if constexpr (K == 2) asm volatile(MF "v_fma_f32 %3, %3, %1, %2\n v_fma_f32 %4, %4, %1, %2\n" : "+v"(acc), "+v"(a), "+v"(b), "+v"(o0), "+v"(o1));
        else asm volatile(MF "v_fma_f32 %3, %3, %1, %2\n v_fma_f32 %4, %4, %1, %2\n v_fma_f32 %3, %3, %1, %2\n v_fma_f32 %4, %4, %1, %2\n" : "+v"(acc), "+v"(a), "+v"(b), "+v"(o0), "+v"(o1));
    } else if constexpr (OWN == OWN_VNOP) {
        if constexpr (K == 1) asm volatile(MF "v_nop\n" : "+v"(acc) : "v"(a), "v"(b));
        else if constexpr (K == 2) asm volatile(MF "v_nop\n v_nop\n" : "+v"(acc) : "v"(a), "v"(b));
        else asm volatile(MF "v_nop\n v_nop\n v_nop\n v_nop\n" : "+v"(acc) : "v"(a), "v"(b));
    } else if constexpr (OWN == OWN_SNOP) {
        if constexpr (K == 1) asm volatile(MF "s_nop 0\n" : "+v"(acc) : "v"(a), "v"(b));
        else asm volatile(MF "s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n" : "+v"(acc) : "v"(a), "v"(b));
    } else if constexpr (OWN == OWN_DSREAD) {
        if constexpr (K == 1) asm volatile(MF "ds_read_b128 %3, %4\n" : "+v"(acc), "+v"(a), "+v"(b), "=v"(q) : "v"(lp));
        else asm volatile(MF "ds_read_b128 %3, %4\n ds_read_b128 %3, %4 offset:16\n" : "+v"(acc), "+v"(a), "+v"(b), "=v"(q) : "v"(lp));
    } else if constexpr (OWN == OWN_VMOV) {
        asm volatile(MF "v_mov_b32 %3, %1\n" : "+v"(acc), "+v"(a), "+v"(b), "+v"(o0));
    } else if constexpr (OWN == OWN_SLEEP) {
        asm volatile(MF "s_sleep 1\n" : "+v"(acc) : "v"(a), "v"(b));
    } else {
        asm volatile(MF "s_setprio 0\n" : "+v"(acc) : "v"(a), "v"(b));
    }
}

template <int OWN, int K, int PK>
__global__ __launch_bounds__(512, 2) void k(float* out, const float* src, unsigned long long* cyc, int iters, int mode) {
    __shared__ float lds[512 * 4 + 64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    unsigned long long t0 = 0, t1 = 0;
    float s = 0.f;
    if (wave < 4) {
        f32x16 acc = {0};
        float a = 1e-3f * lane, b = 2e-3f * lane, o0 = 1.f, o1 = 2.f;
        f32x4 q = {0, 0, 0, 0};
        const unsigned lp = (unsigned)(size_t)(lds + wave * 64 * 4 + lane * 4);
        __syncthreads();
        t0 = __builtin_amdgcn_s_memtime();
        if (mode & 1)
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int e = 0; e < 64; ++e) mfma_step<OWN, K>(acc, a, b, o0, o1, q, lp);
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        t1 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < 16; ++i) s += acc[i];
        s += o0 + o1 + q.x;
    } else {
        __builtin_amdgcn_s_setprio(3);
        float v[8];
        for (int i = 0; i < 8; ++i) v[i] = 1e-3f * (lane + i);
        f32x4 q[8];
        for (int i = 0; i < 8; ++i) q[i] = f32x4{v[i], v[i], v[i], v[i]};
        f32x2 pk[8], pkc = {1.0001f, 1.0002f};
        for (int i = 0; i < 8; ++i) pk[i] = f32x2{v[i], v[i]};
        float* mine = lds + wave * 64 * 4 + lane * 4;
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
                        else if (PK == P_PKFMA) { asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(pk[i]) : "v"(pkc)); }
                        else if (PK == P_DSREAD) { q[i] = *reinterpret_cast<volatile f32x4*>(mine); }
                        else { q[i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(gp + 256 * i)); asm volatile("" :: "v"(q[i])); }
                    }
            }
        t1 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < 8; ++i) s += v[i] + q[i].x + pk[i].x;
    }
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int OWN, int K, int PK>
void run(const char* own, const char* partner) {
    float *out, *src; unsigned long long* cyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8 * 8); hipMalloc(&src, 256 * 4096 * 4 + 65536);
    hipMemset(src, 0, 256 * 4096 * 4 + 65536);
    const int iters = 200;
    double res[3][2];
    for (int mode = 1; mode <= 3; ++mode) {
        for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((k<OWN, K, PK>), dim3(256), dim3(512), 0, 0, out, src, cyc, iters, mode);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(256 * 8);
        hipMemcpy(h.data(), cyc, 256 * 8 * 8, hipMemcpyDeviceToHost);
        double m = 0, v = 0;
        for (int b = 0; b < 256; ++b) for (int w = 0; w < 8; ++w) (w < 4 ? m : v) += (double)h[b * 8 + w];
        res[mode - 1][0] = m / 1024 / iters / 64.0;        // MFMA wave: cycles per MFMA (+ its own instructions)
        res[mode - 1][1] = v / 1024 / iters / 64.0;        // partner: cycles per instruction
    }
    printf("own %-16s x%d  partner %-20s | MFMA period alone %6.1f, with partner %6.1f | partner cyc/instr alone %7.2f, beside %7.2f | partner instr per MFMA period %6.2f\n",
           own, K, partner, res[0][0], res[2][0], res[1][1], res[2][1], res[2][0] / res[2][1]);
    hipFree(out); hipFree(cyc); hipFree(src);
}

template <int OWN, int K>
void rows(const char* own) {
    run<OWN, K, P_FMA>(own, "v_fma_f32");
    run<OWN, K, P_PKFMA>(own, "v_pk_fma_f32");
    run<OWN, K, P_DSREAD>(own, "ds_read_b128");
    run<OWN, K, P_GLOAD>(own, "global_load_dwordx4");
}

int main() {
    rows<OWN_NONE, 0>("none");
    rows<OWN_VFMA, 1>("v_fma_f32");
    rows<OWN_VFMA, 2>("v_fma_f32");
    rows<OWN_VFMA, 4>("v_fma_f32");
    rows<OWN_VNOP, 1>("v_nop");
    rows<OWN_VNOP, 2>("v_nop");
    rows<OWN_VNOP, 4>("v_nop");
    rows<OWN_SNOP, 1>("s_nop 0");
    rows<OWN_SNOP, 4>("s_nop 0");
    rows<OWN_DSREAD, 1>("ds_read_b128");
    rows<OWN_DSREAD, 2>("ds_read_b128");
    rows<OWN_VMOV, 1>("v_mov_b32");
    rows<OWN_SLEEP, 1>("s_sleep 1");
    rows<OWN_SETPRIO, 1>("s_setprio 0");
    return 0;
}
