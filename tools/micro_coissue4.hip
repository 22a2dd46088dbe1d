// Micro-benchmark (diagnostic, round 5): does the SIMD partner's issue slot follow MFMA *boundaries* or MFMA *time*?
// One 512-thread workgroup per CU, waves 0-3 (one per SIMD) run fp32 MFMAs of one shape at the same FLOP rate
//   32x32x2  (16 passes, 64 cycles, 4096 FLOP)   - the shipped consumer chain
//   16x16x4  ( 8 passes, 32 cycles, 2048 FLOP)   - 4 independent accumulators, round robin
//   4x4x1    ( 2 passes,  8 cycles,  512 FLOP)   - 8 independent accumulators, round robin
// waves 4-7 (their SIMD partners, s_setprio 3) run an independent stream of one instruction class.  Reported: cycles per
// 4096 FLOP of the MFMA wave (64 = the fp32 roof) and the partner's instructions per 64 cycles (= per 32x32x2 MFMA time)
// alone and beside the chain.  DESIGN.md 5.22 (one partner instruction per MFMA) was measured on 32x32x2 only.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

enum { P_FMA = 0, P_DSREAD = 1, P_DSWRITE = 2, P_GLOAD = 3, P_READLANE = 4, P_SALU = 5, P_PKFMA = 6 };
enum { M_32 = 0, M_16 = 1, M_4 = 2, M_32SELF = 3, M_16SELF = 4 };

template <int MF, int PK>
__global__ __launch_bounds__(512, 2) void k(float* out, const float* src, unsigned long long* cyc, int iters, int mode) {
    __shared__ float lds[512 * 4 + 64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    unsigned long long t0 = 0, t1 = 0;
    float s = 0.f;
    if (wave < 4) {
        f32x16 acc = {0};
        f32x4 a16[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
        f32x4 a4[8];
        for (int i = 0; i < 8; ++i) a4[i] = f32x4{0, 0, 0, 0};
        float a = 1e-3f * lane, b = 2e-3f * lane;
        float own[4] = {1.f, 2.f, 3.f, 4.f};
        __syncthreads();
        t0 = __builtin_amdgcn_s_memtime();
        if (mode & 1)
            for (int it = 0; it < iters; ++it) {              // one iteration = 64 x 4096 FLOP per lane block
                if (MF == M_32) {
#pragma unroll
                    for (int e = 0; e < 64; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
                } else if (MF == M_32SELF) {                  // the MFMA wave's OWN vector instruction between MFMAs
#pragma unroll
                    for (int e = 0; e < 64; ++e) {
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
                        own[e & 3] = __builtin_fmaf(own[e & 3], 1.0001f, 1e-6f);
                        asm volatile("" : "+v"(own[e & 3]), "+v"(acc));
                    }
                } else if (MF == M_16) {
#pragma unroll
                    for (int e = 0; e < 32; ++e)
#pragma unroll
                        for (int c = 0; c < 4; ++c) a16[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, a16[c], 0, 0, 0);
                } else if (MF == M_16SELF) {
#pragma unroll
                    for (int e = 0; e < 32; ++e)
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            a16[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, a16[c], 0, 0, 0);
                            own[c] = __builtin_fmaf(own[c], 1.0001f, 1e-6f);
                            asm volatile("" : "+v"(own[c]), "+v"(a16[c]));
                        }
                } else {
#pragma unroll
                    for (int e = 0; e < 64; ++e)
#pragma unroll
                        for (int c = 0; c < 8; ++c) a4[c] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, a4[c], 0, 0, 0);
                }
            }
        t1 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < 16; ++i) s += acc[i];
        for (int c = 0; c < 4; ++c) s += a16[c].x + own[c];
        for (int c = 0; c < 8; ++c) s += a4[c].y;
    } else {
        __builtin_amdgcn_s_setprio(3);
        float v[8];
        for (int i = 0; i < 8; ++i) v[i] = 1e-3f * (lane + i);
        f32x4 q[8];
        for (int i = 0; i < 8; ++i) q[i] = f32x4{v[i], v[i], v[i], v[i]};
        float* mine = lds + (wave - 4) * 64 * 4 + lane * 4;
        const float* gp = src + (size_t)blockIdx.x * 4096 + lane * 4;
        int r = 0;
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        f32x2 pk[8], pkc = {1.0001f, 1.0002f};
        for (int i = 0; i < 8; ++i) pk[i] = f32x2{v[i], v[i]};
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
                        else if (PK == P_DSWRITE) { *reinterpret_cast<volatile f32x4*>(mine) = q[i]; }
                        else if (PK == P_GLOAD) { q[i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(gp + 256 * i)); asm volatile("" :: "v"(q[i])); }
                        else if (PK == P_SALU) { asm volatile("s_add_u32 %0, %0, 3\n s_add_u32 %0, %0, 5\n s_add_u32 %0, %0, 7\n s_add_u32 %0, %0, 9" : "+s"(r)); }
                        else { r += __builtin_amdgcn_readlane(lane + r, i); asm volatile("" : "+s"(r)); }
                    }
            }
        t1 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < 8; ++i) s += v[i] + q[i].x + q[i].z + pk[i].x + pk[i].y;
        s += r;
    }
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int MF, int PK>
void run(const char* shape, const char* partner) {
    float *out, *src; unsigned long long* cyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8 * 8); hipMalloc(&src, 256 * 4096 * 4 + 65536);
    hipMemset(src, 0, 256 * 4096 * 4 + 65536);
    const int iters = 200;
    double res[2][2];
    for (int mode = 2; mode <= 3; ++mode) {
        for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((k<MF, PK>), dim3(256), dim3(512), 0, 0, out, src, cyc, iters, mode);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(256 * 8);
        hipMemcpy(h.data(), cyc, 256 * 8 * 8, hipMemcpyDeviceToHost);
        double m = 0, v = 0;
        for (int b = 0; b < 256; ++b) for (int w = 0; w < 8; ++w) (w < 4 ? m : v) += (double)h[b * 8 + w];
        res[mode - 2][0] = m / 1024 / iters / 64.0;        // MFMA wave: cycles per 4096 FLOP
        res[mode - 2][1] = v / 1024 / iters / 64.0;        // partner: cycles per instruction (per group of 4 for SALU)
    }
    // s_memtime ticks at 100 MHz on this part?  No: it returns the shader clock counter here (2.38 GHz in-kernel);
    // everything below is in those cycles.
    const double per64_alone = 64.0 / res[0][1], per64_with = 64.0 / res[1][1];
    printf("%-10s + %-22s | MFMA wave %6.1f cyc / 4096 FLOP | partner %7.2f cyc/instr alone, %7.2f beside | per 64 cyc: %6.2f alone, %5.2f beside | per MFMA-wave-time-unit (4096 FLOP): %5.2f\n",
           shape, partner, res[1][0], res[0][1], res[1][1], per64_alone, per64_with, res[1][0] / res[1][1]);
    hipFree(out); hipFree(cyc); hipFree(src);
}

template <int MF>
void shape_rows(const char* shape) {
    run<MF, P_FMA>(shape, "v_fma_f32");
    run<MF, P_PKFMA>(shape, "v_pk_fma_f32");
    run<MF, P_DSREAD>(shape, "ds_read_b128");
    run<MF, P_DSWRITE>(shape, "ds_write_b128");
    run<MF, P_GLOAD>(shape, "global_load_dwordx4");
    run<MF, P_READLANE>(shape, "v_readlane_b32");
    run<MF, P_SALU>(shape, "4 x s_add_u32");
}

int main() {
    shape_rows<M_32>("32x32x2");
    shape_rows<M_16>("16x16x4");
    shape_rows<M_4>("4x4x1");
    // the MFMA wave's own VALU instruction between its MFMAs (no partner stream: mode 1 only matters, but run() does 2 and 3)
    run<M_32SELF, P_FMA>("32x32x2+own", "v_fma_f32");
    run<M_16SELF, P_FMA>("16x16x4+own", "v_fma_f32");
    return 0;
}
