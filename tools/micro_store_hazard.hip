// Micro-test (diagnostic): does a VALU write of the data registers of `buffer_store_dwordx4` right behind the store corrupt what
// is stored?  LLVM inserts a wait state for "VMEM store of more than 8 bytes -> VALU write of the store data" only when the
// store's scalar offset is NOT a register.  Here: every lane stores a 16-byte pattern and overwrites the four data registers
// with a poison value after GAP wait states (s_nop), with the scalar offset in an SGPR or as the literal 0; all 256 CUs store
// at once (8 waves each, many tiles) so that the memory pipeline is busy.  Counts the poisoned dwords that reached memory.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int GAP, bool SREG>
__global__ __launch_bounds__(512) void k(unsigned* out, int tiles) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(out, 0, 0x7fffff00, 0x00020000);
    const unsigned good = 0x600D0000u, bad = 0xBAD0BAD0u;
    for (int t = 0; t < tiles; ++t) {
        const unsigned row = (unsigned)((blockIdx.x * tiles + t) * 512 + threadIdx.x);     // 16 bytes per thread
        const unsigned voff = SREG ? (threadIdx.x & 63) * 16u : row * 16u;
        const unsigned soff = __builtin_amdgcn_readfirstlane(SREG ? (row - (threadIdx.x & 63)) * 16u : 0u);
        const unsigned g = good | (row & 0xFFFFu);
        if (SREG) {
            asm volatile(
                "v_mov_b32 v10, %0\n v_mov_b32 v11, %0\n v_mov_b32 v12, %0\n v_mov_b32 v13, %0\n s_nop 4\n"
                "buffer_store_dwordx4 v[10:13], %1, %2, %3 offen\n"
                ".if %5 > 0\n s_nop %5 - 1\n .endif\n"
                "v_mov_b32 v10, %4\n v_mov_b32 v11, %4\n v_mov_b32 v12, %4\n v_mov_b32 v13, %4\n"
                :: "v"(g), "v"(voff), "s"(r), "s"(soff), "v"(bad), "n"(GAP) : "v10", "v11", "v12", "v13", "memory");
        } else {
            asm volatile(
                "v_mov_b32 v10, %0\n v_mov_b32 v11, %0\n v_mov_b32 v12, %0\n v_mov_b32 v13, %0\n s_nop 4\n"
                "buffer_store_dwordx4 v[10:13], %1, %2, 0 offen\n"
                ".if %4 > 0\n s_nop %4 - 1\n .endif\n"
                "v_mov_b32 v10, %3\n v_mov_b32 v11, %3\n v_mov_b32 v12, %3\n v_mov_b32 v13, %3\n"
                :: "v"(g), "v"(voff), "s"(r), "v"(bad), "n"(GAP) : "v10", "v11", "v12", "v13", "memory");
        }
    }
}

// the same for a global (flat-address) store: `global_store_dwordx4 v[addr], v[10:13], off`
template <int GAP>
__global__ __launch_bounds__(512) void kg(unsigned* out, int tiles) {
    const unsigned good = 0x600D0000u, bad = 0xBAD0BAD0u;
    for (int t = 0; t < tiles; ++t) {
        const unsigned row = (unsigned)((blockIdx.x * tiles + t) * 512 + threadIdx.x);
        unsigned* p = out + (size_t)row * 4;
        const unsigned g = good | (row & 0xFFFFu);
        asm volatile(
            "v_mov_b32 v10, %0\n v_mov_b32 v11, %0\n v_mov_b32 v12, %0\n v_mov_b32 v13, %0\n s_nop 4\n"
            "global_store_dwordx4 %1, v[10:13], off\n"
            ".if %3 > 0\n s_nop %3 - 1\n .endif\n"
            "v_mov_b32 v10, %2\n v_mov_b32 v11, %2\n v_mov_b32 v12, %2\n v_mov_b32 v13, %2\n"
            :: "v"(g), "v"(p), "v"(bad), "n"(GAP) : "v10", "v11", "v12", "v13", "memory");
    }
}
template <int GAP>
void run_global() {
    const int tiles = 64, blocks = 256;
    const size_t n = (size_t)blocks * tiles * 512 * 4;
    unsigned* out;
    (void)hipMalloc(&out, n * 4);
    long bad = 0;
    for (int rep = 0; rep < 5; ++rep) {
        (void)hipMemset(out, 0, n * 4);
        hipLaunchKernelGGL((kg<GAP>), dim3(blocks), dim3(512), 0, 0, out, tiles);
        (void)hipDeviceSynchronize();
        std::vector<unsigned> h(n);
        (void)hipMemcpy(h.data(), out, n * 4, hipMemcpyDeviceToHost);
        for (size_t i = 0; i < n; ++i) bad += h[i] == 0xBAD0BAD0u;
    }
    printf("global_store_dwordx4      %d wait state(s) between the store and the overwrite: %ld of %zu dwords poisoned\n", GAP, bad, 5 * n);
    (void)hipFree(out);
}

// and for an LDS store: `ds_write_b128 v_addr, v[10:13]`, read back by the same thread after the write has completed
template <int GAP>
__global__ __launch_bounds__(512) void kd(unsigned* out, int tiles) {
    __shared__ unsigned lds[512 * 4];
    const unsigned good = 0x600D0000u, bad = 0xBAD0BAD0u;
    unsigned poisoned = 0;
    for (int t = 0; t < tiles; ++t) {
        const unsigned g = good | (unsigned)(t & 0xFFFF);
        const unsigned addr = threadIdx.x * 16;
        asm volatile(
            "v_mov_b32 v10, %0\n v_mov_b32 v11, %0\n v_mov_b32 v12, %0\n v_mov_b32 v13, %0\n s_nop 4\n"
            "ds_write_b128 %1, v[10:13]\n"
            ".if %3 > 0\n s_nop %3 - 1\n .endif\n"
            "v_mov_b32 v10, %2\n v_mov_b32 v11, %2\n v_mov_b32 v12, %2\n v_mov_b32 v13, %2\n"
            "s_waitcnt lgkmcnt(0)\n"
            :: "v"(g), "v"(addr), "v"(bad), "n"(GAP) : "v10", "v11", "v12", "v13", "memory");
        for (int i = 0; i < 4; ++i) poisoned += ((volatile unsigned*)lds)[threadIdx.x * 4 + i] == bad;
    }
    out[blockIdx.x * 512 + threadIdx.x] = poisoned;
}
template <int GAP>
void run_lds() {
    const int tiles = 2048, blocks = 256;
    unsigned* out;
    (void)hipMalloc(&out, blocks * 512 * 4);
    hipLaunchKernelGGL((kd<GAP>), dim3(blocks), dim3(512), 0, 0, out, tiles);
    (void)hipDeviceSynchronize();
    std::vector<unsigned> h(blocks * 512);
    (void)hipMemcpy(h.data(), out, blocks * 512 * 4, hipMemcpyDeviceToHost);
    long bad = 0;
    for (unsigned v : h) bad += v;
    printf("ds_write_b128             %d wait state(s) between the store and the overwrite: %ld of %ld dwords poisoned\n", GAP, bad,
           (long)blocks * 512 * 4 * tiles);
    (void)hipFree(out);
}

template <int GAP, bool SREG>
void run() {
    const int tiles = 64, blocks = 256;
    const size_t n = (size_t)blocks * tiles * 512 * 4;
    unsigned* out;
    (void)hipMalloc(&out, n * 4);
    long bad = 0;
    for (int rep = 0; rep < 5; ++rep) {
        (void)hipMemset(out, 0, n * 4);
        hipLaunchKernelGGL((k<GAP, SREG>), dim3(blocks), dim3(512), 0, 0, out, tiles);
        (void)hipDeviceSynchronize();
        std::vector<unsigned> h(n);
        (void)hipMemcpy(h.data(), out, n * 4, hipMemcpyDeviceToHost);
        for (size_t i = 0; i < n; ++i) bad += h[i] == 0xBAD0BAD0u;
    }
    printf("scalar offset %-8s  %d wait state(s) between the store and the overwrite: %ld of %zu dwords poisoned\n",
           SREG ? "SGPR" : "literal", GAP, bad, 5 * n);
    (void)hipFree(out);
}
int main() {
    run<0, true>(); run<1, true>(); run<2, true>(); run<3, true>(); run<4, true>();
    run<0, false>(); run<1, false>(); run<2, false>();
    run_global<0>(); run_global<1>(); run_global<2>(); run_global<3>();
    run_lds<0>(); run_lds<1>(); run_lds<2>();
    return 0;
}
