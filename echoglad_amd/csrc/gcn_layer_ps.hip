// Producer / consumer form of the fused GCN layer for the implicit topology (gfx950).
//
// One workgroup of 8 waves per CU (256 VGPRs per wave, 134 KB LDS):
//   waves 4..7  PRODUCERS  aggregate tile k+1 (two 8-node segments each, all row loads of both segments in flight
//               at once) into LDS buffer (k+1)&1: the aggregated tile [64][128] and the raw self rows (residual);
//   waves 0..3  CONSUMERS  one per SIMD: output channels 32c..32c+31, W slice in 64 VGPRs, 128 chained
//               v_mfma_f32_32x32x2_f32 per tile on buffer k&1, then the epilogue straight from the accumulators
//               (scale/shift, ReLU, residual from the LDS stash, 16-B stores).
// ONE workgroup barrier per tile hands buffer (k+1)&1 to the consumers and buffer k&1 back to the producers, so
// memory traffic of tile k+1 always overlaps the matrix work of tile k instead of relying on two workgroups
// drifting apart.  Tile ids come from the per-XCD queues (tile.h), claimed two tiles ahead by one producer lane.
#include <stdlib.h>

#include "seg_wide.h"

namespace eg {

constexpr int PS_THREADS = 512;

struct PsDims {
    int n_per_frame, batch, tiles_per_frame, relu, transpose_w, has_res;
};

#ifdef EG_STAMP
#define PSTAMP(i) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long _t = __builtin_amdgcn_s_memtime(); \
                       __builtin_amdgcn_s_waitcnt(0xC07F); st[i] += _t - t_prev; t_prev = _t; __builtin_amdgcn_sched_barrier(0); } while (0)
#define PSTAMP_INIT unsigned long long st[8] = {0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long t_prev = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F)
#define PSTAMP_FLUSH(base) do { if (lane_k == 0) { unsigned long long* stats = reinterpret_cast<unsigned long long*>(counters + WALK_GROUPS * WALK_CTR_STRIDE); \
                                for (int i = 0; i < 4; ++i) atomicAdd(&stats[(base) + i], st[i]); if ((base) == 0) atomicAdd(&stats[8], 1ull); } } while (0)
#else
#define PSTAMP(i) do {} while (0)
#define PSTAMP_INIT do {} while (0)
#define PSTAMP_FLUSH(base) do {} while (0)
#endif

// Asynchronous claim: the atomic on the own XCD's queue is issued early (ps_claim_issue), its result is looked at
// only after the tile's row loads have come back (ps_claim_commit); only an exhausted queue takes the slow walk.
__device__ inline int ps_claim_issue(int* __restrict__ counters, int group) {
    return atomicAdd(&counters[group * WALK_CTR_STRIDE], 1);
}
__device__ inline void ps_claim(int* __restrict__ counters, int group, int n_tiles, int* slot);
__device__ inline void ps_claim_commit(int* __restrict__ counters, int group, int n_tiles, int got, int* slot) {
    const int chunk = (n_tiles + WALK_GROUPS - 1) / WALK_GROUPS;
    const int lo = group * chunk;
    const int size = (n_tiles - lo) < chunk ? (n_tiles - lo) : chunk;
    if (got < size) *slot = lo + got;
    else ps_claim(counters, (group + 1) % WALK_GROUPS, n_tiles, slot);
}

__device__ inline void ps_claim(int* __restrict__ counters, int group, int n_tiles, int* slot) {
    const int chunk = (n_tiles + WALK_GROUPS - 1) / WALK_GROUPS;
    int t = -1;
    for (int k = 0; k < WALK_GROUPS && t < 0; ++k) {
        const int q = (group + k) % WALK_GROUPS;
        const int lo = q * chunk;
        const int size = (n_tiles - lo) < chunk ? (n_tiles - lo) : chunk;
        if (size <= 0) continue;
        const int got = atomicAdd(&counters[q * WALK_CTR_STRIDE], 1);
        if (got < size) t = lo + got;
    }
    *slot = t;
}

__global__ __launch_bounds__(PS_THREADS, 2) void k_gcn_layer_ps(const float* __restrict__ x, const float* __restrict__ W,
                                                                const float* __restrict__ scale, const float* __restrict__ shift,
                                                                float* __restrict__ out, const float* __restrict__ dis,
                                                                const Topo* __restrict__ T, const TileDesc* __restrict__ tiles,
                                                                const SegDesc* __restrict__ segs, const float* __restrict__ pats,
                                                                int* __restrict__ counters, const PsDims a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* s_a0 = smem;                               // [2][TILE * LDA]  aggregated tiles
    float* s_x0 = smem + 2 * TILE * LDA;              // [2][TILE * LDA]  raw self rows (residual); then the output tile
    int* s_tile = reinterpret_cast<int*>(smem + 4 * TILE * LDA);                  // [4] ring of tile ids

    const int tid = threadIdx.x;
    const int lane_k = tid & 63;
    const int wave = wave_id();
    const bool consumer = wave < 4;
    const int n_tiles = a.tiles_per_frame * a.batch;
    const int group = xcc_id();

    // ---- prologue: tiles 0 and 1 claimed (before anyone reads the ring) ----------------------------------------
    if (tid == 256) { ps_claim(counters, group, n_tiles, &s_tile[0]); ps_claim(counters, group, n_tiles, &s_tile[1]); }
    __syncthreads();

    // The two roles run separate loops (so that neither carries the other's persistent registers); both execute
    // exactly one workgroup barrier per tile, in lock step:   [prologue barrier]  (tile k work)  [barrier k] ...
    if (wave < 4) {
        // =========================== CONSUMER: channels 32*wave .. 32*wave+31 ===================================
        float wreg[64];
        load_w_slice(W, wave, lane_k, a.transpose_w, wreg);
        f32x4 sc[4], sh[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int ch0 = 32 * wave + 8 * g + 4 * (lane_k >> 5);
            sc[g] = scale ? *reinterpret_cast<const f32x4*>(scale + ch0) : f32x4{1.f, 1.f, 1.f, 1.f};
            sh[g] = shift ? *reinterpret_cast<const f32x4*>(shift + ch0) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        __syncthreads();                                   // tile 0 is in buffer 0
        PSTAMP_INIT;
        for (int k = 0;; ++k) {
            const int t_cur = __builtin_amdgcn_readfirstlane(s_tile[k & 3]);
            if (t_cur < 0) break;
            int lane = lane_k;
            asm volatile("" : "+v"(lane));
            PSTAMP(3);
            const float* s_a = s_a0 + (k & 1) * TILE * LDA;
            float* s_x = s_x0 + (k & 1) * TILE * LDA;
            const int frame = t_cur / a.tiles_per_frame;
            const int t_in = t_cur - frame * a.tiles_per_frame;
            int seg_first[8], seg_cnt[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {                          // scalar loads, back long before the epilogue
                seg_first[i] = segs[t_in * 8 + i].n_first;
                seg_cnt[i] = segs[t_in * 8 + i].cnt;
            }
            f32x16 acc0 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            f32x16 acc1 = acc0;
#ifndef EG_ABL_NO_MFMA
            mfma_rowblock(s_a, 0, lane, wreg, acc0);
            if (seg_cnt[4] > 0) mfma_rowblock(s_a, 32, lane, wreg, acc1);
#else
            acc0[0] = wreg[0] + s_a[lane]; acc1[0] = wreg[63];
#endif
            PSTAMP(0);
            // Epilogue.  Lane (row j of the 32-row block, half h) holds 16 channels of ONE row: stored from there a
            // wave instruction would touch 32 rows x 32 B.  The finished values go back into this wave's own channel
            // slice of the stash instead (where the residual was read from; no other wave touches that slice), are
            // re-read 8 lanes per row, and leave as whole 128-B line segments: 8 stores per wave and tile.
            const int j = lane & 31, h = lane >> 5;
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) {
                float* xp = s_x + (32 * rb + j) * LDA + 32 * wave + 4 * h;     // LDS row = 8 * patch row + column
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 v;
                    if (rb == 0) v = f32x4{acc0[4 * g], acc0[4 * g + 1], acc0[4 * g + 2], acc0[4 * g + 3]};
                    else v = f32x4{acc1[4 * g], acc1[4 * g + 1], acc1[4 * g + 2], acc1[4 * g + 3]};
                    v = v * sc[g] + sh[g];
                    if (a.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                    if (a.has_res) v += *reinterpret_cast<const f32x4*>(xp + 8 * g);
                    *reinterpret_cast<f32x4*>(xp + 8 * g) = v;
                }
            }
            {
                const int u = lane >> 3, c4 = 4 * (lane & 7);
                float* ob = out + (size_t)frame * a.n_per_frame * C + 32 * wave + c4;
                f32x4 o[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) o[i] = *reinterpret_cast<const f32x4*>(s_x + (8 * i + u) * LDA + 32 * wave + c4);
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    if (u < seg_cnt[i]) *reinterpret_cast<f32x4*>(ob + (size_t)(seg_first[i] + u) * C) = o[i];
            }
            PSTAMP(1);
            __syncthreads();                               // barrier k+1: buffer (k+1)&1 is full, buffer k&1 is free
            PSTAMP(2);
        }
        PSTAMP_FLUSH(0);
    } else {
        // =========================== PRODUCER: patch rows 2p, 2p+1 of every tile ================================
        const int p = wave - 4;
#ifndef EG_PS_NO_PRIO
        __builtin_amdgcn_s_setprio(3);
#endif
        PSTAMP_INIT;
        auto produce = [&](int tile_i, int buf, int lane) {
            const int frame = tile_i / a.tiles_per_frame;
            const int t_in = tile_i - frame * a.tiles_per_frame;
            const float* __restrict__ xf = x + (size_t)frame * a.n_per_frame * C;
            const SegDesc sd0 = segs[t_in * 8 + 2 * p];
            const SegDesc sd1 = segs[t_in * 8 + 2 * p + 1];
            float* s_a = s_a0 + buf * TILE * LDA;
            float* s_x = a.has_res ? s_x0 + buf * TILE * LDA : nullptr;
            f32x4 acc0[4], acc1[4];
            if (sd0.pad0) {
                // ---- usual case: both segments on the fast path, vertically adjacent, same parents ----
                SegPair A;
                segp_issue(sd0, sd1, pats, xf, lane, A);
#ifdef EG_ABL_NO_LOADS        // timing-only ablation: producers do nothing (results wrong)
                return;
#endif
                __builtin_amdgcn_sched_barrier(0);                  // every load of both segments is issued above this line
                PSTAMP(0);
                segw_rows(lane, A.wa_a, A.Sa, A.La, A.Ra, A.U, A.Sb, A.P, acc0, s_x, 16 * p);
                segw_rows(lane, A.wa_b, A.Sb, A.Lb, A.Rb, A.Sa, A.D, A.P, acc1, s_x, 16 * p + 8);
                pin_acc4(acc0);
                pin_acc4(acc1);                                     // the 44 main-stage registers are dead from here on
                __builtin_amdgcn_sched_barrier(0);
                PSTAMP(1);
                if (sd0.aux) {                                      // uniform: aux level, children exist as slots
                    SegKids K0, K1;
                    segw_kids_issue(sd0, xf, lane, K0);
                    segw_kids_issue(sd1, xf, lane, K1);
                    segw_kids_add(lane, A.wa_a, A.wb_a, K0, acc0);
                    segw_kids_add(lane, A.wa_b, A.wb_b, K1, acc1);
                }
                segw_store(lane, A.wa_a, acc0, s_a, 16 * p);
                segw_store(lane, A.wa_b, acc1, s_a, 16 * p + 8);
            } else {
                // ---- ragged patches, coordinate nodes, frame end: one segment after the other ----
#pragma unroll 1
                for (int e = 0; e < 2; ++e) {
                    const SegDesc& sd = e ? sd1 : sd0;
                    const int rl = 16 * p + 8 * e;
                    SegW A;
                    segw_issue(sd, pats, xf, lane, A);
                    segw_main(T, sd, xf, dis, lane, A, acc0, s_a, s_x, rl);
                    if (A.mode == 1) {
                        if (sd.aux) {
                            SegKids K;
                            segw_kids_issue(sd, xf, lane, K);
                            segw_kids_add(lane, A.wa, A.wb, K, acc0);
                        }
                        segw_store(lane, A.wa, acc0, s_a, rl);
                    }
                }
            }
            PSTAMP(2);
        };
        {
            const int t0 = __builtin_amdgcn_readfirstlane(s_tile[0]);
            if (t0 >= 0) produce(t0, 0, lane_k);
        }
        __syncthreads();                                   // tile 0 is in buffer 0
        for (int k = 0;; ++k) {
            const int t_cur = __builtin_amdgcn_readfirstlane(s_tile[k & 3]);
            if (t_cur < 0) break;
            const int t_next = __builtin_amdgcn_readfirstlane(s_tile[(k + 1) & 3]);
            int lane = lane_k;
            asm volatile("" : "+v"(lane));
            int got = 0;
            if (tid == 256) got = ps_claim_issue(counters, group);                          // two tiles ahead, asynchronous
            PSTAMP(3);
            if (t_next >= 0) produce(t_next, (k + 1) & 1, lane);
            if (tid == 256) ps_claim_commit(counters, group, n_tiles, got, &s_tile[(k + 2) & 3]);
            __syncthreads();                               // barrier k+1
            PSTAMP(3);
        }
        PSTAMP_FLUSH(4);
    }
}

static int env_int_ps(const char* name, int dflt) {
    const char* v = getenv(name);
    return v && *v ? atoi(v) : dflt;
}

}  // namespace eg

using namespace eg;

// Used by eg_gcn_layer_fwd (gcn_layer.hip) for topology handles when the residual is NULL or x itself.
// Returns EG_ERR_UNSUPPORTED when the caller should fall back to the symmetric kernel.
int eg_launch_layer_ps(const eg_graph* g, int batch, const float* x, const float* W, const float* scale,
                       const float* shift, const float* residual, int relu, int transpose_w, float* out,
                       hipStream_t stream) {
    if (!g || g->kind != GRAPH_TOPO || (residual != nullptr && residual != x)) return EG_ERR_UNSUPPORTED;
    if (env_int_ps("EG_LAYER_IMPL", 0) == 0) return EG_ERR_UNSUPPORTED;
    PsDims a{};
    a.n_per_frame = (int)g->n_nodes; a.batch = batch; a.tiles_per_frame = g->n_tiles;
    a.relu = relu; a.transpose_w = transpose_w; a.has_res = residual != nullptr;
    const long long n_tiles = (long long)a.tiles_per_frame * batch;
    if (n_tiles <= 0) return EG_OK;
    const size_t lds = (size_t)(4 * TILE * LDA + 4) * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        EG_HIP_TRY(hipFuncSetAttribute((const void*)k_gcn_layer_ps, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    EG_HIP_TRY(hipMemsetAsync(g->walk_counters, 0, sizeof(int) * WALK_GROUPS * WALK_CTR_STRIDE, stream));
    long long grid = n_tiles < 256 ? n_tiles : env_int_ps("EG_PS_GRID", 256);      // one persistent workgroup per CU
    hipLaunchKernelGGL(k_gcn_layer_ps, dim3((unsigned)grid), dim3(PS_THREADS), lds, stream, x, W, scale, shift, out, g->dis,
                       g->topo_dev, g->tiles_dev, g->segs_dev, g->pats_dev, g->walk_counters, a);
    EG_HIP_TRY(hipGetLastError());
    return EG_OK;
}
