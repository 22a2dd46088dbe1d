// Producer / consumer form of the fused GCN layer for the implicit topology (gfx950).
//
// One workgroup of 8 waves per CU (256 VGPRs per wave, 134 KB LDS):
//   waves 4..7  PRODUCERS  aggregate tile k+1 (two 8-node segments each, all row loads of both segments in flight
//               at once) into LDS buffer (k+1)&1: the aggregated tile [64][128] and the raw self rows (residual);
//   waves 0..3  CONSUMERS  one per SIMD: output channels 32c..32c+31, W slice in 64 VGPRs, 128 chained
//               v_mfma_f32_32x32x2_f32 per tile on buffer k&1 in the rows x channels orientation (tile.h, SWAP): an
//               accumulator register is 2 x 128 contiguous bytes of two output rows, so the epilogue (scale/shift, ReLU,
//               residual, child sums) runs on the accumulators and stores them as they stand -- no LDS round trip, no
//               consumer-side synchronisation; it is issued between the MFMAs of the NEXT 32-row block (of the next tile,
//               for the second block: its residual rows are read before the barrier, the rest lives in registers).
// ONE workgroup barrier per tile hands buffer (k+1)&1 to the consumers and buffer k&1 back to the producers, so
// memory traffic of tile k+1 always overlaps the matrix work of tile k instead of relying on two workgroups
// drifting apart.  Tile ids come from the per-XCD queues (tile.h), claimed two tiles ahead by one producer lane.
#include <stdlib.h>

#include "seg_wide.h"

namespace eg {

constexpr int PS_THREADS = 512;
constexpr int PS_LDS_DIS = 4 * TILE * LDA + 16 + 64;  // float offsets inside the dynamic LDS block (tile-id ring, CLS counter, descriptor ring)
constexpr int PS_LDS_PAT = PS_LDS_DIS + PS_DIS_RING * TILE;

struct PsDims {
    int n_per_frame, batch, tiles_per_frame, relu, transpose_w, has_res;
    int kid_rows;      // rows per frame of the child-sum side buffers (kin / kout)
    int n_pats;
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

// Classifier heads fused behind the LAST layer of a stack (CLS = true): node-type filter + 4 x [Linear(128,32)-BN-ReLU-
// Linear(32,16)-BN-ReLU-Linear(16,1)] of src/core/models.py:363-377, :485-490, eval-mode BN folded by the caller
// (same packing as eg_classifier_fwd).  The layer's output tile never leaves LDS.

// KOUT: the launch writes the child sums of its output (kout != NULL), a launch-uniform property compiled in so that the
// epilogue stays branch-free inside the MFMA chains.
// KIN: the launch reads the child sums of its input (kin != NULL), likewise compiled in: a launch without them pulls the
// children of aux nodes as rows, which takes the registers the child-sum rows of the next tile would travel in.
template <bool CLS, bool KOUT, bool KIN>
__global__ __launch_bounds__(PS_THREADS, 2) void k_gcn_layer_ps(const float* __restrict__ x, const float* __restrict__ W,
                                                                const float* __restrict__ scale, const float* __restrict__ shift,
                                                                float* __restrict__ out, const float* __restrict__ dis,
                                                                const Topo* __restrict__ T, const TileDesc* __restrict__ tiles,
                                                                const SegDesc* __restrict__ segs, const float* __restrict__ pats,
                                                                const float* __restrict__ patsq, const float* __restrict__ kin, float* __restrict__ kout,
                                                                float* __restrict__ sink_base, int* __restrict__ counters, const PsDims a,
                                                                const ClsArgs ca) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* s_a0 = smem;                               // [2][TILE * LDA]  aggregated tiles
    float* s_x0 = smem + 2 * TILE * LDA;              // [2][TILE * LDA]  raw self rows (residual)
    int* s_tile = reinterpret_cast<int*>(smem + 4 * TILE * LDA);                  // [8] ring of tile ids
    int* s_cd = s_tile + 16;                          // [2][8 segments][4] descriptor words of the consumers' epilogue (below)
    float* s_dis0 = smem + PS_LDS_DIS;                // [PS_DIS_RING][TILE] (deg+1)^-1/2 of the tile's rows (child sums of the output)
    float* s_pat = smem + PS_LDS_PAT;                 // [n_pats][64] weight patterns, quad layout (seg_wide.h)
    float* s_bn = s_pat + a.n_pats * PATQ;            // CLS only: [2][128] first classifier layers: scale s1, folded shift t1c

    const int tid = threadIdx.x;
    const int lane_k = tid & 63;
    const int wave = wave_id();
    const int n_tiles = a.tiles_per_frame * a.batch;
    const int group = xcc_id();

    // ---- prologue: tiles 0 .. 3 claimed (before anyone reads the ring) -----------------------------------------
    if (tid == 256) {
        ps_claim(counters, group, n_tiles, &s_tile[0]);
        ps_claim(counters, group, n_tiles, &s_tile[1]);
        ps_claim(counters, group, n_tiles, &s_tile[2]);
        ps_claim(counters, group, n_tiles, &s_tile[3]);
    }
    for (int i = tid; i < a.n_pats * PATQ; i += PS_THREADS) s_pat[i] = patsq[i];
    if (CLS) {
        if (tid < C) {
            s_bn[tid] = ca.s1[tid];
            s_bn[C + tid] = ca.t1c[tid];
        }
        if (tid < 64) {                                    // second / third classifier layers: [s2 | t2 | w3] x [4 heads x 16]
            s_bn[2 * C + tid] = ca.s2[tid];
            s_bn[2 * C + 64 + tid] = ca.t2[tid];
            s_bn[2 * C + 128 + tid] = ca.w3[tid];
        }
    }
    __syncthreads();

    // The two roles run separate loops (so that neither carries the other's persistent registers); both execute
    // exactly one workgroup barrier per tile, in lock step:   [prologue barrier]  (tile k work)  [barrier k] ...
    if (wave < 4) {
        // Descriptor words the epilogue needs -- {n_first, cnt, par0, pad1} of the tile's 8 segments: first node and count,
        // parent row and parent count of the segment pair -- come through an LDS ring the producers fill from the
        // descriptors they hold anyway (slot = buffer parity).  The consumers issue NO vector-memory load: a load whose
        // result is needed at the top of the next tile made the compiler drain vmcnt there, i.e. wait for every output
        // store of the tile before.
        if constexpr (!CLS) {
        // =========================== CONSUMER: channels 32*wave .. 32*wave+31 ===================================
        float wreg[64];
        load_w_slice(W, wave, lane_k, a.transpose_w, wreg);
        // lane (j, h): channel 32 wave + j; accumulator register e of the 32-row block rb is tile row
        // 32 rb + (e & 3) + 8 (e >> 2) + 4 h  =  patch row 4 rb + (e >> 2), column (e & 3) + 4 h
        const int ch = 32 * wave + (lane_k & 31);
        const float scj = scale ? scale[ch] : 1.0f;
        const float shj = shift ? shift[ch] : 0.0f;
        const float relu_floor = a.relu ? 0.f : -__builtin_inff();
        const bool has_res = a.has_res != 0;
        __syncthreads();                                   // tile 0 is in buffer 0
        PSTAMP_INIT;
#ifdef EG_STAMP
        // in-kernel clock (MI355X_MICROARCH.md, DVFS give-back item 6): shader cycles / 100 MHz ticks around the tile loop
        const unsigned long long clk0 = __builtin_amdgcn_s_memtime(), rt0 = __builtin_amdgcn_s_memrealtime();
#endif
        // The second 32-row block of a tile is finished one tile late (inside the first MFMA chain of the next tile, after
        // the barrier that hands its LDS buffers back to the producers): everything it needs is in registers by then.
        // Both chains of a tile carry an epilogue UNCONDITIONALLY (a second instance of a chain for "nothing pending" costs
        // an accumulator copy behind every chain): when there is nothing to finish -- first tile of the workgroup, or a
        // ragged tile, whose two blocks take the conditional-store path below -- the row pointers aim at this workgroup's
        // slice of a dump area (eg_graph::sink) and the values stored are never read.
        float* const sink = sink_base + (size_t)blockIdx.x * PS_SINK_FLOATS;
        float* p_orow[4] = {sink, sink, sink, sink};       // out + (frame rows + first node of patch rows 4..7) * C
        float* p_krow[2] = {sink, sink};                   // kout + (frame kid rows + first parent row of patch-row pairs 2, 3) * C
        int p_slot = 0;                                    // slice of the s_dis ring that holds the pending tile's (deg+1)^-1/2
        f32x16 acc0, acc1;
        float res0[16], res1[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) { acc0[e] = 0.f; acc1[e] = 0.f; res0[e] = 0.f; res1[e] = 0.f; }
        const int hq = lane_k >> 5;
        const unsigned lane_off = (unsigned)(4 * hq * C + ch);                        // + first node * C + (e & 3) * C
        const unsigned kid_off = (unsigned)(2 * hq * C + ch);                         // + first parent row * C + (0 | 1) * C

        // one patch row (accumulator registers 4c .. 4c+3) of a block, straight from the accumulators; d = (deg+1)^-1/2 of
        // the lane's 4 nodes of that patch row
        float ks[2] = {0.f, 0.f};
        auto epi_row = [&](const f32x16& acc, const float (&res)[16], int c, float* orow, float* krow, const f32x4& d) {
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float t = fmaf(acc[4 * c + e], scj, shj);
                t = fmaxf(t, relu_floor);
                t += has_res ? res[4 * c + e] : 0.f;
                v[e] = t;
                orow[lane_off + e * C] = t;
            }
            if (KOUT) {
                // children (2 pr, 2 pc), (2 pr, 2 pc + 1), (2 pr + 1, 2 pc), (2 pr + 1, 2 pc + 1) of parent (pr, pc = 2 h + e2)
                // are registers 2 e2, 2 e2 + 1 of this patch row and of the next one, in THIS lane; summed in that order
                if ((c & 1) == 0) {
                    ks[0] = fmaf(d[1], v[1], d[0] * v[0]);
                    ks[1] = fmaf(d[3], v[3], d[2] * v[2]);
                } else {
                    ks[0] = fmaf(d[1], v[1], fmaf(d[0], v[0], ks[0]));
                    ks[1] = fmaf(d[3], v[3], fmaf(d[2], v[2], ks[1]));
                    krow[kid_off] = ks[0];
                    krow[kid_off + C] = ks[1];
                }
            }
        };

        for (int k = 0;; ++k) {
            const int t_cur = __builtin_amdgcn_readfirstlane(s_tile[k & 7]);
            if (t_cur < 0) break;
            int lane = lane_k;
            asm volatile("" : "+v"(lane));
            const int cd = s_cd[(k & 1) * 32 + (lane & 31)];
            const float* s_a = s_a0 + (k & 1) * TILE * LDA;
            const float* s_x = s_x0 + (k & 1) * TILE * LDA;
            const int frame = t_cur / a.tiles_per_frame;
            const int h = lane >> 5;
            int seg_first[8], seg_cnt[8], par0[4], npar[4];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                seg_first[i] = __builtin_amdgcn_readlane(cd, 4 * i);
                seg_cnt[i] = __builtin_amdgcn_readlane(cd, 4 * i + 1);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                par0[i] = __builtin_amdgcn_readlane(cd, 8 * i + 2);
                npar[i] = __builtin_amdgcn_readlane(cd, 8 * i + 3);
            }
            // fast tile: a full 8x8 patch (and, when child sums are written, every 2x2 block of it has its parent)
            bool fast = true;
#pragma unroll
            for (int i = 0; i < 8; ++i) fast = fast && seg_cnt[i] == 8;
            if (KOUT) {
#pragma unroll
                for (int i = 0; i < 4; ++i) fast = fast && npar[i] == 4;
            }
            float* const ofr = out + (size_t)frame * a.n_per_frame * C;
            float* const kfr = KOUT ? kout + (size_t)frame * a.kid_rows * C : sink;
            const float* const resp = s_x + (4 * h) * LDA + ch;               // + tile row * LDA
            const float* const disp = s_dis0 + 4 * h;                         // + ring slot * TILE + 8 * patch row
            auto read_res = [&](float (&res)[16], int rb, int c) {
#pragma unroll
                for (int e = 0; e < 4; ++e) res[4 * c + e] = resp[(32 * rb + 8 * c + e) * LDA];
            };
            auto read_dis = [&](int slot, int prow) -> f32x4 {
                return KOUT ? *reinterpret_cast<const f32x4*>(disp + slot * TILE + 8 * prow) : f32x4{0.f, 0.f, 0.f, 0.f};
            };
            const int slot = k & (PS_DIS_RING - 1);

            // ---- rows 0..31: MFMA chain; in its issue gaps the residual rows of this block are read and the deferred
            // second block of the previous tile is finished (every LDS value is read one chunk ahead of its use)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc0[e] = 0.f;
            f32x4 dq[4];
            dq[0] = read_dis(p_slot, 4);
            mfma_rowblock_with<true>(s_a, 0, lane, wreg, acc0, [&](int c) {
                if (c < 3) dq[c + 1] = read_dis(p_slot, 5 + c);
                read_res(res0, 0, c);
                epi_row(acc1, res1, c, p_orow[c], p_krow[c >> 1], dq[c]);
            });
            PSTAMP(0);
#pragma unroll
            for (int e = 0; e < 16; ++e) acc1[e] = 0.f;
            // ---- rows 32..63: MFMA chain with the epilogue of rows 0..31 in its gaps
            float* orow0[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) orow0[c] = fast ? ofr + (size_t)seg_first[c] * C : sink;
            float* krow0[2];
#pragma unroll
            for (int c = 0; c < 2; ++c) krow0[c] = (fast && KOUT) ? kfr + (size_t)par0[c] * C : sink;
            dq[0] = read_dis(slot, 0);
            mfma_rowblock_with<true>(s_a, 32, lane, wreg, acc1, [&](int c) {
                if (c < 3) dq[c + 1] = read_dis(slot, 1 + c);
                read_res(res1, 1, c);
                epi_row(acc0, res0, c, orow0[c], krow0[c >> 1], dq[c]);
            });
            PSTAMP(1);
#pragma unroll
            for (int c = 0; c < 4; ++c) p_orow[c] = fast ? ofr + (size_t)seg_first[4 + c] * C : sink;
#pragma unroll
            for (int c = 0; c < 2; ++c) p_krow[c] = (fast && KOUT) ? kfr + (size_t)par0[2 + c] * C : sink;
            p_slot = slot;
            if (!fast) {
                // ---- ragged patches, levels without parents (rare): both blocks finished here with conditional stores (the
                // unconditional epilogues of this tile went / go to the dump area); the child sums go through this wave's own
                // channel slice of the stash
                float* s_xw = s_x0 + (k & 1) * TILE * LDA;
#pragma unroll
                for (int rb = 0; rb < 2; ++rb) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int p = 4 * rb + (e >> 2), col = (e & 3) + 4 * h;
                        float t = fmaf(rb ? acc1[e] : acc0[e], scj, shj);
                        t = fmaxf(t, relu_floor);
                        t += has_res ? (rb ? res1[e] : res0[e]) : 0.f;
                        if (KOUT) s_xw[(8 * p + col) * LDA + ch] = t;
                        if (col < seg_cnt[p]) ofr[(size_t)(seg_first[p] + col) * C + ch] = t;
                    }
                }
                if (KOUT) {
                    // lane -> (parent q = (lane >> 3) + 8 i, 16-B chunk lane & 7) of the wave's 32 channels
                    const int c4 = 4 * (lane & 7);
                    const float* dsl = s_dis0 + slot * TILE;
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const int q = (lane >> 3) + 8 * i, pr = q >> 2, pc = q & 3;
                        const int np = pr == 0 ? npar[0] : (pr == 1 ? npar[1] : (pr == 2 ? npar[2] : npar[3]));
                        const int pb = pr == 0 ? par0[0] : (pr == 1 ? par0[1] : (pr == 2 ? par0[2] : par0[3]));
                        const int ra = 16 * pr + 2 * pc;                       // LDS row of child (2 pr, 2 pc)
                        const float* sp = s_xw + ra * LDA + 32 * wave + c4;
                        const float* dp = dsl + ra;
                        f32x4 kk = dp[0] * *reinterpret_cast<const f32x4*>(sp);
                        kk += dp[1] * *reinterpret_cast<const f32x4*>(sp + LDA);
                        kk += dp[8] * *reinterpret_cast<const f32x4*>(sp + 8 * LDA);
                        kk += dp[9] * *reinterpret_cast<const f32x4*>(sp + 9 * LDA);
                        if (pc < np) *reinterpret_cast<f32x4*>(kfr + (size_t)(pb + pc) * C + 32 * wave + c4) = kk;
                    }
                }
            }
            __syncthreads();                               // barrier k+1: buffer (k+1)&1 is full, buffer k&1 is free
            PSTAMP(2);
        }
        {   // the last tile's second block (or the dump area)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const f32x4 d = KOUT ? *reinterpret_cast<const f32x4*>(s_dis0 + 4 * hq + p_slot * TILE + 8 * (4 + c)) : f32x4{0.f, 0.f, 0.f, 0.f};
                epi_row(acc1, res1, c, p_orow[c], p_krow[c >> 1], d);
            }
        }
        PSTAMP_FLUSH(0);
#ifdef EG_STAMP
        if (wave == 0 && lane_k == 0) {
            unsigned long long* stats = reinterpret_cast<unsigned long long*>(counters + WALK_GROUPS * WALK_CTR_STRIDE);
            atomicAdd(&stats[9], __builtin_amdgcn_s_memtime() - clk0);
            atomicAdd(&stats[10], __builtin_amdgcn_s_memrealtime() - rt0);
        }
#endif
        } else {
        // =========================== CONSUMER with the classifier heads: wave = head ==============================
        // The stack's last layer has no activation (models.py:335), so layer + residual + first classifier layer are ONE
        // linear map of what is in LDS anyway:
        //     hidden_pre = (A_hat x) Wc^T + x W1^T + const,    Wc = W1 diag(scale) W,   const = W1 shift   (eg_cls_fold)
        // i.e. a K = 256 product over the aggregated tile and the stash of raw rows: no output tile, no LDS round trip, no
        // meeting point of the four waves.  Orientation channels x rows (lane = tile row, 16 hidden channels per lane), so
        // Linear(32,16) runs straight from the accumulator: step t of v_mfma_f32_32x32x2_f32 takes accumulator register t as its
        // B operand (k = lane half <-> hidden channel (t & 3) + 8 (t >> 2) + 4 h) against W2 laid out the same way; BN + ReLU +
        // the 16-wide dot follow on the result (8 outputs per lane, one cross-half add).
        float wA[64], wB[64];
        load_w_slice(ca.wc, wave, lane_k, 0, wA);
        const bool has_res = a.has_res != 0;
        if (has_res) load_w_slice(ca.w1, wave, lane_k, 0, wB);
        else {
#pragma unroll
            for (int i = 0; i < 64; ++i) wB[i] = 0.f;
        }
        const int j = lane_k & 31, hq = lane_k >> 5;
        float w2r[16];                                    // W2[head][o = j][hidden channel of (step t, half hq)], 0 for o >= 16
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const int c = (t & 3) + 8 * (t >> 2) + 4 * hq;
            w2r[t] = j < 16 ? ca.w2[(size_t)(wave * 16 + j) * 32 + c] : 0.f;
        }
        __syncthreads();                                   // tile 0 is in buffer 0
        PSTAMP_INIT;
#ifdef EG_STAMP
        const unsigned long long clk0 = __builtin_amdgcn_s_memtime(), rt0 = __builtin_amdgcn_s_memrealtime();
#endif
        // s_bn: [0,128) s1, [128,256) t1c (first-layer scale / shift per hidden channel); second-layer constants per head
        const float* const s1p = s_bn + 32 * wave + 4 * hq;               // + 8 g: the lane's hidden channels 4g .. 4g+3 of group g
        const float* const t1p = s1p + C;
        // second-layer result: lane (row, half hq') holds outputs o = (e & 3) + 8 (e >> 2) + 4 hq', e = 0..7
        const float* const l2p = s_bn + 2 * C + wave * 16 + 4 * hq;      // s2 | t2 | w3 ([3][64]) at this lane's outputs
        const float b3v = ca.b3[wave];
        // hidden = relu(s1 * pre + t1c) in place on an accumulator (group g = registers 4g .. 4g+3)
        auto hidden_group = [&](f32x16& acc, int g) {
            const f32x4 sv = *reinterpret_cast<const f32x4*>(s1p + 8 * g), tv = *reinterpret_cast<const f32x4*>(t1p + 8 * g);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[4 * g + e] = fmaxf(fmaf(acc[4 * g + e], sv[e], tv[e]), 0.f);
        };
        auto second_layer = [&](const f32x16& hid, float (&zs)[8]) {       // (only outputs 0..15 = registers 0..7 are kept)
            f32x16 z;
#pragma unroll
            for (int e = 0; e < 16; ++e) z[e] = 0.f;
#pragma unroll
            for (int t = 0; t < 16; ++t) z = __builtin_amdgcn_mfma_f32_32x32x2f32(w2r[t], hid[t], z, 0, 0, 0);
#pragma unroll
            for (int e = 0; e < 8; ++e) zs[e] = z[e];
        };
        // BN + ReLU + Linear(16,1) (+ sigmoid) of one 32-row block and its store; lg: this lane's logit address (a word of the
        // dump area for lanes without a node: no branch inside an MFMA chain)
        const bool sigm = ca.sigmoid != 0;
        auto third_layer = [&](const float (&z)[8], int node) {          // node: frame rows + node id of this lane's row, -1: none
            const f32x4 s2a = *reinterpret_cast<const f32x4*>(l2p), s2b = *reinterpret_cast<const f32x4*>(l2p + 8);
            const f32x4 t2a = *reinterpret_cast<const f32x4*>(l2p + 64), t2b = *reinterpret_cast<const f32x4*>(l2p + 72);
            const f32x4 w3a = *reinterpret_cast<const f32x4*>(l2p + 128), w3b = *reinterpret_cast<const f32x4*>(l2p + 136);
            float y = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                y = fmaf(w3a[e], fmaxf(fmaf(z[e], s2a[e], t2a[e]), 0.f), y);
                y = fmaf(w3b[e], fmaxf(fmaf(z[4 + e], s2b[e], t2b[e]), 0.f), y);
            }
            y += __shfl_xor(y, 32);
            y += b3v;
            const float ys = 1.0f / (1.0f + __expf(-y));
            float* const lg = node >= 0 ? ca.logits + ((size_t)(unsigned)node * 4 + wave)
                                        : sink_base + ((size_t)blockIdx.x * PS_SINK_FLOATS + lane_k);
            *lg = sigm ? ys : y;
        };
        f32x16 acc0, acc1;
        float z0[8], z1[8];
        int lg0 = -1, lg1 = -1;                           // logit rows of the previous tile's two blocks (finished one tile late)
#pragma unroll
        for (int e = 0; e < 16; ++e) { acc0[e] = 0.f; acc1[e] = 0.f; }
#pragma unroll
        for (int e = 0; e < 8; ++e) { z0[e] = 0.f; z1[e] = 0.f; }
        for (int k = 0;; ++k) {
            const int t_cur = __builtin_amdgcn_readfirstlane(s_tile[k & 7]);
            if (t_cur < 0) break;
            int lane = lane_k;
            asm volatile("" : "+v"(lane));
            const float* s_a = s_a0 + (k & 1) * TILE * LDA;
            const float* s_x = s_x0 + (k & 1) * TILE * LDA;
            const int frame = t_cur / a.tiles_per_frame;
            // this lane's node of each block: patch row 4 rb + (j >> 3), column j & 7 (descriptor ring: {n_first, cnt, ., .})
            const int* cdp = s_cd + (k & 1) * 32 + 4 * (j >> 3);
            const int f0 = cdp[0], c0 = cdp[1], f1 = cdp[16], c1 = cdp[17];
            const int nlg0 = (hq == 0 && (j & 7) < c0) ? frame * a.n_per_frame + f0 + (j & 7) : -1;
            const int nlg1 = (hq == 0 && (j & 7) < c1) ? frame * a.n_per_frame + f1 + (j & 7) : -1;
#pragma unroll
            for (int e = 0; e < 16; ++e) acc0[e] = 0.f;
            // rows 0..31: K = 128 over the aggregated tile (third layers of the previous tile in the gaps), K = 128 over the raw rows
            mfma_rowblock_lean<false>(s_a, 0, lane, wA, acc0, [&](int c) {
                if (c == 0) third_layer(z0, lg0);
                if (c == 1) third_layer(z1, lg1);
            });
            if (has_res) mfma_rowblock_lean<false>(s_x, 0, lane, wB, acc0, [](int) {});
            PSTAMP(0);
#pragma unroll
            for (int e = 0; e < 16; ++e) acc1[e] = 0.f;
            // rows 32..63, with BN + ReLU of rows 0..31 in the gaps
            mfma_rowblock_lean<false>(s_a, 32, lane, wA, acc1, [&](int c) { hidden_group(acc0, c); });
            if (has_res) mfma_rowblock_lean<false>(s_x, 32, lane, wB, acc1, [](int) {});
            PSTAMP(1);
            // second layers (16 MFMAs per block); BN + ReLU of rows 32..63 runs beside those of rows 0..31
            second_layer(acc0, z0);
#pragma unroll
            for (int g = 0; g < 4; ++g) hidden_group(acc1, g);
            second_layer(acc1, z1);
            lg0 = nlg0;
            lg1 = nlg1;
            PSTAMP(3);
            __syncthreads();                               // barrier k+1: buffer (k+1)&1 is full, buffer k&1 is free
            PSTAMP(2);
        }
        third_layer(z0, lg0);                              // the last tile's blocks
        third_layer(z1, lg1);
        PSTAMP_FLUSH(0);
#ifdef EG_STAMP
        if (wave == 0 && lane_k == 0) {
            unsigned long long* stats = reinterpret_cast<unsigned long long*>(counters + WALK_GROUPS * WALK_CTR_STRIDE);
            atomicAdd(&stats[9], __builtin_amdgcn_s_memtime() - clk0);
            atomicAdd(&stats[10], __builtin_amdgcn_s_memrealtime() - rt0);
        }
#endif
        }
    } else {
        // =========================== PRODUCER: patch rows 2p, 2p+1 of every tile ================================
        const int p = wave - 4;
#ifndef EG_PS_NO_PRIO
        __builtin_amdgcn_s_setprio(3);
#endif
        PSTAMP_INIT;
        // The two segment descriptors of a tile (32 dwords) travel in ONE VGPR, lane l holding dword l & 31: they are
        // fetched a whole tile ahead (a scalar load at the point of use costs a loaded-memory round trip, ~9k cycles
        // measured, in front of the row loads) and turned into SGPRs with v_readlane when the tile is produced.
        auto load_desc = [&](int tile_i, int lane) -> int {
            const int frame = tile_i / a.tiles_per_frame;
            const int t_in = tile_i - frame * a.tiles_per_frame;
            return reinterpret_cast<const int*>(segs)[(t_in * 8 + 2 * p) * 16 + (lane & 31)];
        };
        auto desc_of = [](int dv, int o) -> SegDesc {
            SegDesc d;
            d.n_first = __builtin_amdgcn_readlane(dv, o + 0);  d.cnt = __builtin_amdgcn_readlane(dv, o + 1);
            d.mode = __builtin_amdgcn_readlane(dv, o + 2);     d.pat = __builtin_amdgcn_readlane(dv, o + 3);
            d.up0 = __builtin_amdgcn_readlane(dv, o + 4);      d.down0 = __builtin_amdgcn_readlane(dv, o + 5);
            d.par0 = __builtin_amdgcn_readlane(dv, o + 6);     d.left = __builtin_amdgcn_readlane(dv, o + 7);
            d.right = __builtin_amdgcn_readlane(dv, o + 8);    d.c0 = __builtin_amdgcn_readlane(dv, o + 9);
            d.c1 = __builtin_amdgcn_readlane(dv, o + 10);      d.c2 = __builtin_amdgcn_readlane(dv, o + 11);
            d.c3 = __builtin_amdgcn_readlane(dv, o + 12);      d.aux = __builtin_amdgcn_readlane(dv, o + 13);
            d.pad0 = __builtin_amdgcn_readlane(dv, o + 14);    d.pad1 = __builtin_amdgcn_readlane(dv, o + 15);
            return d;
        };
        // The rows of a tile sit in R (112 VGPRs); `have` = R holds -- in flight since the previous period -- the rows of the
        // tile about to be produced.  produce() aggregates tile_i into buffer `buf` and, stage by stage, re-issues every
        // register group for tile_n (descriptors dvn) right after its last use; returns whether R now holds tile_n.
#ifdef EG_ABL_HALO      // timing-only ablation (results wrong): the rows above / below a pair are loaded only where no other wave
                        // of the workgroup loads them as its own rows -- the bytes a halo exchange through LDS would save
        const bool HALO_U = p == 0, HALO_D = p == 3;
#else
        const bool HALO_U = true, HALO_D = true;
#endif
        PairRegs R;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            R.Sa[k] = R.Sb[k] = R.U[k] = R.D[k] = R.Ka[k] = R.Kb[k] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        R.LRa = R.LRb = R.P[0] = R.P[1] = f32x4{0.f, 0.f, 0.f, 0.f};
        auto pick = [](bool c, const SegDesc& u, const SegDesc& v) -> SegDesc {          // wave-uniform select, field by field
            SegDesc d;
            d.n_first = c ? u.n_first : v.n_first;  d.cnt = c ? u.cnt : v.cnt;      d.mode = c ? u.mode : v.mode;
            d.pat = c ? u.pat : v.pat;              d.up0 = c ? u.up0 : v.up0;      d.down0 = c ? u.down0 : v.down0;
            d.par0 = c ? u.par0 : v.par0;           d.left = c ? u.left : v.left;   d.right = c ? u.right : v.right;
            d.c0 = c ? u.c0 : v.c0;  d.c1 = c ? u.c1 : v.c1;  d.c2 = c ? u.c2 : v.c2;  d.c3 = c ? u.c3 : v.c3;
            d.aux = c ? u.aux : v.aux;              d.pad0 = c ? u.pad0 : v.pad0;   d.pad1 = c ? u.pad1 : v.pad1;
            return d;
        };
        auto produce = [&](int tile_i, int buf, int dslot, int lane, int dv, bool have, int tile_n, int dvn) -> bool {
            const int frame = tile_i / a.tiles_per_frame;
            const float* __restrict__ xf = x + (size_t)frame * a.n_per_frame * C;
            const SegDesc sd0 = desc_of(dv, 0);
            const SegDesc sd1 = desc_of(dv, 16);
            {   // the consumers' descriptor words of this tile (segments 2p, 2p+1: words 0, 1, 6, 15 -> ring slot [seg][0..3])
                const int w = lane & 15;
                const int f = w == 0 ? 0 : (w == 1 ? 1 : (w == 6 ? 2 : (w == 15 ? 3 : -1)));
                if (lane < 32 && f >= 0) s_cd[buf * 32 + (2 * p + (lane >> 4)) * 4 + f] = dv;
            }
            // the tile after this one: pipe-able when both of its segments are on the pair path; otherwise the re-issued
            // loads simply fetch this tile's rows again (no branch around a write to the loop-carried register set: the
            // allocator would keep both generations alive) and the next call starts from scratch
            const SegDesc m0 = desc_of(dvn, 0), m1 = desc_of(dvn, 16);
            const bool nxt = tile_n >= 0 && m0.pad0 != 0;
            const SegDesc n0 = pick(nxt, m0, sd0), n1 = pick(nxt, m1, sd1);
            const int frame_n = nxt ? tile_n / a.tiles_per_frame : frame;
            const float* __restrict__ xfn = x + (size_t)frame_n * a.n_per_frame * C;
            const bool kin_n = KIN && nxt && n0.aux;
            const float* __restrict__ kfn = KIN ? kin + (size_t)frame_n * a.kid_rows * C : xfn;
            const PairLane pl{lane >> 5, lane & 31};
            float* s_a = s_a0 + buf * TILE * LDA;
            float* s_x = a.has_res ? s_x0 + buf * TILE * LDA : nullptr;
            f32x4 acc0[4], acc1[4];
            // everything of a pair at once, in the order the stages consume it (vmcnt counts in issue order)
            auto issue_all = [&](const SegDesc& u0, const SegDesc& u1, const float* __restrict__ xf_, const float* __restrict__ kf_, bool k_) {
                R.LRa = *reinterpret_cast<const f32x4*>(xf_ + bcast_off(pl.h ? u0.left : u0.right, pl));
                if constexpr (KIN) {
                    if (k_) {
                        const unsigned oa = pair_off(u0.n_first, pl);
#pragma unroll
                        for (int k = 0; k < 4; ++k) R.Ka[k] = ld4(kf_, oa, k);
                    } else {
#pragma unroll
                        for (int k = 0; k < 4; ++k) R.Ka[k] = f32x4{0.f, 0.f, 0.f, 0.f};
                    }
                }
                if (HALO_U) pair_issue_u(u0, xf_, pl, R);
                R.LRb = *reinterpret_cast<const f32x4*>(xf_ + bcast_off(pl.h ? u1.left : u1.right, pl));
                if constexpr (KIN) {
                    if (k_) {
                        const unsigned ob = pair_off(u1.n_first, pl);
#pragma unroll
                        for (int k = 0; k < 4; ++k) R.Kb[k] = ld4(kf_, ob, k);
                    } else {
#pragma unroll
                        for (int k = 0; k < 4; ++k) R.Kb[k] = f32x4{0.f, 0.f, 0.f, 0.f};
                    }
                }
                pair_issue_s(u0, u1, xf_, pl, R);
                if (HALO_D) pair_issue_d(u1, xf_, pl, R);
                pair_issue_p(u0, xf_, pl, R);
            };
            if (sd0.pad0) {
                // ---- usual case: both segments on the fast path, vertically adjacent, same parents ----
                const bool use_kin = KIN && sd0.aux;                // uniform: children already summed by the previous layer
                if (!have) issue_all(sd0, sd1, xf, KIN ? kin + (size_t)frame * a.kid_rows * C : xf, use_kin);    // nothing in flight for this tile
                __builtin_amdgcn_sched_barrier(0);
                PSTAMP(0);
                const float* wqa = s_pat + sd0.pat * PATQ + 32 * (lane >> 5);      // this lane's weights (LDS, quad layout)
                const float* wqb = s_pat + sd1.pat * PATQ + 32 * (lane >> 5);
                if (KIN || !sd0.aux) {
                    // ---- pipelined form: segment a is finished before segment b (one accumulator set live at a time), and
                    // every register group goes out again for the next tile right after its last use
                    // segment a: self + left + right (raw rows into the stash), child sums, above, below (= segment b), parents
                    pair_stage_self(lane, wqa, R.Sa, R.LRa, acc0, s_x, 16 * p);
                    pin_acc4(acc0);
                    __builtin_amdgcn_sched_barrier(0);
                    R.LRa = *reinterpret_cast<const f32x4*>(xfn + bcast_off(pl.h ? n0.left : n0.right, pl));
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (KIN) {
                        if (use_kin) {
                            const f32x4 fa = quad_w(wqa, SLOT_HASKIDS);            // 1.0 / 0.0
#pragma unroll
                            for (int k = 0; k < 4; ++k) acc0[k] += fa[k] * R.Ka[k];
                        }
                        pin_acc4(acc0);
                        __builtin_amdgcn_sched_barrier(0);
                        if (kin_n) {
                            const unsigned oa = pair_off(n0.n_first, pl);
#pragma unroll
                            for (int k = 0; k < 4; ++k) R.Ka[k] = ld4(kfn, oa, k);
                        } else {
#pragma unroll
                            for (int k = 0; k < 4; ++k) R.Ka[k] = f32x4{0.f, 0.f, 0.f, 0.f};
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    pair_stage_rows(wqa, SLOT_UP, R.U, acc0);
                    pin_acc4(acc0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (HALO_U) pair_issue_u(n0, xfn, pl, R);
                    __builtin_amdgcn_sched_barrier(0);
                    pair_stage_rows(wqa, SLOT_DOWN, R.Sb, acc0);
                    {
                        const f32x4 w5 = quad_w(wqa, SLOT_PARENT);
                        f32x4 plo[2], pup[2];
                        halves(R.P[0], plo[0], pup[0]);
                        halves(R.P[1], plo[1], pup[1]);
#pragma unroll
                        for (int k = 0; k < 4; ++k) acc0[k] += w5[k] * ((k & 1) ? pup[k >> 1] : plo[k >> 1]);
                    }
                    segw_store(lane, wqa, acc0, s_a, 16 * p);
                    pin_acc4(acc0);
                    __builtin_amdgcn_sched_barrier(0);
                    // segment b: self + left + right, child sums, above (= segment a), below, parents
                    pair_stage_self(lane, wqb, R.Sb, R.LRb, acc1, s_x, 16 * p + 8);
                    pin_acc4(acc1);
                    __builtin_amdgcn_sched_barrier(0);
                    R.LRb = *reinterpret_cast<const f32x4*>(xfn + bcast_off(pl.h ? n1.left : n1.right, pl));
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (KIN) {
                        if (use_kin) {
                            const f32x4 fb = quad_w(wqb, SLOT_HASKIDS);
#pragma unroll
                            for (int k = 0; k < 4; ++k) acc1[k] += fb[k] * R.Kb[k];
                        }
                        pin_acc4(acc1);
                        __builtin_amdgcn_sched_barrier(0);
                        if (kin_n) {
                            const unsigned ob = pair_off(n1.n_first, pl);
#pragma unroll
                            for (int k = 0; k < 4; ++k) R.Kb[k] = ld4(kfn, ob, k);
                        } else {
#pragma unroll
                            for (int k = 0; k < 4; ++k) R.Kb[k] = f32x4{0.f, 0.f, 0.f, 0.f};
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    pair_stage_rows(wqb, SLOT_UP, R.Sa, acc1);
                    pin_acc4(acc1);
                    __builtin_amdgcn_sched_barrier(0);
                    {   // both segments' self rows are dead now
                        const unsigned osa = pair_off(n0.n_first, pl), osb = pair_off(n1.n_first, pl);
#pragma unroll
                        for (int k = 0; k < 4; ++k) R.Sa[k] = ld4(xfn, osa, k);
#pragma unroll
                        for (int k = 0; k < 4; ++k) R.Sb[k] = ld4(xfn, osb, k);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    pair_stage_rows(wqb, SLOT_DOWN, R.D, acc1);
                    pin_acc4(acc1);
                    __builtin_amdgcn_sched_barrier(0);
                    if (HALO_D) pair_issue_d(n1, xfn, pl, R);
                    __builtin_amdgcn_sched_barrier(0);
                    {
                        const f32x4 w5 = quad_w(wqb, SLOT_PARENT);
                        f32x4 plo[2], pup[2];
                        halves(R.P[0], plo[0], pup[0]);
                        halves(R.P[1], plo[1], pup[1]);
#pragma unroll
                        for (int k = 0; k < 4; ++k) acc1[k] += w5[k] * ((k & 1) ? pup[k >> 1] : plo[k >> 1]);
                    }
                    pin_acc4(acc1);
                    __builtin_amdgcn_sched_barrier(0);
                    pair_issue_p(n0, xfn, pl, R);
                    __builtin_amdgcn_sched_barrier(0);
                    segw_store(lane, wqb, acc1, s_a, 16 * p + 8);
                    PSTAMP(1);
                } else if constexpr (!KIN) {
                    // ---- aux level of an unchained / first layer: the children are pulled as rows (64 more registers), so the
                    // next tile's rows go out only when this tile is done
                    pair_stage_self(lane, wqa, R.Sa, R.LRa, acc0, s_x, 16 * p);
                    pair_stage_self(lane, wqb, R.Sb, R.LRb, acc1, s_x, 16 * p + 8);
                    pair_stage_rows(wqa, SLOT_UP, R.U, acc0);
                    pair_stage_rows(wqb, SLOT_UP, R.Sa, acc1);
                    pair_stage_rows(wqa, SLOT_DOWN, R.Sb, acc0);
                    pair_stage_rows(wqb, SLOT_DOWN, R.D, acc1);
                    pair_stage_par(wqa, wqb, R.P, acc0, acc1);
                    pair_pin(acc0, acc1);                           // the main-stage registers are dead from here on
                    {
                        SegKids K;
                        segw_kids_issue(sd0, pats, xf, lane, K);
                        segw_kids_add(lane, K, acc0);
                    }
                    pin_acc4(acc0);                                 // one segment's 16 child loads in flight at a time (registers)
                    {
                        SegKids K;
                        segw_kids_issue(sd1, pats, xf, lane, K);
                        segw_kids_add(lane, K, acc1);
                    }
                    pair_pin(acc0, acc1);
                    PSTAMP(1);
                    issue_all(n0, n1, xfn, kfn, kin_n);
                    __builtin_amdgcn_sched_barrier(0);
                    segw_store(lane, wqa, acc0, s_a, 16 * p);
                    segw_store(lane, wqb, acc1, s_a, 16 * p + 8);
                }
                if (kout && (lane & 31) == 0) {                     // (deg+1)^-1/2 of the 16 nodes, for the consumers' child sums
                    const f32x4 da = quad_w(wqa, SLOT_SELF), db = quad_w(wqb, SLOT_SELF);
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        s_dis0[dslot * TILE + 16 * p + 2 * k + (lane >> 5)] = da[k];
                        s_dis0[dslot * TILE + 16 * p + 8 + 2 * k + (lane >> 5)] = db[k];
                    }
                }
            } else {
                // ---- ragged patches, coordinate nodes, frame end (rare): node by node, scalar neighbour decode ----
#pragma unroll 1
                for (int e = 0; e < 2; ++e) {
                    const int n0f = e ? sd1.n_first : sd0.n_first, cnt = e ? sd1.cnt : sd0.cnt;
                    const int rl = 16 * p + 8 * e;
#pragma unroll 1
                    for (int u = 0; u < cnt; ++u) {
                        *reinterpret_cast<f32x2*>(&s_a[(rl + u) * LDA + 2 * lane]) = agg_stencil(T, xf, dis, n0f + u, lane);
                        if (s_x) *reinterpret_cast<f32x2*>(&s_x[(rl + u) * LDA + 2 * lane]) = load_row2(xf, n0f + u, lane);
                    }
                }
                issue_all(n0, n1, xfn, kfn, kin_n);                 // (nothing to pipe: this tile's own rows again, never used)
            }
            PSTAMP(2);
            return nxt;
        };
        // descriptors travel two tiles ahead (dv_a: the tile produced next, dv_b: the one after it, whose rows are issued)
        int dv_a = 0, dv_b = 0;
        bool have = false;
        {
            const int t0 = __builtin_amdgcn_readfirstlane(s_tile[0]);
            const int t1 = __builtin_amdgcn_readfirstlane(s_tile[1]);
            const int t2 = __builtin_amdgcn_readfirstlane(s_tile[2]);
            int dv0 = 0;
            if (t0 >= 0) dv0 = load_desc(t0, lane_k);
            if (t1 >= 0) dv_a = load_desc(t1, lane_k);
            if (t2 >= 0) dv_b = load_desc(t2, lane_k);
            if (t0 >= 0) have = produce(t0, 0, 0, lane_k, dv0, false, t1, dv_a);
        }
        __syncthreads();                                   // tile 0 is in buffer 0
        for (int k = 0;; ++k) {
            const int t_cur = __builtin_amdgcn_readfirstlane(s_tile[k & 7]);
            if (t_cur < 0) break;
            const int t_next = __builtin_amdgcn_readfirstlane(s_tile[(k + 1) & 7]);
            const int t_nn = __builtin_amdgcn_readfirstlane(s_tile[(k + 2) & 7]);
            const int t_n3 = __builtin_amdgcn_readfirstlane(s_tile[(k + 3) & 7]);
            int lane = lane_k;
            asm volatile("" : "+v"(lane));
            const int dv_cur = dv_a, dv_nxt = dv_b;
            dv_a = dv_b;
            if (t_n3 >= 0) dv_b = load_desc(t_n3, lane);                                    // used two iterations from now
            int got = 0;
            if (tid == 256) got = ps_claim_issue(counters, group);                          // four tiles ahead, asynchronous
            PSTAMP(3);
#ifndef EG_ABL_NO_PROD       // timing-only ablation (results wrong): the consumers alone, on whatever is in LDS
            if (t_next >= 0) have = produce(t_next, (k + 1) & 1, (k + 1) & (PS_DIS_RING - 1), lane, dv_cur, have, t_nn, dv_nxt);
#endif
            if (tid == 256) ps_claim_commit(counters, group, n_tiles, got, &s_tile[(k + 4) & 7]);
            __syncthreads();                               // barrier k+1
            PSTAMP(3);
        }
        PSTAMP_FLUSH(4);
    }
}

}  // namespace eg

using namespace eg;

// Used by eg_gcn_layer_fwd (gcn_layer.hip) for topology handles when the residual is NULL or x itself.
// Returns EG_ERR_UNSUPPORTED when the caller should fall back to the symmetric kernel.
int eg_launch_layer_ps(const eg_graph* g, int batch, const float* x, const float* W, const float* scale,
                       const float* shift, const float* residual, int relu, int transpose_w, float* out,
                       const float* kin, float* kout, const eg::ClsArgs* cls, hipStream_t stream) {
    if (!g || g->kind != GRAPH_TOPO || (residual != nullptr && residual != x)) return EG_ERR_UNSUPPORTED;
    const bool chained = kin || kout;
    if (chained && g->kid_rows == 0) return EG_ERR_UNSUPPORTED;
    if (cls && g->kid_rows == 0 && !g->flat) return EG_ERR_UNSUPPORTED;
    // plain calls: this kernel by default on single-level topologies (no tiles that pull child rows), the symmetric
    // kernel otherwise; EG_LAYER_IMPL = 0 / 1 forces one of them
    if (!chained && !cls && (g->knobs.layer_impl < 0 ? g->flat : g->knobs.layer_impl) == 0) return EG_ERR_UNSUPPORTED;
    PsDims a{};
    a.n_per_frame = (int)g->n_nodes; a.batch = batch; a.tiles_per_frame = g->n_tiles;
    a.relu = relu; a.transpose_w = transpose_w; a.has_res = residual != nullptr;
    a.kid_rows = g->kid_rows; a.n_pats = g->n_pats;
    const long long n_tiles = (long long)a.tiles_per_frame * batch;
    if (n_tiles <= 0) return EG_OK;
    const size_t lds = ps_lds_bytes(g->n_pats, cls != nullptr);
    if (lds > 160 * 1024) return EG_ERR_UNSUPPORTED;             // more weight patterns than fit beside the tile buffers
    {   // 160 KB of dynamic LDS needs the attribute once per device (idempotent, so a benign race sets it twice at worst)
        static std::atomic<bool> attr_set[64];
        int dev = 0;
        EG_HIP_TRY(hipGetDevice(&dev));
        if (dev < 0 || dev >= 64 || !attr_set[dev].load(std::memory_order_acquire)) {
            const void* kernels[] = {(const void*)k_gcn_layer_ps<false, false, false>, (const void*)k_gcn_layer_ps<false, false, true>,
                                     (const void*)k_gcn_layer_ps<false, true, false>, (const void*)k_gcn_layer_ps<false, true, true>,
                                     (const void*)k_gcn_layer_ps<true, false, false>, (const void*)k_gcn_layer_ps<true, false, true>};
            for (const void* f : kernels) EG_HIP_TRY(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            if (dev >= 0 && dev < 64) attr_set[dev].store(true, std::memory_order_release);
        }
    }
    int* const queue = g->next_queue_slice();
    EG_HIP_TRY(hipMemsetAsync(queue, 0, sizeof(int) * QUEUE_SLICE_INTS, stream));
    long long grid = n_tiles < 256 ? n_tiles : g->knobs.ps_grid;      // one persistent workgroup per CU
    if (grid > PS_MAX_GRID) grid = PS_MAX_GRID;
    const ClsArgs none{};
    auto launch = [&](auto kernel) {
        hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(PS_THREADS), lds, stream, x, W, scale, shift, out, g->dis, g->topo_dev,
                           g->tiles_dev, g->segs_dev, g->pats_dev, g->patsq_dev, kin, kout, g->sink, queue, a, cls ? *cls : none);
    };
    if (cls) { if (kin) launch(k_gcn_layer_ps<true, false, true>); else launch(k_gcn_layer_ps<true, false, false>); }
    else if (kout) { if (kin) launch(k_gcn_layer_ps<false, true, true>); else launch(k_gcn_layer_ps<false, true, false>); }
    else { if (kin) launch(k_gcn_layer_ps<false, false, true>); else launch(k_gcn_layer_ps<false, false, false>); }
    EG_HIP_TRY(hipGetLastError());
    return EG_OK;
}
