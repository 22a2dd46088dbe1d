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
#include "train_common.h"

namespace eg {

constexpr int PS_THREADS = 512;
constexpr int PS_LDS_DIS = 4 * TILE * LDA + 16 + 64;  // float offsets inside the dynamic LDS block (tile-id ring, CLS counter, descriptor ring)
constexpr int PS_LDS_PAT = PS_LDS_DIS + 2 * TILE;

struct PsDims {
    int n_per_frame, batch, tiles_per_frame, relu, transpose_w, has_res;
    int kid_rows;      // rows per frame of the child-sum side buffers (kin / kout)
    int n_pats;
    // train forward (TRAIN = true): the aggregated rows A_hat x are kept (agg_out, nullable), the column sums of the output and
    // its square -- the BatchNorm batch statistics -- leave per workgroup (stats_partial [grid][2][128]), and tiles are walked
    // statically (a workgroup's accumulation order must not depend on who wins a queue: bit-reproducible statistics)
    float* agg_out;
    float* stats_partial;
    int static_walk;
    const float* res;  // MODE 2: residual rows (has_res is set; the stash holds THESE rows, not the input's)
    // connection nodes (DIAG instantiations; conn.hip): this launch's slice of the handle's scratch, floats per frame in it, and the
    // offsets of the connection nodes' aggregated / scaled rows inside a frame's part
    const float* conn;
    int conn_stride, conn_agg, conn_scaled, n_conn;
    int self_reset;    // the last workgroup out zeroes the launch's slice of the queue ring (no memset in front of the next user)
    LowerSums lower;   // MODE 3: the BatchNorm-backward sums of the layer below, taken from the rows this launch writes (common.h)
};

// Static walk: blockIdx % 8 labels the chunk of the tile order (round-robin dispatch puts those workgroups on one XCD: a
// heuristic that only locality depends on), a workgroup takes every `per`-th tile of its chunk.
__device__ inline int ps_static_tile(int ord, int n_tiles) {
    const int chunk = (n_tiles + WALK_GROUPS - 1) / WALK_GROUPS;
    const int g = blockIdx.x % WALK_GROUPS, l = blockIdx.x / WALK_GROUPS;
    const int per = ((int)gridDim.x + WALK_GROUPS - 1 - g) / WALK_GROUPS;
    const int end = (g + 1) * chunk < n_tiles ? (g + 1) * chunk : n_tiles;
    const long long t = (long long)g * chunk + l + (long long)per * ord;
    return t < end ? (int)t : -1;
}

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

// EG_PROBE_SITE / EG_PROBE_N (experiment): N extra vector instructions (v_mov on two scratch registers) at ONE place of the
// tile loop -- how many cycles does an instruction COST there?  (1 producer: behind the load issue; 2 producer: behind the main
// stage, in front of the child sums / LDS stores; 3 producer: end of the tile; 4 consumer: top of the tile; 5 consumer: end of
// the tile, in front of the barrier)
#ifdef EG_PROBE_SITE
#ifndef EG_PROBE_N
#define EG_PROBE_N 64
#endif
#define EG_PROBE(site) do { if ((site) == EG_PROBE_SITE) { int pa_ = lane, pb_ = lane; _Pragma("unroll") for (int i_ = 0; i_ < EG_PROBE_N / 2; ++i_) { \
    asm volatile("v_mov_b32 %0, %0" : "+v"(pa_)); asm volatile("v_mov_b32 %0, %0" : "+v"(pb_)); } asm volatile("" :: "v"(pa_), "v"(pb_)); } } while (0)
#else
#define EG_PROBE(site) do {} while (0)
#endif

// frame = tile / tiles_per_frame (the divisor is loop-invariant: the compiler keeps its reciprocal, a division is ~16 scalar
// instructions).  EG_FRAME_HINT: try the last frame first (consecutive tiles of a queue mostly lie in one frame).
__device__ inline int frame_of(int tile, int tiles_per_frame, int& hint) {
#ifdef EG_FRAME_HINT       // measured (round 4): 16 scalar instructions fewer per role and tile, and 1.3 % SLOWER per step together with the
                           // branch-free parent-row select below; each of the two alone changes nothing
    const unsigned t_in = (unsigned)(tile - hint * tiles_per_frame);
    if (t_in >= (unsigned)tiles_per_frame) hint = tile / tiles_per_frame;
    return hint;
#else
    (void)hint;
    return tile / tiles_per_frame;
#endif
}

// One v_max_f32.  fmaxf() on a value that comes straight out of an MFMA costs two: the compiler first canonicalises a
// possible signalling NaN with a v_max_f32 x, x, x of its own.
__device__ inline float max_raw(float x, float y) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
    return r;
}

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

// JK: JumpingKnowledge('max') of the reference (src/core/models.py:380-382, :479-482) carried through the fused stack as a
// running element-wise maximum: jk_out = max(jk_in, out) is written beside the layer output (the first layer passes its input
// as jk_in); with the classifier heads fused in, the heads run on max(jk_in, out) instead of out.
// DIAG: the handle has 'grid-diagonal' levels (8-neighbour grids); their segment pairs aggregate through segp_diag_rows, and
// the rare node-by-node path reads the handle's per-frame CSR (rowptr / colidx) instead of decoding the plain stencil.
template <bool CLS, bool JK = false, int MODE = 0, bool DIAG = false>
__global__ __launch_bounds__(PS_THREADS, 2) void k_gcn_layer_ps(const float* __restrict__ x, const float* __restrict__ W,
                                                                const float* __restrict__ scale, const float* __restrict__ shift,
                                                                float* __restrict__ out, const float* __restrict__ dis,
                                                                const Topo* __restrict__ T, const TileDesc* __restrict__ tiles,
                                                                const SegDesc* __restrict__ segs, const float* __restrict__ pats,
                                                                const float* __restrict__ patsq, const float* __restrict__ kin, float* __restrict__ kout,
                                                                const float* __restrict__ jk_in, float* __restrict__ jk_out,
                                                                int* __restrict__ counters, const PsDims a, const ClsArgs ca,
                                                                const int* __restrict__ rowptr, const int* __restrict__ colidx) {
    constexpr bool TRAIN = MODE == 1;                 // train forward: aggregated rows kept, BatchNorm sums, static walk
    constexpr bool RSEP = MODE >= 2;                  // the residual is a tensor of its own (a.res: the backward's dX = (A_hat dz) W + dy)
    constexpr bool SUMS = MODE == 3;                  // ... and the rows written are the lower layer's dy: its BatchNorm-backward sums leave per tile
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* s_a0 = smem;                               // [2][TILE * LDA]  aggregated tiles
    float* s_x0 = smem + 2 * TILE * LDA;              // [2][TILE * LDA]  raw self rows (residual); then the output tile
    int* s_tile = reinterpret_cast<int*>(smem + 4 * TILE * LDA);                  // [8] ring of tile ids
    int* s_sync = s_tile + 8;                         // consumer-only tile counter (CLS)
    int* s_cd = s_tile + 16;                          // [2][8 segments][4] descriptor words of the consumers' epilogue (below)
    float* s_dis0 = smem + PS_LDS_DIS;                // [2][TILE] (deg+1)^-1/2 of the tile's rows (child sums of the output)
    float* s_pat = smem + PS_LDS_PAT;                 // [n_pats][64] weight patterns, quad layout (seg_wide.h)
    float* s_bn = s_pat + a.n_pats * PATQ;            // CLS only: [4][128] layer scale, shift, classifier s1, t1

    const int tid = threadIdx.x;
    const int lane_k = tid & 63;
    const int wave = wave_id();
    const bool consumer = wave < 4;
    const int n_tiles = a.tiles_per_frame * a.batch;
    const int group = xcc_id();

    // ---- prologue: tiles 0, 1 and 2 claimed (before anyone reads the ring) -------------------------------------
    if (tid == 256) {
        if (TRAIN && a.static_walk) {
            for (int i = 0; i < 3; ++i) s_tile[i] = ps_static_tile(i, n_tiles);
        } else {
            ps_claim(counters, group, n_tiles, &s_tile[0]);
            ps_claim(counters, group, n_tiles, &s_tile[1]);
            ps_claim(counters, group, n_tiles, &s_tile[2]);
        }
    }
    for (int i = tid; i < a.n_pats * PATQ; i += PS_THREADS) s_pat[i] = patsq[i];
    if (TRAIN) {    // rows of a tile that hold no node are masked out of the statistics by a factor 0: they must be finite
        for (int i = tid; i < 4 * TILE * LDA; i += PS_THREADS) smem[i] = 0.f;
    }
    if (CLS) {
        if (tid < C) {
            s_bn[tid] = scale ? scale[tid] : 1.0f;
            s_bn[C + tid] = shift ? shift[tid] : 0.0f;
            s_bn[2 * C + tid] = ca.s1[tid];
            s_bn[3 * C + tid] = ca.t1[tid];
        }
        if (tid == 0) *s_sync = 0;
    }
    __syncthreads();

    // The two roles run separate loops (so that neither carries the other's persistent registers); both execute
    // exactly one workgroup barrier per tile, in lock step:   [prologue barrier]  (tile k work)  [barrier k] ...
    if (wave < 4) {
        // =========================== CONSUMER: channels 32*wave .. 32*wave+31 ===================================
        constexpr bool RES_LATE = CLS;                 // (the fused-classifier variant has no registers for the residual prefetch)
        float wreg[64];
        load_w_slice(W, wave, lane_k, a.transpose_w, wreg);
        // The per-channel scale of the epilogue (eval-mode BatchNorm folded by the caller) goes into the W slice -- lane
        // (i = l & 31) holds output channel 32 wave + i -- and the shift into the accumulators' initial value: the epilogue
        // is max + residual add, with no LDS or register operand of its own.
        if (scale) {
            const float sv = scale[32 * wave + (lane_k & 31)];
#pragma unroll
            for (int t = 0; t < 64; ++t) wreg[t] *= sv;
        }
        f32x16 shv;                                    // !CLS: shift of channel (r & 3) + 8 (r >> 2) + 4 h in register r
        if (!CLS) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int ch0 = 32 * wave + 8 * g + 4 * (lane_k >> 5);
                const f32x4 q = shift ? *reinterpret_cast<const f32x4*>(shift + ch0) : f32x4{0.f, 0.f, 0.f, 0.f};
                shv[4 * g] = q.x; shv[4 * g + 1] = q.y; shv[4 * g + 2] = q.z; shv[4 * g + 3] = q.w;
            }
        }
        // CLS: wave = classifier head.  First-layer slice of the stacked [128,128] weight in 64 more VGPRs (the per-channel
        // scale / shift vectors of both stages then live in LDS, not in registers); second layer as in classifier.hip:
        // MFMA 16x16x4 A operand lane (o = l & 15, kq = l >> 4) holds W2[head][o][8 kq + s].
        float wreg2[64];
        float w2a[8];
        f32x4 t2v, w3v;
        float b3v = 0.f;
        if (CLS) {
            load_w_slice(ca.w1, wave, lane_k, 0, wreg2);
            {
                const float sv = ca.s1[32 * wave + (lane_k & 31)];      // the first head layer's BatchNorm scale, folded the same way
#pragma unroll
                for (int t = 0; t < 64; ++t) wreg2[t] *= sv;
            }
            const f32x4* pw = reinterpret_cast<const f32x4*>(ca.w2 + (size_t)(wave * 16 + (lane_k & 15)) * 32 + 8 * (lane_k >> 4));
            const f32x4 q0 = pw[0], q1 = pw[1];
            w2a[0] = q0.x; w2a[1] = q0.y; w2a[2] = q0.z; w2a[3] = q0.w; w2a[4] = q1.x; w2a[5] = q1.y; w2a[6] = q1.z; w2a[7] = q1.w;
            {
                const float sv = ca.s2[wave * 16 + (lane_k & 15)];         // second BatchNorm's scale into its rows of W2, the shift t2
#pragma unroll
                for (int t = 0; t < 8; ++t) w2a[t] *= sv;                  // into the accumulators' initial value (below)
            }
            const int o4 = wave * 16 + 4 * (lane_k >> 4);
            t2v = *reinterpret_cast<const f32x4*>(ca.t2 + o4);
            w3v = *reinterpret_cast<const f32x4*>(ca.w3 + o4);
            b3v = ca.b3[wave];
        }
        // Descriptor words the epilogue needs -- {n_first, cnt, par0, pad1} of the tile's 8 segments: first node and count,
        // parent row and parent count of the segment pair -- come through an LDS ring the producers fill from the descriptors
        // they hold anyway (slot = buffer parity).  The consumers issue NO vector-memory load: a load whose result is needed
        // at the top of the next tile made the compiler drain vmcnt there, i.e. wait for every output store of the tile before.
        // TRAIN: column sums of the output and its square, per lane (= per row slot j and 16 channels) across the workgroup's
        // tiles; reduced over the row slots once, after the last tile
        f32x16 csum, csq;
#pragma unroll
        for (int i = 0; i < 16; ++i) { csum[i] = 0.f; csq[i] = 0.f; }
        // SUMS: this lane's 4 channels in the store phase are 32 wave + 4 (lane & 7) + {0..3} for every row it touches
        f32x4 lsc = {0.f, 0.f, 0.f, 0.f}, lsh = lsc;
        unsigned long long lseed = 0;
        if (SUMS) {
            lsc = *reinterpret_cast<const f32x4*>(a.lower.scale + 32 * wave + 4 * (lane_k & 7));
            lsh = *reinterpret_cast<const f32x4*>(a.lower.shift + 32 * wave + 4 * (lane_k & 7));
            lseed = a.lower.seed + epoch_now(a.lower.epoch);
        }
        __syncthreads();                                   // tile 0 is in buffer 0
        int frame_hint = 0;
        PSTAMP_INIT;
#ifdef EG_STAMP
        // in-kernel clock (MI355X_MICROARCH.md, DVFS give-back item 6): shader cycles / 100 MHz ticks around the tile loop
        const unsigned long long clk0 = __builtin_amdgcn_s_memtime(), rt0 = __builtin_amdgcn_s_memrealtime();
#endif
        for (int k = 0;; ++k) {
            const int t_cur = __builtin_amdgcn_readfirstlane(s_tile[k & 7]);
            if (t_cur < 0) break;
            int lane = lane_k;
            asm volatile("" : "+v"(lane));
            const int cd = s_cd[(k & 1) * 32 + (lane & 31)];
            PSTAMP(3);
            EG_PROBE(4);
            float* s_a = s_a0 + (k & 1) * TILE * LDA;
            float* s_x = s_x0 + (k & 1) * TILE * LDA;
#ifdef EG_ABL_HOT          // timing-only ablation: every frame's tiles read and write FRAME 0 (36.9 MB in, as much out: served by L2 / the
            const int frame = 0;   // memory-side cache, no HBM traffic to speak of); the instruction streams of both roles are unchanged
            (void)frame_hint;
#else
            const int frame = frame_of(t_cur, a.tiles_per_frame, frame_hint);
#endif
            int seg_first[8], seg_cnt[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                seg_first[i] = __builtin_amdgcn_readlane(cd, 4 * i);
                seg_cnt[i] = __builtin_amdgcn_readlane(cd, 4 * i + 1);
            }
            // SUMS: the lower layer's z rows of the tile, in the store phase's layout (8 lanes per row), issued here -- a whole tile of
            // matrix work in front of their use, inside ONE iteration (the compiler counts the stores issued behind them: no drain)
            f32x4 zr[8];
            f32x4 ls1 = {0.f, 0.f, 0.f, 0.f}, lt1 = ls1;
            if (SUMS) {
                const __amdgpu_buffer_rsrc_t zrsrc = __builtin_amdgcn_make_buffer_rsrc(
                    const_cast<float*>(a.lower.z) + (size_t)frame * a.n_per_frame * C, 0, a.n_per_frame * (C * 4), 0x00020000);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int last = seg_cnt[i] - 1;
                    const int u = (lane >> 3) < last ? (lane >> 3) : last;        // (absent segment: u = -1, out of range, reads 0)
                    zr[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                        zrsrc, u * (C * 4) + (32 * wave + 4 * (lane & 7)) * 4, seg_first[i] * (C * 4), 0));
                }
            }
            // Child sums of the OUTPUT for the next layer (kout): lane -> (parent q = (lane >> 3) + 8 i, 16-B chunk lane & 7);
            // the parent's four children are rows 2 pr, 2 pr + 1, columns 2 pc, 2 pc + 1 of this patch.
            int kout_row[2];
            if (kout) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const bool odd = (lane >> 5) != 0;                  // pr = 2 i + odd: segments 4 i (even pr) / 4 i + 2
                    const int pc = (lane >> 3) & 3;
                    // all four words are read first and selected per lane afterwards: a readlane inside the conditional made
                    // the compiler branch on `odd` (two exec-masked paths per select, ~70 instructions per tile)
                    const int npar_e = __builtin_amdgcn_readlane(cd, 4 * (4 * i) + 3), npar_o = __builtin_amdgcn_readlane(cd, 4 * (4 * i + 2) + 3);
                    const int par0_e = __builtin_amdgcn_readlane(cd, 4 * (4 * i) + 2), par0_o = __builtin_amdgcn_readlane(cd, 4 * (4 * i + 2) + 2);
#ifdef EG_OLD_KOUT_ROW
                    const int npar = odd ? __builtin_amdgcn_readlane(cd, 4 * (4 * i + 2) + 3) : __builtin_amdgcn_readlane(cd, 4 * (4 * i) + 3);
                    const int par0 = odd ? __builtin_amdgcn_readlane(cd, 4 * (4 * i + 2) + 2) : __builtin_amdgcn_readlane(cd, 4 * (4 * i) + 2);
                    (void)npar_e; (void)npar_o; (void)par0_e; (void)par0_o;
#else
                    const int npar = odd ? npar_o : npar_e;
                    const int par0 = odd ? par0_o : par0_e;
#endif
                    kout_row[i] = pc < npar ? par0 + pc : -1;
                }
            }
            float rowmask[2] = {1.f, 1.f};              // TRAIN: 1 where row slot j of the 32-row block holds a node
            if (TRAIN) {
#pragma unroll
                for (int rb = 0; rb < 2; ++rb) {
                    const int q = (lane >> 3) & 3;         // patch row 4 rb + q, column lane & 7
                    const int cnt = q == 0 ? seg_cnt[4 * rb] : (q == 1 ? seg_cnt[4 * rb + 1] : (q == 2 ? seg_cnt[4 * rb + 2] : seg_cnt[4 * rb + 3]));
                    rowmask[rb] = (lane & 7) < cnt ? 1.f : 0.f;
                }
            }
            // child sums leave as buffer stores too: no parent in this lane's slot (or no child-sum output at all) = offset -1
            const __amdgpu_buffer_rsrc_t krsrc = __builtin_amdgcn_make_buffer_rsrc(
                kout ? kout + (size_t)frame * a.kid_rows * C : nullptr, 0, kout ? a.kid_rows * (C * 4) : 0, 0x00020000);
            int kvoff[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) kvoff[i] = (kout && kout_row[i] >= 0) ? kout_row[i] * (C * 4) + (32 * wave + 4 * (lane & 7)) * 4 : -1;
            auto kout_half = [&](int i) {                 // parents of patch rows 4i .. 4i+3 (LDS rows 32i .. 32i+31), branch-free
                const int q = (lane >> 3) + 8 * i;
                const int ra = 16 * (q >> 2) + 2 * (q & 3);            // LDS row of child (2 pr, 2 pc)
                const float* sp = s_x + ra * LDA + 32 * wave + 4 * (lane & 7);
                const float* dp = s_dis0 + (k & 1) * TILE + ra;
                f32x4 ks = dp[0] * *reinterpret_cast<const f32x4*>(sp);
                ks += dp[1] * *reinterpret_cast<const f32x4*>(sp + LDA);
                ks += dp[8] * *reinterpret_cast<const f32x4*>(sp + 8 * LDA);
                ks += dp[9] * *reinterpret_cast<const f32x4*>(sp + 9 * LDA);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4v, ks), krsrc, kvoff[i], 0, 0);
                store_data_guard(ks);
            };
            f32x16 acc0, acc1;
            if (CLS) {                                  // (no registers for a persistent copy: the shift comes from LDS each tile)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 q = *reinterpret_cast<const f32x4*>(s_bn + C + 32 * wave + 4 * (lane >> 5) + 8 * g);
                    acc0[4 * g] = q.x; acc0[4 * g + 1] = q.y; acc0[4 * g + 2] = q.z; acc0[4 * g + 3] = q.w;
                }
                acc1 = acc0;
            } else {
                acc0 = shv;
                acc1 = shv;
            }
            // Epilogue pieces, all branch-free so that they can sit in one scheduling region with MFMAs.
            // Lane (row j of a 32-row block, half h) holds 16 channels of ONE row: stored from there a wave instruction
            // would touch 32 rows x 32 B.  The finished values go back into this wave's own channel slice of the stash
            // instead (where the residual was read from; no other wave touches that slice), are re-read 8 lanes per row,
            // and leave as whole 128-B line segments: 8 stores per wave and tile.
            const int j = lane & 31, h = lane >> 5;
            const float relu_floor = a.relu ? 0.f : -__builtin_inff();
            const bool has_res = a.has_res != 0;
            // residual rows of both 32-row blocks, read BEFORE the MFMA chains: an LDS wait inside a chain stalls the
            // wave's next MFMA as well (in-order issue)
            f32x4 res[2][4];
            if (RES_LATE) {                                                   // one register set: rows 0..31 now, rows 32..63 once those are done
#pragma unroll
                for (int g = 0; g < 4; ++g) res[0][g] = *reinterpret_cast<const f32x4*>(s_x + j * LDA + 32 * wave + 4 * h + 8 * g);
            } else {
                if (has_res) {                                                // a real (uniform) branch: 32 selects per tile otherwise
                    asm volatile("" ::: "memory");
#pragma unroll
                    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                        for (int g = 0; g < 4; ++g)
                            res[rb][g] = *reinterpret_cast<const f32x4*>(s_x + (32 * rb + j) * LDA + 32 * wave + 4 * h + 8 * g);
                } else {
                    asm volatile("" ::: "memory");
#pragma unroll
                    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                        for (int g = 0; g < 4; ++g) res[rb][g] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
            auto finish_group = [&](const f32x16& acc, int rb, int g) {          // 4 channels of row 32 rb + j -> LDS
                float* xp = s_x + (32 * rb + j) * LDA + 32 * wave + 4 * h + 8 * g;   // LDS row = 8 * patch row + column
                f32x4 v = f32x4{acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};          // scale and shift are in already
                v.x = max_raw(v.x, relu_floor); v.y = max_raw(v.y, relu_floor); v.z = max_raw(v.z, relu_floor); v.w = max_raw(v.w, relu_floor);
                if (RES_LATE) {
                    const f32x4 r = res[0][g];
                    v.x += has_res ? r.x : 0.f; v.y += has_res ? r.y : 0.f; v.z += has_res ? r.z : 0.f; v.w += has_res ? r.w : 0.f;
                } else {
                    v += res[rb][g];
                }
                if (TRAIN) {                               // rows that hold no node (duplicates of a valid row) do not count
                    const f32x4 t = v * rowmask[rb];
                    csum[4 * g] += t.x; csum[4 * g + 1] += t.y; csum[4 * g + 2] += t.z; csum[4 * g + 3] += t.w;
                    csq[4 * g] += t.x * v.x; csq[4 * g + 1] += t.y * v.y; csq[4 * g + 2] += t.z * v.z; csq[4 * g + 3] += t.w * v.w;
                }
                *reinterpret_cast<f32x4*>(xp) = v;
            };
            const int u8 = lane >> 3, c4 = 4 * (lane & 7);
            // output rows leave as buffer stores: frame descriptor + scalar offset of the segment's first row; the lane part
            // (node inside the segment, channel chunk) is the only vector arithmetic of a store
            const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(
                CLS ? nullptr : out + (size_t)frame * a.n_per_frame * C, 0, CLS ? 0 : a.n_per_frame * (C * 4), 0x00020000);
            const int ocol = (32 * wave + c4) * 4;
            f32x4 o[4], jm[4];
            int node[4], ovoff[4], osoff[4];
            const float* jb = JK ? jk_in + (size_t)frame * a.n_per_frame * C + 32 * wave + c4 : nullptr;
            float* jo = JK ? jk_out + (size_t)frame * a.n_per_frame * C + 32 * wave + c4 : nullptr;
            auto read_segments = [&](int i0) {                                   // patch rows i0 .. i0+3: 8 lanes per row
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (JK) {
                        // an absent segment (ragged patch) repeats segment 0, a short one its last node: identical stores
                        const int i = seg_cnt[i0 + e] > 0 ? i0 + e : 0;
                        const int cnt = seg_cnt[i0 + e] > 0 ? seg_cnt[i0 + e] : seg_cnt[0];
                        const int first = seg_cnt[i0 + e] > 0 ? seg_first[i0 + e] : seg_first[0];
                        const int u = u8 < cnt ? u8 : cnt - 1;
                        o[e] = *reinterpret_cast<const f32x4*>(s_x + (8 * i + u) * LDA + 32 * wave + c4);
                        node[e] = first + u;
                        ovoff[e] = u * (C * 4) + ocol;
                        osoff[e] = first * (C * 4);
                        jm[e] = *reinterpret_cast<const f32x4*>(jb + (size_t)node[e] * C);
                    } else {
                        // a short segment repeats its last node (identical stores); an absent one (ragged patch, cnt = 0) gets
                        // u = -1: a negative buffer offset is out of range and the hardware drops the store
                        const int last = seg_cnt[i0 + e] - 1;
                        const int u = u8 < last ? u8 : last;
                        o[e] = *reinterpret_cast<const f32x4*>(s_x + 8 * (i0 + e) * LDA + __mul24(u, LDA) + 32 * wave + c4);
                        ovoff[e] = u * (C * 4) + ocol;
                        osoff[e] = seg_first[i0 + e] * (C * 4);
                    }
                }
            };
            auto lower_sums = [&](int i0) {                 // SUMS: rows of segments i0 .. i0+3 as they sit in o[] (dy of the lower layer)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int node = seg_first[i0 + e] + u8;
                    const bool live = u8 < seg_cnt[i0 + e] && node < a.lower.row_hi;      // not a repeated / absent row, not a coordinate node
                    f32x4 g = o[e];
                    if (a.lower.p > 0.f)
                        g *= keep_scale4(lseed, ((unsigned long long)frame * a.n_per_frame + node) * C + 32 * wave + c4, a.lower.p, a.lower.inv_keep);
                    const f32x4 zz = zr[i0 + e];
                    const f32x4 v = zz * lsc + lsh;
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const float ge = (live && (!a.lower.relu || v[u] > 0.f)) ? g[u] : 0.f;
                        ls1[u] += ge;
                        lt1[u] += ge * zz[u];
                    }
                }
            };
            auto store_segments = [&]() {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4v, o[e]), orsrc, ovoff[e], osoff[e], 0);
                    if (JK) {
                        const f32x4 m = {fmaxf(jm[e].x, o[e].x), fmaxf(jm[e].y, o[e].y), fmaxf(jm[e].z, o[e].z), fmaxf(jm[e].w, o[e].w)};
                        *reinterpret_cast<f32x4*>(jo + (size_t)node[e] * C) = m;
                    }
                }
                store_data_guard(o);                   // (the stores' data registers: seg_wide.h)
            };
#ifndef EG_ABL_NO_MFMA
            // rows 32..63: the MFMA chain leaves ~60 issue cycles per instruction free; the epilogue of rows 0..31
            // (VALU, LDS, and its global stores) is placed between the chunks of the chain, every LDS read one chunk
            // ahead of its use
            auto between = [&](int c) {
                if (c < 2) { finish_group(acc0, 0, 2 * c); finish_group(acc0, 0, 2 * c + 1); }
                else if (!CLS && c == 2) read_segments(0);
                else if (!CLS) { store_segments(); if (SUMS) lower_sums(0); kout_half(0); }
                else if (c == 2) {                      // CLS: the residual rows of the second block, into the registers the first one has left
#pragma unroll
                    for (int g = 0; g < 4; ++g) res[0][g] = *reinterpret_cast<const f32x4*>(s_x + (32 + j) * LDA + 32 * wave + 4 * h + 8 * g);
                }
            };
#if defined(EG_EPI_MID)      // experiment: block 0's whole epilogue as ONE piece between the two chains, chain 2 uninterrupted
            mfma_rowblock(s_a, 0, lane, wreg, acc0);
            asm volatile("" : "+v"(acc0));
            __builtin_amdgcn_sched_barrier(0);
            if (!CLS) {
#pragma unroll
                for (int g = 0; g < 4; ++g) finish_group(acc0, 0, g);
                read_segments(0); store_segments(); if (SUMS) lower_sums(0); kout_half(0);
            } else {
                between(0); between(1); between(2);
            }
            __builtin_amdgcn_sched_barrier(0);
            mfma_rowblock(s_a, 32, lane, wreg, acc1);
            asm volatile("" : "+v"(acc1));
            __builtin_amdgcn_sched_barrier(0);
#else
            mfma_rowblock(s_a, 0, lane, wreg, acc0);
            mfma_rowblock_with(s_a, 32, lane, wreg, acc1, between);
#endif
#else
            acc0[0] += wreg[0] + s_a[lane]; acc1[0] += wreg[63];
#pragma unroll
            for (int g = 0; g < 4; ++g) finish_group(acc0, 0, g);
            if (!CLS) { read_segments(0); store_segments(); if (SUMS) lower_sums(0); kout_half(0); }
#endif
            PSTAMP(0);
#ifdef EG_STAMP3                  // finer consumer stamp: the first EG_STAMP3 groups of the exposed epilogue count as "loop"
#pragma unroll
            for (int g = 0; g < EG_STAMP3; ++g) finish_group(acc1, 1, g);
            PSTAMP(3);
#pragma unroll
            for (int g = EG_STAMP3; g < 4; ++g) finish_group(acc1, 1, g);
#else
#pragma unroll
            for (int g = 0; g < 4; ++g) finish_group(acc1, 1, g);
#endif
            if (!CLS) {
                read_segments(4);
                store_segments();
                if (SUMS) {
                    lower_sums(4);
                    // over the 8 rows a wave instruction covers (lane bits 3..5), fixed order; lanes 0..7 hold the tile's sums of
                    // channels 32 wave + 4 lane .. + 3
#pragma unroll
                    for (int m = 8; m < 64; m <<= 1) {
#pragma unroll
                        for (int u = 0; u < 4; ++u) { ls1[u] += __shfl_xor(ls1[u], m); lt1[u] += __shfl_xor(lt1[u], m); }
                    }
                    if (lane < 8) {
                        float* tp = a.lower.tile_partial + (size_t)t_cur * 2 * C + 32 * wave + 4 * lane;
                        *reinterpret_cast<f32x4*>(tp) = ls1;
                        *reinterpret_cast<f32x4*>(tp + C) = lt1;
                    }
                }
                kout_half(1);
            } else {
                // ---- classifier heads on the finished tile (all 128 channels of it are needed: the four consumer waves
                // meet on an LDS counter; the producers are not involved and keep filling the other buffer) -------------
                PSTAMP(1);
                if (JK) {
                    // this wave's 32 channels of all 64 rows, 8 lanes per row: output tile <- max(jk_in, output tile)
                    const float* jb = jk_in + (size_t)frame * a.n_per_frame * C + 32 * wave + c4;
#pragma unroll
                    for (int i0 = 0; i0 < 8; i0 += 4) {
                        f32x4 jm[4];
                        int row[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            // an absent segment repeats segment 0, a short one its last node (rows without a node are never used)
                            const int i = seg_cnt[i0 + e] > 0 ? i0 + e : 0;
                            const int cnt = seg_cnt[i0 + e] > 0 ? seg_cnt[i0 + e] : seg_cnt[0];
                            const int first = seg_cnt[i0 + e] > 0 ? seg_first[i0 + e] : seg_first[0];
                            const int u = u8 < cnt ? u8 : cnt - 1;
                            row[e] = 8 * i + u;
                            jm[e] = *reinterpret_cast<const f32x4*>(jb + (size_t)(first + u) * C);
                        }
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float* xp = s_x + row[e] * LDA + 32 * wave + c4;
                            const f32x4 v = *reinterpret_cast<const f32x4*>(xp);
                            *reinterpret_cast<f32x4*>(xp) = f32x4{fmaxf(jm[e].x, v.x), fmaxf(jm[e].y, v.y), fmaxf(jm[e].z, v.z), fmaxf(jm[e].w, v.w)};
                        }
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                if (lane == 0) __hip_atomic_fetch_add(s_sync, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                const int target = 4 * (k + 1);
                while (__hip_atomic_load(s_sync, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < target) __builtin_amdgcn_s_sleep(1);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                PSTAMP(2);                                 // (stamp builds: the consumers' wait for each other counts as barrier time)
                // first layers: hidden[row][32 wave + c] = relu(bn(h3[row][:] . W1[32 wave + c][:])), into this wave's column
                // slice of the A tile (dead now: every wave is past its MFMAs on it)
                f32x16& hc0 = acc0;
                f32x16& hc1 = acc1;
#pragma unroll
                for (int g = 0; g < 4; ++g) {                      // start from the BatchNorm shift t1 (the scale s1 is in wreg2)
                    const f32x4 q = *reinterpret_cast<const f32x4*>(s_bn + 3 * C + 32 * wave + 4 * h + 8 * g);
                    hc0[4 * g] = q.x; hc0[4 * g + 1] = q.y; hc0[4 * g + 2] = q.z; hc0[4 * g + 3] = q.w;
                }
                hc1 = hc0;
                auto hidden_group = [&](const f32x16& hc, int rb, int g) {
                    f32x4 v = f32x4{hc[4 * g], hc[4 * g + 1], hc[4 * g + 2], hc[4 * g + 3]};
                    v.x = max_raw(v.x, 0.f); v.y = max_raw(v.y, 0.f); v.z = max_raw(v.z, 0.f); v.w = max_raw(v.w, 0.f);
                    *reinterpret_cast<f32x4*>(s_a + (32 * rb + j) * LDA + 32 * wave + 4 * h + 8 * g) = v;
                };
                mfma_rowblock(s_x, 0, lane, wreg2, hc0);
                mfma_rowblock_with(s_x, 32, lane, wreg2, hc1, [&](int c) { hidden_group(hc0, 0, c); });
                PSTAMP(3);                                 // (stamp builds: the second GEMM counts as "loop")
                const int j16 = lane & 15, kq = lane >> 4;
                f32x4 hb[4][2];
                auto load_hidden = [&](int b4) {
                    const f32x4* hp = reinterpret_cast<const f32x4*>(s_a + (16 * b4 + j16) * LDA + 32 * wave + 8 * kq);
                    hb[b4][0] = hp[0]; hb[b4][1] = hp[1];
                };
                load_hidden(0); load_hidden(1);                       // rows 0..31 (ahead of the LDS writes below)
#pragma unroll
                for (int g = 0; g < 4; ++g) hidden_group(hc1, 1, g);
                load_hidden(2); load_hidden(3);
                // second / third layers per 16-row block: K = 32 on the MFMA (8 x 16x16x4), BN + ReLU + the 16-wide dot
                // on the accumulator (4 outputs per lane, two cross-lane adds), as in classifier.hip
                // logits leave as buffer stores off a frame descriptor: lane (kq = 0, j16) owns node (patch row 2 b4 + (j16 >> 3),
                // column j16 & 7); every other lane, and columns past the segment's count, get offset -1 (out of range: dropped)
                // (the node-type filter: a frame's logits start at its row row_lo; rows in front of it get a negative offset, rows
                // behind the range one past the descriptor's end -- both out of range for the unsigned bounds check: dropped)
                const __amdgpu_buffer_rsrc_t lrsrc = __builtin_amdgcn_make_buffer_rsrc(
                    ca.logits + (size_t)frame * ca.n_valid * 4, 0, ca.n_valid * 16, 0x00020000);
                int lvoff[4];
#pragma unroll
                for (int b4 = 0; b4 < 4; ++b4) {
                    const bool hi = (j16 >> 3) != 0;
                    const int first = hi ? seg_first[2 * b4 + 1] : seg_first[2 * b4];
                    const int cnt = hi ? seg_cnt[2 * b4 + 1] : seg_cnt[2 * b4];
                    lvoff[b4] = (kq == 0 && (j16 & 7) < cnt) ? (first + (j16 & 7) - ca.row_lo) * 16 + wave * 4 : -1;
                }
                f32x4v z[4];
#pragma unroll
                for (int b4 = 0; b4 < 4; ++b4) z[b4] = f32x4v{t2v.x, t2v.y, t2v.z, t2v.w};
                // rows 0..31 first: their hidden values were written inside the chain above, so these MFMAs run while the
                // writes of rows 32..63 and the reads behind them are in flight
#pragma unroll
                for (int t = 0; t < 8; ++t)
#pragma unroll
                    for (int b4 = 0; b4 < 2; ++b4)
                        z[b4] = __builtin_amdgcn_mfma_f32_16x16x4f32(w2a[t], hb[b4][t >> 2][t & 3], z[b4], 0, 0, 0);
#pragma unroll
                for (int t = 0; t < 8; ++t)
#pragma unroll
                    for (int b4 = 2; b4 < 4; ++b4)
                        z[b4] = __builtin_amdgcn_mfma_f32_16x16x4f32(w2a[t], hb[b4][t >> 2][t & 3], z[b4], 0, 0, 0);
                float y[4];
#pragma unroll
                for (int b4 = 0; b4 < 4; ++b4) {
                    float v = w3v.x * max_raw(z[b4].x, 0.f);
                    v += w3v.y * max_raw(z[b4].y, 0.f);
                    v += w3v.z * max_raw(z[b4].z, 0.f);
                    v += w3v.w * max_raw(z[b4].w, 0.f);
                    // sum over the four lane quarters (kq): two register swaps, no LDS round trip
                    const auto r16 = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
                    v = __uint_as_float(r16[0]) + __uint_as_float(r16[1]);
                    const auto r32 = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
                    y[b4] = __uint_as_float(r32[0]) + __uint_as_float(r32[1]) + b3v;
                }
                if (ca.sigmoid) {
#pragma unroll
                    for (int b4 = 0; b4 < 4; ++b4) y[b4] = 1.0f / (1.0f + __expf(-y[b4]));
                }
#pragma unroll
                for (int b4 = 0; b4 < 4; ++b4) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(y[b4]), lrsrc, lvoff[b4], 0, 0);
            }
            EG_PROBE(5);
            PSTAMP(1);
            __syncthreads();                               // barrier k+1: buffer (k+1)&1 is full, buffer k&1 is free
            PSTAMP(2);
        }
        if (TRAIN) {
            __syncthreads();                               // (with the producers) every tile buffer is dead
            float* st = smem + (wave * 64 + lane_k) * 32;  // [wave][lane][sum 16 | sum of squares 16], 32 KB over the tile buffers
#pragma unroll
            for (int i = 0; i < 16; ++i) { st[i] = csum[i]; st[16 + i] = csq[i]; }
            __syncthreads();
            // fixed order over the 32 row slots: thread t -> (quantity t >> 7, channel t & 127 = 32 wave + 8 g + 4 h + e)
            const int t = tid, q = t >> 7, c = t & 127, w = c >> 5, g = (c >> 3) & 3, hh = (c >> 2) & 1, e = c & 3;
            float sacc = 0.f;
#pragma unroll 4
            for (int jj = 0; jj < 32; ++jj) sacc += smem[((w * 64 + hh * 32 + jj) * 32) + q * 16 + 4 * g + e];
            a.stats_partial[(size_t)blockIdx.x * 2 * C + t] = sacc;
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
        // =========================== PRODUCER: patch rows 2p, 2p+1 of every tile ================================
        const int p = wave - 4;
#ifndef EG_PS_NO_PRIO
        __builtin_amdgcn_s_setprio(3);
#endif
        PSTAMP_INIT;
        // The two segment descriptors of a tile (32 dwords) travel in ONE VGPR, lane l holding dword l & 31: they are
        // fetched a whole tile ahead (a scalar load at the point of use costs a loaded-memory round trip, ~9k cycles
        // measured, in front of the row loads) and turned into SGPRs with v_readlane when the tile is produced.
        int desc_hint = 0, prod_hint = 0;
        auto load_desc = [&](int tile_i, int lane) -> int {
            const int frame = frame_of(tile_i, a.tiles_per_frame, desc_hint);
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
        auto produce = [&](int tile_i, int buf, int lane, int dv) {
#ifdef EG_ABL_HOT
            const int frame = 0;
            (void)prod_hint;
#else
            const int frame = frame_of(tile_i, a.tiles_per_frame, prod_hint);
#endif
            const float* __restrict__ xf = x + (size_t)frame * a.n_per_frame * C;
            const SegDesc sd0 = desc_of(dv, 0);
            const SegDesc sd1 = desc_of(dv, 16);
            {   // the consumers' descriptor words of this tile (segments 2p, 2p+1: words 0, 1, 6, 15 -> ring slot [seg][0..3])
                const int w = lane & 15;
                const int f = w == 0 ? 0 : (w == 1 ? 1 : (w == 6 ? 2 : (w == 15 ? 3 : -1)));
                if (lane < 32 && f >= 0) s_cd[buf * 32 + (2 * p + (lane >> 4)) * 4 + f] = dv;
            }
            float* s_a = s_a0 + buf * TILE * LDA;
            float* s_x = a.has_res ? s_x0 + buf * TILE * LDA : nullptr;
            f32x4 acc0[4], acc1[4];
            if (sd0.pad0) {
#ifdef EG_STAMP2              // finer producer stamps: 0 = descriptor wait, 1 = load issue, 2 = fma + kids + store, 3 = claim + barrier
                PSTAMP(0);
#define PS_ISSUE 1
#define PS_MAIN 2
#else
#define PS_ISSUE 0
#define PS_MAIN 1
#endif
                // ---- usual case: both segments on the fast path, vertically adjacent, same parents ----
                SegPair A;
                const bool use_kin = (sd0.aux & 1) && kin;                // uniform: children already summed by the previous layer
                const RowSrc xs = row_src(xf, a.n_per_frame * (C * 4), lane);
                f32x4 Ra[4], Rb[4];                                  // RSEP: the residual's own rows of both segments
                if (RSEP) {
                    const RowSrc rs = row_src(a.res + (size_t)frame * a.n_per_frame * C, a.n_per_frame * (C * 4), lane);
#pragma unroll
                    for (int k = 0; k < 4; ++k) { Ra[k] = ldp(rs, sd0.n_first, k); Rb[k] = ldp(rs, sd1.n_first, k); }
                }
                segp_issue(sd0, sd1, xs, A);
                SegPairDiag DE;
                const bool diag = DIAG && (sd0.aux & SEG_AUX_DIAG);          // uniform
                if (diag) segp_diag_issue(sd0, sd1, xs, a.n_per_frame - 1, DE);
                const int hub = DIAG ? (sd0.aux >> 2) - 1 : -1;             // uniform: the connection node of this level, or none
                f32x4 HS = {0.f, 0.f, 0.f, 0.f};
                if (DIAG && hub >= 0)                                       // its d-scaled row, the same in both half-waves
                    HS = ld_two_rows(row_src(a.conn + (size_t)frame * a.conn_stride + a.conn_scaled, a.n_conn * (C * 4), lane), hub, hub);
                // the child-sum rows travel with the first batch: issued after the main stage they cost the aux tiles a second,
                // fully exposed memory round trip (the producers do most of their work after the consumers' MFMA chain has ended,
                // DESIGN 5.22)
                SegKidsum KS;
                if (use_kin) segp_kidsum_issue(sd0, sd1, row_src(kin + (size_t)frame * a.kid_rows * C, a.kid_rows * (C * 4), lane), KS);
                // unchained calls (a stack's first layer): the first segment's 16 child rows as well; the second segment's
                // follow once the main stage has freed its registers
                SegKids K0;
                if (!RSEP && !DIAG && (sd0.aux & 1) && !use_kin) segw_kids_issue(sd0, pats, xs, lane, K0);      // (RSEP, DIAG: no registers left for them here)
#ifdef EG_ABL_NO_LOADS        // timing-only ablation: producers do nothing (results wrong)
                return;
#endif
                __builtin_amdgcn_sched_barrier(0);                  // every load of both segments is issued above this line
                PSTAMP(PS_ISSUE);
                EG_PROBE(1);
                const float* wqa = s_pat + sd0.pat * PATQ + 32 * (lane >> 5);      // this lane's weights (LDS, quad layout)
                const float* wqb = s_pat + sd1.pat * PATQ + 32 * (lane >> 5);
                float* s_t = (s_x && !RSEP) ? s_x : s_a;            // where the segments' own rows pass through LDS
                if (RSEP) {                                         // first back (first issued): hand the residual rows to the consumers
                    const PairLane pl{lane >> 5, lane & 31};
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        *reinterpret_cast<f32x4*>(&s_x[(16 * p + 2 * k + pl.h) * LDA + 4 * pl.q]) = Ra[k];
                        *reinterpret_cast<f32x4*>(&s_x[(16 * p + 8 + 2 * k + pl.h) * LDA + 4 * pl.q]) = Rb[k];
                    }
                }
                if (diag) {
                    if (s_x && !RSEP) {                             // the raw self rows for the consumers' residual
                        const PairLane pl{lane >> 5, lane & 31};
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            *reinterpret_cast<f32x4*>(&s_x[(16 * p + 2 * k + pl.h) * LDA + 4 * pl.q]) = A.Sa[k];
                            *reinterpret_cast<f32x4*>(&s_x[(16 * p + 8 + 2 * k + pl.h) * LDA + 4 * pl.q]) = A.Sb[k];
                        }
                    }
                    segp_diag_rows(lane, wqa, wqb, A, DE, acc0, acc1, s_a + 16 * p * LDA);
                } else {
                    segw_rows(lane, wqa, A.Sa, A.LRa, A.U, A.Sb, A.P, acc0, s_t, 16 * p);
                    segw_rows(lane, wqb, A.Sb, A.LRb, A.Sa, A.D, A.P, acc1, s_t, 16 * p + 8);
                }
                if (DIAG && hub >= 0) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) { acc0[k] += HS; acc1[k] += HS; }
                }
                pin_acc4(acc0);
                pin_acc4(acc1);                                     // the 40 main-stage registers are dead from here on
                __builtin_amdgcn_sched_barrier(0);
                PSTAMP(PS_MAIN);
                EG_PROBE(2);
                if (use_kin) {
                    segp_kidsum_add(KS, wqa, wqb, acc0, acc1);
                } else if (sd0.aux & 1) {                           // uniform: aux level, children pulled as rows
                    if (RSEP || DIAG) segw_kids_issue(sd0, pats, xs, lane, K0);
                    segw_kids_add(lane, K0, acc0);
                    pin_acc4(acc0);                                 // (issuing the second segment's loads ahead of this add measured slower)
                    {
                        SegKids K;
                        segw_kids_issue(sd1, pats, xs, lane, K);
                        segw_kids_add(lane, K, acc1);
                    }
                }
                segw_store(lane, wqa, acc0, s_a, 16 * p);
                segw_store(lane, wqb, acc1, s_a, 16 * p + 8);
                if (TRAIN && a.agg_out) {
                    const __amdgpu_buffer_rsrc_t agg = __builtin_amdgcn_make_buffer_rsrc(
                        a.agg_out + (size_t)frame * a.n_per_frame * C, 0, a.n_per_frame * (C * 4), 0x00020000);
                    segw_store_agg(lane, wqa, acc0, agg, sd0.n_first, sd0.cnt);
                    segw_store_agg(lane, wqb, acc1, agg, sd1.n_first, sd1.cnt);
                }
                if (kout && (lane & 31) == 0) {                     // (deg+1)^-1/2 of the 16 nodes, for the consumers' child sums
                    const f32x4 da = quad_w(wqa, SLOT_SELF), db = quad_w(wqb, SLOT_SELF);
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        s_dis0[buf * TILE + 16 * p + 2 * k + (lane >> 5)] = da[k];
                        s_dis0[buf * TILE + 16 * p + 8 + 2 * k + (lane >> 5)] = db[k];
                    }
                }
            } else {
                // ---- ragged patches, coordinate nodes, frame end (rare): node by node, scalar neighbour decode ----
#pragma unroll 1
                for (int e = 0; e < 2; ++e) {
                    const int n0 = e ? sd1.n_first : sd0.n_first, cnt = e ? sd1.cnt : sd0.cnt;
                    const bool given = DIAG && (e ? sd1.mode : sd0.mode) == 3;      // connection nodes: rows aggregated by the pre-pass
                    const int rl = 16 * p + 8 * e;
#pragma unroll 1
                    for (int u = 0; u < cnt; ++u) {
                        const f32x2 av = given ? load_row2(a.conn + (size_t)frame * a.conn_stride + a.conn_agg, n0 + u, lane)
                                               : (DIAG ? agg_csr(xf, dis, rowptr, colidx, n0 + u, lane) : agg_stencil(T, xf, dis, n0 + u, lane));
                        *reinterpret_cast<f32x2*>(&s_a[(rl + u) * LDA + 2 * lane]) = av;
                        if (TRAIN && a.agg_out) *reinterpret_cast<f32x2*>(a.agg_out + ((size_t)frame * a.n_per_frame + n0 + u) * C + 2 * lane) = av;
                        if (s_x) *reinterpret_cast<f32x2*>(&s_x[(rl + u) * LDA + 2 * lane]) =
                            load_row2(RSEP ? a.res + (size_t)frame * a.n_per_frame * C : xf, n0 + u, lane);
                    }
                }
            }
            EG_PROBE(3);
            PSTAMP(2);
        };
        int dv_next = 0;
        {
            const int t0 = __builtin_amdgcn_readfirstlane(s_tile[0]);
            const int t1 = __builtin_amdgcn_readfirstlane(s_tile[1]);
            if (t0 >= 0) produce(t0, 0, lane_k, load_desc(t0, lane_k));
            if (t1 >= 0) dv_next = load_desc(t1, lane_k);
        }
        __syncthreads();                                   // tile 0 is in buffer 0
        for (int k = 0;; ++k) {
            const int t_cur = __builtin_amdgcn_readfirstlane(s_tile[k & 7]);
            if (t_cur < 0) break;
            const int t_next = __builtin_amdgcn_readfirstlane(s_tile[(k + 1) & 7]);
            const int t_nn = __builtin_amdgcn_readfirstlane(s_tile[(k + 2) & 7]);
            int lane = lane_k;
            asm volatile("" : "+v"(lane));
            const int dv_cur = dv_next;
            if (t_nn >= 0) dv_next = load_desc(t_nn, lane);                                 // used by the NEXT iteration
            int got = 0;
            const bool walk_static = TRAIN && a.static_walk;
            if (tid == 256 && !walk_static) got = ps_claim_issue(counters, group);          // three tiles ahead, asynchronous
            PSTAMP(3);
            if (t_next >= 0) produce(t_next, (k + 1) & 1, lane, dv_cur);
            if (tid == 256) {
                if (walk_static) s_tile[(k + 3) & 7] = ps_static_tile(k + 3, n_tiles);
                else ps_claim_commit(counters, group, n_tiles, got, &s_tile[(k + 3) & 7]);
            }
            __syncthreads();                               // barrier k+1
            PSTAMP(3);
        }
        if (TRAIN) {                                       // the consumers' hand-over of the statistics (two barriers)
            __syncthreads();
            __syncthreads();
        }
        PSTAMP_FLUSH(4);
#ifndef EG_STAMP
        // The slice of the queue ring is left the way it was found -- zeroed -- by the LAST workgroup out (every claim of this
        // workgroup has returned by now: each was committed inside the loop), so that a launch needs no memset node in front of
        // it (4.7 us each, three per inference step).  Device-scope atomics only: the counters never sit dirty in an XCD's L2.
        if (tid == 256 && !(TRAIN && a.static_walk) && a.self_reset) {
            const int done = atomicAdd(&counters[QUEUE_DONE_IDX], 1);
            if (done == (int)gridDim.x - 1) {
#pragma unroll
                for (int q = 0; q < WALK_GROUPS; ++q) atomicExch(&counters[q * WALK_CTR_STRIDE], 0);
                atomicExch(&counters[QUEUE_DONE_IDX], 0);
            }
        }
#endif
    }
}

}  // namespace eg

using namespace eg;

int eg_launch_conn_prepass(const eg_graph* g, int batch, const float* x, int slot, hipStream_t stream, const float** base);

// Used by eg_gcn_layer_fwd (gcn_layer.hip) for topology handles when the residual is NULL or x itself.
// Returns EG_ERR_UNSUPPORTED when the caller should fall back to the symmetric kernel.
int eg_launch_layer_ps(const eg_graph* g, int batch, const float* x, const float* W, const float* scale,
                       const float* shift, const float* residual, int relu, int transpose_w, float* out,
                       const float* kin, float* kout, const eg::ClsArgs* cls, hipStream_t stream, const float* jk_in, float* jk_out,
                       float* agg_out, float* stats_partial, int* grid_out, const eg::LowerSums* lower) {
    if (!g || g->kind != GRAPH_TOPO) return EG_ERR_UNSUPPORTED;
    if (lower && !(residual != nullptr && residual != x)) return set_error(EG_ERR_ARG, "the lower layer's sums go with the dX launch (a residual tensor of its own)");
    if (lower && (!lower->z || !lower->scale || !lower->shift || !lower->tile_partial || lower->row_hi < 1))
        return set_error(EG_ERR_ARG, "incomplete LowerSums");
    const bool train = stats_partial != nullptr;
    const bool rsep = residual != nullptr && residual != x;       // a residual tensor of its own: MODE 2, plain calls only
    if (rsep && (train || kin || kout || cls || jk_in || jk_out)) return EG_ERR_UNSUPPORTED;
    if (agg_out && !train) return set_error(EG_ERR_ARG, "agg_out goes with stats_partial (train forward)");
    if (train && (kout || cls || jk_in || jk_out)) return set_error(EG_ERR_ARG, "the train forward writes no child sums (its activation pass does) and takes no heads or running maximum");
    const bool jk = jk_in != nullptr;
    if ((jk_out != nullptr) != (jk && !cls)) return set_error(EG_ERR_ARG, "jk_out goes with jk_in on a plain layer, the fused heads take jk_in alone");
    const bool chained = kin || kout || jk;
    if ((kin || kout) && g->kid_rows == 0) return EG_ERR_UNSUPPORTED;
    if (jk && g->kid_rows == 0 && !g->flat) return EG_ERR_UNSUPPORTED;
    if (cls && g->kid_rows == 0 && !g->flat) return EG_ERR_UNSUPPORTED;
    // plain calls: this kernel by default wherever its fast paths cover the topology -- single-level grids and regular pyramids
    // (child sums available); since the producers were rebuilt (round 3) its unchained form, which pulls the four child rows of
    // an aux node, is faster than the symmetric kernel as well (0.280 vs 0.309 ms per launch at configs[1], round 5) -- and the
    // symmetric kernel on irregular frames; EG_LAYER_IMPL = 0 / 1 forces one of them
    const bool diag = g->hybrid != 0;                             // 'grid-diagonal' levels: this kernel is the stencil path of such handles
    if (diag && jk) return EG_ERR_UNSUPPORTED;                    // (the running maximum has no diagonal instantiation: the caller takes it outside)
    if (!chained && !cls && !train && !rsep && !diag &&
        (g->knobs.layer_impl < 0 ? (g->flat || g->kid_rows > 0) : g->knobs.layer_impl != 0) == false) return EG_ERR_UNSUPPORTED;
    PsDims a{};
    a.n_per_frame = (int)g->n_nodes; a.batch = batch; a.tiles_per_frame = g->n_tiles;
    a.relu = relu; a.transpose_w = transpose_w; a.has_res = residual != nullptr;
    a.kid_rows = g->kid_rows; a.n_pats = g->n_pats;
    a.agg_out = agg_out; a.stats_partial = stats_partial; a.static_walk = train ? 1 : 0;
    a.res = rsep ? residual : nullptr;
    if (lower) a.lower = *lower;
    const long long n_tiles = (long long)a.tiles_per_frame * batch;
    if (n_tiles <= 0) return EG_OK;
    const size_t lds = (size_t)(PS_LDS_PAT + g->n_pats * PATQ + (cls ? 4 * C : 0)) * sizeof(float);      // (graph.hip checks the same sum)
    if (lds > 160 * 1024) return EG_ERR_UNSUPPORTED;             // more weight patterns than fit beside the tile buffers
    {   // 160 KB of dynamic LDS needs the attribute once per device (idempotent, so a benign race sets it twice at worst)
        static std::atomic<bool> attr_set[64];
        int dev = 0;
        EG_HIP_TRY(hipGetDevice(&dev));
        if (dev < 0 || dev >= 64 || !attr_set[dev].load(std::memory_order_acquire)) {
            const void* kernels[] = {(const void*)k_gcn_layer_ps<false, false>, (const void*)k_gcn_layer_ps<true, false>,
                                     (const void*)k_gcn_layer_ps<false, true>, (const void*)k_gcn_layer_ps<true, true>,
                                     (const void*)k_gcn_layer_ps<false, false, 1>, (const void*)k_gcn_layer_ps<false, false, 2>,
                                     (const void*)k_gcn_layer_ps<false, false, 0, true>, (const void*)k_gcn_layer_ps<true, false, 0, true>,
                                     (const void*)k_gcn_layer_ps<false, false, 1, true>, (const void*)k_gcn_layer_ps<false, false, 2, true>,
                                     (const void*)k_gcn_layer_ps<false, false, 3>, (const void*)k_gcn_layer_ps<false, false, 3, true>};
            for (const void* f : kernels) EG_HIP_TRY(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            if (dev >= 0 && dev < 64) attr_set[dev].store(true, std::memory_order_release);
        }
    }
    int* queue = nullptr;
    int slot = -1;
    {
        const int rc = g->acquire_queue_slice(stream, &queue, &slot);
        if (rc != EG_OK) return rc == EG_ERR_UNSUPPORTED ? EG_ERR_RING : rc;
    }
    // every queue-walking kernel zeroes its slice on the way out (see this kernel's last lines and k_gcn_layer's): a slice is clean
    // whenever it is handed out -- a device invariant, so nothing here depends on host state that a HIP-graph replay would not see
    // EG_QUEUE_SELF_RESET=0 (diagnostic) puts a memset in front of every EAGER launch in addition; the kernel resets its slice all
    // the same, and a launch that is being captured never records a memset node: a graph of memset + kernel nodes captured after
    // the handle had launched on another stream replayed WITHOUT running its tiles (round 5, `tools/gpu_job.sh` knob matrix:
    // the GPU suite under every run-time knob), so replays must not depend on one
#ifdef EG_STAMP
    const bool self_reset = false;
    EG_HIP_TRY(hipMemsetAsync(queue, 0, sizeof(int) * QUEUE_SLICE_INTS, stream));
#else
    const bool self_reset = true;
    if (g->knobs.queue_self_reset == 0) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(stream, &cs) != hipSuccess) { (void)hipGetLastError(); cs = hipStreamCaptureStatusNone; }
        if (cs == hipStreamCaptureStatusNone) EG_HIP_TRY(hipMemsetAsync(queue, 0, sizeof(int) * QUEUE_SLICE_INTS, stream));
    }
#endif
    a.self_reset = self_reset ? 1 : 0;
    if (g->n_conn > 0) {                                          // connection nodes: level sums of THIS launch's input first (conn.hip)
        const float* slice = nullptr;
        const int rc = eg_launch_conn_prepass(g, batch, x, slot, stream, &slice);
        if (rc != EG_OK) return rc;                               // (EG_ERR_UNSUPPORTED: more frames than the scratch holds -> the caller's other kernel)
        a.conn = slice;
        a.conn_stride = (g->conn_chunks + 2 * g->n_conn) * C;
        a.conn_agg = g->conn_chunks * C;
        a.conn_scaled = (g->conn_chunks + g->n_conn) * C;
        a.n_conn = g->n_conn;
    }
    long long grid = n_tiles < 256 ? n_tiles : g->knobs.ps_grid;      // one persistent workgroup per CU
    const ClsArgs none{};
    auto launch = [&](auto kernel) {
        hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(PS_THREADS), lds, stream, x, W, scale, shift, out, g->dis, g->topo_dev,
                           g->tiles_dev, g->segs_dev, g->pats_dev, g->patsq_dev, kin, kout, jk_in, jk_out, queue, a, cls ? *cls : none,
                           (const int*)g->rowptr, (const int*)g->colidx);
    };
    if (grid_out) *grid_out = (int)grid;
    eg::LaunchTimer timer(train ? EG_LAUNCH_PS_TRAIN_FWD : rsep ? EG_LAUNCH_PS_DX : cls ? EG_LAUNCH_PS_CLS : EG_LAUNCH_PS_PLAIN, stream);
    if (diag) {
        if (train) launch(k_gcn_layer_ps<false, false, 1, true>);
        else if (rsep && lower) launch(k_gcn_layer_ps<false, false, 3, true>);
        else if (rsep) launch(k_gcn_layer_ps<false, false, 2, true>);
        else if (cls) launch(k_gcn_layer_ps<true, false, 0, true>);
        else launch(k_gcn_layer_ps<false, false, 0, true>);
    } else if (train) launch(k_gcn_layer_ps<false, false, 1>);
    else if (rsep && lower) launch(k_gcn_layer_ps<false, false, 3>);
    else if (rsep) launch(k_gcn_layer_ps<false, false, 2>);
    else if (cls) { if (jk) launch(k_gcn_layer_ps<true, true>); else launch(k_gcn_layer_ps<true, false>); }
    else { if (jk) launch(k_gcn_layer_ps<false, true>); else launch(k_gcn_layer_ps<false, false>); }
    g->commit_queue_slice(slot, stream);
    g->ps_launches.fetch_add(1u, std::memory_order_relaxed);
    g->layer_launches.fetch_add(1u, std::memory_order_relaxed);
    EG_HIP_TRY(hipGetLastError());
    return EG_OK;
}
