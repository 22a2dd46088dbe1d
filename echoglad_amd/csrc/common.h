// Shared declarations of the gfx950 hot-path library (not part of the public ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/echoglad_hip.h"

namespace eg {

constexpr int C = EG_CHANNELS;     // 128 channels per node row (512 B)
constexpr int TILE = 64;           // node rows per workgroup tile
constexpr int LDA = C + 4;         // LDS row stride (floats): 132 -> ds_read_b128 of 16 rows is conflict-free
constexpr int MAX_LEVELS = 14;     // aux levels + main grid
constexpr int MAX_SLOTS = 10;      // self + 4 in-level + parent + 4 children

int set_error(int code, const std::string& msg);
#define EG_HIP_TRY(expr)                                                                         \
    do {                                                                                         \
        hipError_t _e = (expr);                                                                  \
        if (_e != hipSuccess)                                                                    \
            return ::eg::set_error(EG_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
    } while (0)

// Closed-form hierarchical topology.  Kernels read a device copy through a const
// pointer: every field is wave-uniform, so the accesses are scalar loads (a by-value
// kernel argument would be spilled to scratch by the dynamic level indexing).
// Per-level constants of the implicit stencil, 16 ints so one s_load_dwordx16 fetches a level.
//   parent of (r,c):   pbase + (poff + (r>>1))*pside + poff + (c>>1)      iff r < plim && c < plim
//   children of (r,c): cbase + 2(r-clo)*cside + 2(c-clo) (+1, +cside, +cside+1) iff clo <= r,c < chi
struct LevelDesc {
    int base, end, side, kind;        // node ids [base, end); kind: 0 aux, 1 main, 2 coordinate nodes
    int lg;                           // log2(side) for aux levels
    int pbase, pside, poff, plim;
    int cbase, cside, clo, chi;
    int pad0, pad1, pad2;
};

// One 8x8 (or smaller, at level edges) patch of one level: the unit of work of the fused layer kernel.
// LDS row rl = 8*tr + u  <->  node id  desc[level].base + (r0 + tr) * side + c0 + u,  tr < nrows, u < ncols.
struct TileDesc {
    int level, r0, c0, nrows, ncols, pad0, pad1, pad2;
};

enum { KIND_AUX = 0, KIND_MAIN = 1, KIND_COORD = 2, KIND_CONN = 3 };      // LevelDesc::kind (CONN: the connection nodes' pseudo-level)

// Precomputed (host, at handle creation) description of one 8-node segment = patch row tr of a patch:
// everything the aggregation needs that does not depend on the features.  One s_load_dwordx16 per segment.
struct SegDesc {
    int n_first, cnt, mode;      // first node id, nodes in the segment, 0 none / 1 fast (runs) / 2 per-node slow path /
                                 // 3 connection nodes: the aggregated rows come precomputed (conn.hip)
    int pat;                     // index into the weight-pattern table (128 floats per pattern: 64 wa + 64 wb)
    int up0, down0, par0;        // first row of the 8 rows above / below, of the 4 parent rows (clamped into the frame)
    int left, right;             // the two edge rows (clamped)
    int c0, c1, c2, c3;          // child runs: rows 2r cols 0-7 / 8-15 of the segment's children, rows 2r+1 likewise (aux)
    int aux;                     // bit 0: aux level (children exist as slots); bit 1: 'grid-diagonal' level; bits 2..: 1 + the
                                 // connection node wired to every node of this level (0: none)
    int pad0;                    // patch row 2p only: rows 2p and 2p+1 can be aggregated as a pair (shared rows)
    int pad1;                    // patch row 2p only: number of parents whose 4 children are columns 2j, 2j+1 of rows 2p, 2p+1
};

struct Topo {
    int n_desc;              // aux levels + main (+ coordinate pseudo-level)
    int pad_[3];
    LevelDesc desc[MAX_LEVELS + 1];
    int n_nodes;             // nodes per frame, incl. coordinate nodes
    int n_levels;            // aux levels + 1
    int n_aux;               // 0 when use_main_graph_only
    int frame;               // F
    int n_conn;              // connection nodes at the head of every frame (naux + 1 with use_connection_nodes, else 0): node g - 1
                             // is wired to every node of aux level g, g = 1 .. naux - 1, and to the other connection nodes
    int diag_main, diag_aux; // 'grid-diagonal' main grid / aux levels (8-neighbour grids, datasets.py:1469-1475, :1494-1500)
    int crop0, ncrop;        // run of rows/cols of the last aux level wired to the main grid
    int coord_base;          // first coordinate node id (== n_nodes when there are none)
    int base[MAX_LEVELS];    // first node id of each level (aux 1..naux, then main)
    int side[MAX_LEVELS];    // grid side of each level
    unsigned long long magic;// floor(2^40 / F) + 1 : idx / F == (idx * magic) >> 40 for idx < 2^24
};

// One node's neighbourhood: ids[0] is the node itself; slots beyond the real
// neighbours repeat the node with valid=0 so the caller can issue all loads
// unconditionally.  Follows reference src/core/datasets.py:1441-1584.
struct Nbrs {
    int id[MAX_SLOTS];
    int valid[MAX_SLOTS];
    int count;      // slots to visit: 6 for main-grid and coordinate nodes, 10 for aux nodes
    int degree;     // real neighbours, excluding self
};

__host__ __device__ inline int level_of(const Topo& T, int n) {
    int l = 0;
#pragma unroll 1
    for (int k = 1; k < T.n_levels; ++k) l += (n >= T.base[k]) ? 1 : 0;
    return l;
}

__host__ __device__ inline void neighbours(const Topo& T, int n, Nbrs& nb) {
#pragma unroll
    for (int s = 0; s < MAX_SLOTS; ++s) { nb.id[s] = n; nb.valid[s] = 0; }
    nb.valid[0] = 1;
    if (n >= T.coord_base) {                          // isolated K4 (datasets.py:1517-1523)
#pragma unroll
        for (int j = 0; j < 4; ++j) {                 // slots 1..4 = the four coordinate nodes, self masked out
            const int o = T.coord_base + j;
            nb.id[1 + j] = o;
            nb.valid[1 + j] = (o != n);
        }
        nb.count = 6; nb.degree = 3;
        return;
    }
    const int l = level_of(T, n);
    const int side = T.side[l];
    const int idx = n - T.base[l];
    const bool is_main = (l == T.n_levels - 1);
    int r, c;
    if (is_main) {
        r = (int)(((unsigned long long)(unsigned)idx * T.magic) >> 40);
        c = idx - r * side;
    } else {                                          // aux sides are powers of two
        const int lg = l + 1;
        r = idx >> lg;
        c = idx & (side - 1);
    }
    int deg = 0;
    // in-level 4-neighbourhood (nx.grid_graph, datasets.py:1465,1490)
    nb.valid[1] = r > 0;            nb.id[1] = nb.valid[1] ? n - side : n;
    nb.valid[2] = r < side - 1;     nb.id[2] = nb.valid[2] ? n + side : n;
    nb.valid[3] = c > 0;            nb.id[3] = nb.valid[3] ? n - 1 : n;
    nb.valid[4] = c < side - 1;     nb.id[4] = nb.valid[4] ? n + 1 : n;
    deg += nb.valid[1] + nb.valid[2] + nb.valid[3] + nb.valid[4];
    // parent
    if (is_main) {
        if (T.n_aux > 0 && r < 2 * T.ncrop && c < 2 * T.ncrop) {       // datasets.py:1558-1584
            const int pl = T.n_levels - 2;
            nb.id[5] = T.base[pl] + (T.crop0 + (r >> 1)) * T.side[pl] + T.crop0 + (c >> 1);
            nb.valid[5] = 1; ++deg;
        }
        nb.count = 6;
    } else {
        if (l > 0) {                                                   // datasets.py:1534-1556
            nb.id[5] = T.base[l - 1] + (r >> 1) * T.side[l - 1] + (c >> 1);
            nb.valid[5] = 1; ++deg;
        }
        // children
        const int cl = l + 1;
        int cr = -1, cc = -1;
        if (cl < T.n_levels - 1) { cr = 2 * r; cc = 2 * c; }
        else if (r >= T.crop0 && r < T.crop0 + T.ncrop && c >= T.crop0 && c < T.crop0 + T.ncrop) {
            cr = 2 * (r - T.crop0); cc = 2 * (c - T.crop0);
        }
        if (cr >= 0) {
            const int cs = T.side[cl];
            const int b = T.base[cl] + cr * cs + cc;
            nb.id[6] = b;          nb.id[7] = b + 1;
            nb.id[8] = b + cs;     nb.id[9] = b + cs + 1;
            nb.valid[6] = nb.valid[7] = nb.valid[8] = nb.valid[9] = 1;
            deg += 4;
        }
        nb.count = 10;
    }
    nb.degree = deg;
}

enum GraphKind { GRAPH_TOPO = 1, GRAPH_CSR = 2 };

// Experiment knobs, read from the environment ONCE (per handle at creation; process-wide for handle-less entry points),
// never on a launch path.
struct Knobs {
    int walk_mode;      // EG_WALK_MODE   tile walk of the symmetric kernel (tile.h WALK_*), default queue
    int stagger;        // EG_STAGGER     experiment: second half of the grid starts late
    int grid_cap;       // EG_GRID        persistent grid of the symmetric kernel (default 512 = 2 workgroups per CU)
    int layer_impl;     // EG_LAYER_IMPL  -1 auto, 0 symmetric kernel, 1 producer/consumer kernel for plain calls
    int ps_grid;        // EG_PS_GRID     persistent grid of the producer/consumer kernel (default 256 = 1 per CU)
    int ring_guard;     // EG_RING_GUARD  0: no event behind a launch (diagnostic: the queue ring is then unguarded, as before round 4)
    int csr_tiles;      // EG_CSR_TILES   CSR handles: 2 (default) clustered 64-node tiles with an LDS row stash, 1 the same stash over
                        //                consecutive rows (no clustering), 0 the plain row-by-row aggregator
    int queue_self_reset;   // EG_QUEUE_SELF_RESET  the layer kernels always zero their queue slice on the way out; 0 (diagnostic): a memset in front
                            //                      of every eager launch as well (never recorded into a captured graph)
};
Knobs read_knobs();
const Knobs& process_knobs();

// Tile-queue heads: 8 per-XCD counters, one 128-B line each = 256 ints per launch.  A handle owns a RING of such slices;
// every launch takes the next slice (host atomic) and hands it to its kernel, so launches on different streams that share a handle
// never touch the same counters.  DEVICE INVARIANT: a slice is all zero whenever no launch is using it -- the ring is zeroed at
// creation and every kernel that walks a queue (both layer kernels) zeroes its slice on the way out (last workgroup out, atomics
// only).  No host-side state says whether a slice is clean, so a launch recorded into a HIP graph and replayed later sees exactly
// what an eager launch sees.
constexpr int QUEUE_SLICE_INTS = 8 * 32;
constexpr int QUEUE_DONE_IDX = 1;       // (inside the first counter's 128-B line) workgroups that have left a self-resetting launch
#ifdef EG_STAMP
constexpr int QUEUE_SLOTS = 1;          // stamp builds keep their cycle sums right behind the (single) slice
#else
constexpr int QUEUE_SLOTS = 64;
#endif
constexpr int QUEUE_TAIL_INTS = 64;     // stamp statistics (diagnostic builds)
// internal return code of the launchers: the queue ring refused the launch (the message is set; the public entry points turn it
// into EG_ERR_UNSUPPORTED).  Distinct from EG_ERR_UNSUPPORTED, which between launchers means "use the other kernel".
constexpr int EG_ERR_RING = -100;
inline int public_rc(int rc) { return rc == EG_ERR_RING ? EG_ERR_UNSUPPORTED : rc; }

}  // namespace eg

// The opaque handle of the public ABI.
struct eg_graph {
    int kind;
    int64_t n_nodes;          // per frame (topo) / total (csr)
    eg::Topo topo;            // kind == GRAPH_TOPO (host copy)
    eg::Topo* topo_dev;       // device copy read by the kernels through scalar loads
    eg::TileDesc* tiles_dev;  // device [n_tiles] 2-D patch table of one frame (kind == GRAPH_TOPO)
    eg::SegDesc* segs_dev;    // device [n_tiles * 8] per-segment descriptors
    float* pats_dev;          // device [n_pats * 128] weight patterns (most segments share a handful)
    float* patsq_dev;         // device [n_pats * 64] the same patterns in quad layout (graph.hip)
    int n_pats;
    int kid_rows;             // rows per frame of the child-sum side buffer (= aux nodes), 0 when the topology does not qualify
    int hybrid;               // 1: a closed-form topology whose stencil only the producer/consumer kernel implements ('grid-diagonal'
                              //    levels, connection nodes): rowptr / colidx then hold the CSR of ONE frame for every other path
                              //    (symmetric kernel, eg_gcn_aggregate, the node-by-node path of ragged segments)
    // connection nodes (conn.hip): per-launch scratch, one slice per slot of the queue ring -- level sums in chunks, then per frame
    // the connection nodes' aggregated rows and their (deg + 1)^-1/2-scaled feature rows
    int n_conn;               // connection nodes per frame
    int conn_chunks;          // chunks of <= 256 rows the hub-wired levels are cut into (per frame)
    int* conn_table;          // device [conn_chunks][4] = {level index, first row, rows, 0}
    // device [QUEUE_SLOTS][conn_cap][conn_chunks + 2 n_conn][128]; a LARGER one is allocated (under conn_mutex) the first time a
    // launch brings more frames than it holds -- the one place a launch may allocate, once per new maximum.  The smaller ones are
    // never freed before eg_graph_destroy (conn_retired): a HIP graph captured at a smaller batch keeps replaying on the slice it
    // was given, and a launch on another thread may still be between taking its slice pointer and enqueueing its kernels.
    mutable float* conn_scratch;
    mutable int conn_cap;
    mutable std::mutex conn_mutex;
    mutable std::vector<float*> conn_retired;
    int flat;                 // 1: a single grid level (use_main_graph_only): no parents, no children; the producer/consumer
                              //    kernel needs no child sums there and is the default layer kernel when its tables fit LDS
    int n_tiles;
    float* dis;               // device [n_nodes]   (deg+1)^-1/2
    int* rowptr;              // device [n_nodes+1] kind == GRAPH_CSR
    int* colidx;              // device [nnz]
    int64_t nnz;
    // kind == GRAPH_CSR, inference launches of the layer kernel: the same CSR regrouped into TILES of 64 nodes that are close in
    // the graph (greedy breadth-first balls, graph.hip csr_tiles), so that most of a node's sources are rows of its own tile and
    // come out of the tile's LDS stash instead of being loaded once per edge
    int* t_rows;              // device [n_ctiles * 64]     node id of tile slot s (-1: padding at the end of the last tile)
    int* t_rowptr;            // device [n_ctiles * 64 + 1] edges of slot s: [t_rowptr[s], t_rowptr[s + 1])
    int* t_code;              // device [nnz]               source of the edge: -(slot + 1) inside the tile, else the node id
    int* t_tgt;               // device [nnz]               target of the edge: its slot modulo 8 (a wave of the layer kernel owns 8 slots)
    float* t_w;               // device [nnz]               the edge's weight d_src d_tgt, d = (deg + 1)^-1/2
    float* t_dis;             // device [n_ctiles * 64]     the self loop's weight d^2 of the slot's node (0: padding)
    int n_ctiles;
    int symmetric;            // kind == GRAPH_CSR: the kept edge multiset equals its transpose (A_hat^T == A_hat)
    int* walk_counters;       // device [QUEUE_SLOTS][8 x 32] ring of per-launch tile-queue heads (+ QUEUE_TAIL_INTS)
    mutable std::atomic<unsigned> launch_seq;   // next slice of the ring (the only host state a launch touches)
    mutable std::atomic<unsigned> ps_launches;  // launches of the producer/consumer kernel on this handle (eg_graph_ps_launches)
    mutable std::atomic<unsigned> layer_launches;   // launches of either layer kernel on this handle (eg_graph_layer_launches)
    eg::Knobs knobs;          // environment knobs, read once at creation

    // Guard of the ring: an event per slice, recorded behind the launch that used it, and the stream it was recorded on.
    // A launch that would reuse a slice whose previous user (on ANOTHER stream) has not finished is refused
    // (EG_ERR_UNSUPPORTED) instead of sharing live counters with it.
    hipEvent_t slot_event[eg::QUEUE_SLOTS];
    mutable std::atomic<void*> slot_stream[eg::QUEUE_SLOTS];     // nullptr-with-flag encoding: see graph.hip
    mutable std::atomic<unsigned char> slot_used[eg::QUEUE_SLOTS];   // 0 free / captured, 1 event recorded, 2 used without an event
    mutable std::atomic<void*> only_stream;      // the one stream this handle has launched on so far ...
    mutable std::atomic<unsigned char> multi_stream;   // ... until a second one shows up: from then on every launch records its event
    mutable std::atomic<unsigned char> any_launch;
    hipEvent_t era_event;                         // recorded on only_stream at the moment a second stream shows up: behind it lie all
    mutable std::atomic<unsigned char> era_recorded;   // launches that carry no event of their own (slot_used == 2)

    // the slice of the queue ring for one launch on `stream` (graph.hip); EG_OK / EG_ERR_UNSUPPORTED / EG_ERR_HIP
    int acquire_queue_slice(hipStream_t stream, int** slice, int* slot) const;
    // to be called right after the launch that uses the slice
    void commit_queue_slice(int slot, hipStream_t stream) const;
};

// eg_debug_layer_timing_* (graph.hip): a pair of events around a layer-kernel launch while a measurement is armed
namespace eg {
struct LaunchTimer {
    int idx;
    hipStream_t stream;
    LaunchTimer(int kind, hipStream_t s);     // records the start event (no-op unless armed and not capturing)
    ~LaunchTimer();                           // records the stop event
};
}  // namespace eg

// producer/consumer layer kernel (gcn_layer_ps.hip); EG_ERR_UNSUPPORTED -> caller uses the symmetric kernel
namespace eg {
// Classifier heads fused behind the LAST layer of a stack: node-type filter + 4 x [Linear(128,32)-BN-ReLU-Linear(32,16)-
// BN-ReLU-Linear(16,1)] of src/core/models.py:363-377, :485-490, eval-mode BN folded by the caller (same packing as
// eg_classifier_fwd).  The layer's output tile never leaves LDS.
struct ClsArgs {
    const float *w1, *s1, *t1, *w2, *s2, *t2, *w3, *b3;
    float* logits;
    int sigmoid;
    int row_lo, n_valid;        // the heads' node-type filter: rows [row_lo, row_lo + n_valid) of a frame have a logits row (logits is [batch * n_valid, 4])
};
}  // namespace eg
namespace eg {
// dX launch that ALSO takes the BatchNorm-backward sums of the layer BELOW (k_gcn_layer_ps MODE 3): its output rows are that
// layer's dy, so  sum g  and  sum g * z  (g = dy * dropout keep * ReLU gate of the lower layer's activation) are accumulated where
// the finished rows pass through the consumers' registers on their way out, one [2][128] float partial per TILE (fixed order
// inside a tile, tiles summed in index order afterwards: the same bits whoever wins a queue).  Rows >= row_hi of a frame (the
// coordinate nodes: the coordinate update's backward rewrites them afterwards) are left out.
struct LowerSums {
    const float* z;                 // the lower layer's pre-BatchNorm rows [batch * n, 128]
    const float* scale;             // its BatchNorm forward scale / shift (bn + 2 * 128, bn + 3 * 128 of eg_gcn_layer_train_fwd)
    const float* shift;
    int relu;
    float p, inv_keep;
    unsigned long long seed;
    const unsigned long long* epoch;
    int row_hi;
    float* tile_partial;            // [batch * tiles_per_frame][2][128]
};
}  // namespace eg
// symmetric 8-wave layer kernel (gcn_layer.hip) for any handle; agg_out (nullable) receives the aggregated rows A_hat x,
// stats_partial (nullable) per-workgroup column sums of out and out^2 as float [*grid_out][2][128]
int eg_launch_layer_sym(const eg_graph* g, int batch, const float* x, const float* W, const float* scale, const float* shift,
                        const float* residual, int relu, int transpose_w, float* out, float* agg_out, float* stats_partial,
                        int* grid_out, hipStream_t stream);
int eg_launch_layer_ps(const eg_graph* g, int batch, const float* x, const float* W, const float* scale,
                       const float* shift, const float* residual, int relu, int transpose_w, float* out,
                       const float* kin, float* kout, const eg::ClsArgs* cls, hipStream_t stream, const float* jk_in = nullptr,
                       float* jk_out = nullptr, float* agg_out = nullptr, float* stats_partial = nullptr, int* grid_out = nullptr,
                       const eg::LowerSums* lower = nullptr);
