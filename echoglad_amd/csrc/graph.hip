// Graph handles of the gfx950 hot-path library: closed-form topology, generic CSR, edge digest.
#include <hipcub/hipcub.hpp>

#include <cmath>
#include <algorithm>
#include <map>
#include <vector>

#include "common.h"
#include <cstring>

namespace eg {

static thread_local std::string g_last_error;

int set_error(int code, const std::string& msg) {
    g_last_error = msg;
    return code;
}

#define EG_NO_STREAM_YET ((void*)(intptr_t)-1)          // eg_graph::only_stream before the handle's first launch

#ifdef EG_DEBUG_TOPO
#define EG_DEBUG_TOPO_ON 1          // -DEG_DEBUG_TOPO: why a handle has no child-sum side buffer, on stderr
#else
#define EG_DEBUG_TOPO_ON 0
#endif

static int env_int(const char* name, int dflt) {
    const char* v = getenv(name);
    return v && *v ? atoi(v) : dflt;
}

Knobs read_knobs() {
    Knobs k;
    // Two RUN-TIME knobs are left here, each selecting a fallback that has to exist anyway (include/echoglad_hip.h lists every run-time
    // variable).  What rounds 2 - 5 tuned through the environment is a compile-time constant now (-D to experiment): the winners
    // are the defaults and nobody lands on a losing route by accident.
    k.layer_impl = env_int("EG_LAYER_IMPL", -1);
    k.csr_tiles = env_int("EG_CSR_TILES", 2);
#ifndef EG_WALK_MODE
#define EG_WALK_MODE 0
#endif
#ifndef EG_STAGGER
#define EG_STAGGER 0
#endif
#ifndef EG_GRID
#define EG_GRID 512
#endif
#ifndef EG_PS_GRID
#define EG_PS_GRID 256
#endif
#ifndef EG_RING_GUARD
#define EG_RING_GUARD 1
#endif
#ifndef EG_QUEUE_SELF_RESET
#define EG_QUEUE_SELF_RESET 1
#endif
    k.walk_mode = EG_WALK_MODE;
    k.stagger = EG_STAGGER;
    k.grid_cap = EG_GRID;
    k.ps_grid = EG_PS_GRID;
    k.ring_guard = EG_RING_GUARD;
    k.queue_self_reset = EG_QUEUE_SELF_RESET != 0;
    // the static tile walk of the train forward labels chunks with blockIdx % 8 (a grid below 8 would leave chunks without
    // an owner) and its statistics partials fill at most 2048 slabs of the workspace
    if (k.ps_grid < 8) k.ps_grid = 8;
    if (k.ps_grid > 2048) k.ps_grid = 2048;
    if (k.grid_cap < 8) k.grid_cap = 8;
    if (k.grid_cap > 2048) k.grid_cap = 2048;
    return k;
}

const Knobs& process_knobs() {
    static const Knobs k = read_knobs();      // thread-safe one-time initialisation
    return k;
}

constexpr size_t QUEUE_RING_BYTES = sizeof(int) * ((size_t)QUEUE_SLOTS * QUEUE_SLICE_INTS + QUEUE_TAIL_INTS);

}  // namespace eg

// ---- the tile-queue ring of a handle (common.h) ---------------------------------------------------------------------------
// Every launch takes the next slice.  Slice s was last used by launch k - 64; if that launch ran on the same stream it is
// ordered in front of this one, and if it ran on another stream its event (or, for launches from the time when the handle had
// seen a single stream and recorded none, a query of that whole stream) tells whether it has finished.  A slice whose last
// user is still in flight on another stream is NOT handed out: the call fails with EG_ERR_UNSUPPORTED (the caller serialises,
// or uses a handle per stream).  Launches recorded into a HIP graph (stream capture) carry no event: a captured launch keeps its
// slice for every replay, which is ordered with later eager launches only on the replaying stream itself (header note).
int eg_graph::acquire_queue_slice(hipStream_t stream, int** slice, int* slot) const {
    const unsigned s = launch_seq.fetch_add(1u, std::memory_order_relaxed) % (unsigned)eg::QUEUE_SLOTS;
    // a handle that has only ever launched on ONE stream needs no events (its launches are ordered); the moment a second stream
    // shows up every launch records one.  Slices last used before that moment carry none: their stream is queried as a whole.
    // (the first launch publishes its stream with ONE compare-and-swap from the "none yet" sentinel, and any_launch only afterwards:
    //  a second thread can no longer see any_launch set while only_stream still reads as the legacy default stream)
    {
        void* none = EG_NO_STREAM_YET;
        only_stream.compare_exchange_strong(none, (void*)stream, std::memory_order_acq_rel);
        any_launch.store(1, std::memory_order_release);
    }
    // A capturing stream queries nothing (a query of another stream is not a capturable call, and the answer would describe the
    // moment of the capture, not of a replay): a captured launch is ordered with the handle's other users by the CALLER (header).
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cap) != hipSuccess) { (void)hipGetLastError(); cap = hipStreamCaptureStatusNone; }
    if (any_launch.load(std::memory_order_acquire) && !multi_stream.load(std::memory_order_acquire) &&
        only_stream.load(std::memory_order_acquire) != (void*)stream) {
        // A second stream shows up.  The launches so far carry no events.  If their stream is idle NOW they are all done and their
        // slices are simply free; otherwise ONE event recorded on it now lies behind all of them.  (Querying that stream as a
        // whole at every later reuse, as before, answers "not ready" for the legacy default stream whenever torch has parked a
        // cross-stream wait on it: a train step warmed up on a side stream after eager steps on the default stream was refused.)
        hipStream_t old = (hipStream_t)only_stream.load(std::memory_order_acquire);
        if (cap == hipStreamCaptureStatusNone) {
            if (hipStreamQuery(old) == hipSuccess) {
                for (int i = 0; i < eg::QUEUE_SLOTS; ++i) {
                    unsigned char two = 2;
                    slot_used[i].compare_exchange_strong(two, 0, std::memory_order_acq_rel);
                }
            } else {
                (void)hipGetLastError();
                if (era_event && hipEventRecord(era_event, old) == hipSuccess) era_recorded.store(1, std::memory_order_release);
                else (void)hipGetLastError();
            }
        }
        multi_stream.store(1, std::memory_order_release);
    }
    const unsigned char used = slot_used[s].load(std::memory_order_acquire);
    if (cap == hipStreamCaptureStatusNone && used && slot_stream[s].load(std::memory_order_acquire) != (void*)stream) {
        const hipError_t q = used == 1 ? hipEventQuery(slot_event[s])
                             : (era_recorded.load(std::memory_order_acquire) ? hipEventQuery(era_event)
                                                                             : hipStreamQuery((hipStream_t)slot_stream[s].load(std::memory_order_acquire)));
        if (q == hipErrorNotReady) {
            (void)hipGetLastError();
            return eg::set_error(EG_ERR_UNSUPPORTED, "more than 64 launches of this graph handle are in flight on different streams: "
                                                     "the tile-queue slice of the launch 64 calls ago is still in use");
        }
        if (q != hipSuccess) { (void)hipGetLastError(); }      // (a destroyed stream, an event never recorded: nothing in flight)
    }
    *slice = walk_counters + (size_t)s * eg::QUEUE_SLICE_INTS;
    *slot = (int)s;
    return EG_OK;
}

void eg_graph::commit_queue_slice(int slot, hipStream_t stream) const {
    if (slot < 0 || slot >= eg::QUEUE_SLOTS) return;
    if (!knobs.ring_guard) return;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cs) != hipSuccess) { (void)hipGetLastError(); cs = hipStreamCaptureStatusNone; }
    if (cs != hipStreamCaptureStatusNone) { slot_used[slot].store(0, std::memory_order_release); return; }
    slot_stream[slot].store((void*)stream, std::memory_order_release);
    if (!multi_stream.load(std::memory_order_acquire)) { slot_used[slot].store(2, std::memory_order_release); return; }
    if (hipEventRecord(slot_event[slot], stream) != hipSuccess) { (void)hipGetLastError(); slot_used[slot].store(2, std::memory_order_release); return; }
    slot_used[slot].store(1, std::memory_order_release);
}

namespace eg {

static int create_slot_events(eg_graph* g) {
    for (int i = 0; i < QUEUE_SLOTS; ++i)
        if (hipEventCreateWithFlags(&g->slot_event[i], hipEventDisableTiming) != hipSuccess) return EG_ERR_HIP;
    if (hipEventCreateWithFlags(&g->era_event, hipEventDisableTiming) != hipSuccess) return EG_ERR_HIP;
    return EG_OK;
}

// Python floor division
static inline int floordiv(int a, int b) {
    int q = a / b;
    if ((a % b != 0) && ((a < 0) != (b < 0))) --q;
    return q;
}

// Python `range(len)[start:stop]` with step 1 -> [lo, hi)
static inline void py_slice(int len, int start, int stop, int& lo, int& hi) {
    lo = start < 0 ? (len + start < 0 ? 0 : len + start) : (start > len ? len : start);
    hi = stop < 0 ? (len + stop < 0 ? 0 : len + stop) : (stop > len ? len : stop);
    if (hi < lo) hi = lo;
}

static int build_topo(int frame, int naux, int main_only, int coord_nodes, int conn_nodes, int diag_main, int diag_aux, Topo& T) {
    if (frame < 2 || frame > 4096) return set_error(EG_ERR_ARG, "frame must be in [2, 4096]");
    if (!main_only && (naux < 1 || naux + 1 > MAX_LEVELS)) return set_error(EG_ERR_ARG, "naux out of range");
    T = Topo{};
    T.n_aux = main_only ? 0 : naux;
    T.n_conn = (!main_only && conn_nodes) ? naux + 1 : 0;     // datasets.py:1450-1456: only inside `if not use_main_graph_only`
    int nid = T.n_conn;
    T.diag_main = diag_main ? 1 : 0;
    T.diag_aux = (!main_only && diag_aux) ? 1 : 0;
    for (int k = 1; k <= T.n_aux; ++k) {
        T.base[k - 1] = nid;
        T.side[k - 1] = 1 << k;
        nid += (1 << k) * (1 << k);
    }
    T.n_levels = T.n_aux + 1;
    T.base[T.n_levels - 1] = nid;
    T.side[T.n_levels - 1] = frame;
    nid += frame * frame;
    T.coord_base = nid;
    if (!main_only && coord_nodes) nid += 4;       // datasets.py:1508-1523: only inside `if not use_main_graph_only`
    T.n_nodes = nid;
    T.frame = frame;
    T.magic = ((1ull << 40) / (unsigned long long)frame) + 1ull;
    if ((long long)frame * frame >= (1ll << 24)) return set_error(EG_ERR_UNSUPPORTED, "frame too large");
    T.crop0 = 0; T.ncrop = 0;
    if (T.n_aux > 0) {                             // datasets.py:1565-1567, Python slice semantics
        const int p = 1 << T.n_aux;
        const int half = frame / 2;
        const int c0 = floordiv(p - half, 2);
        int lo, hi;
        py_slice(p, c0, c0 + half, lo, hi);
        T.crop0 = lo; T.ncrop = hi - lo;
    }
    // per-level descriptors for the run-based stencil
    T.n_desc = T.n_levels + (T.coord_base < T.n_nodes ? 1 : 0);
    for (int l = 0; l < T.n_levels; ++l) {
        LevelDesc& d = T.desc[l];
        d = LevelDesc{};
        const bool is_main = (l == T.n_levels - 1);
        d.base = T.base[l];
        d.side = T.side[l];
        d.end = is_main ? T.coord_base : T.base[l + 1];
        d.kind = is_main ? 1 : 0;
        d.lg = is_main ? 0 : l + 1;
        if (is_main) {
            if (T.n_aux > 0) { d.pbase = T.base[l - 1]; d.pside = T.side[l - 1]; d.poff = T.crop0; d.plim = 2 * T.ncrop; }
        } else {
            if (l > 0) { d.pbase = T.base[l - 1]; d.pside = T.side[l - 1]; d.poff = 0; d.plim = d.side; }
            d.cbase = T.base[l + 1]; d.cside = T.side[l + 1];
            if (l + 1 < T.n_levels - 1) { d.clo = 0; d.chi = d.side; }
            else { d.clo = T.crop0; d.chi = T.crop0 + T.ncrop; }
        }
    }
    if (T.n_desc > T.n_levels) {
        LevelDesc& d = T.desc[T.n_levels];
        d = LevelDesc{};
        d.base = T.coord_base; d.end = T.n_nodes; d.side = 4; d.kind = 2;
    }
    if (T.n_conn > 0) {                            // the connection nodes' pseudo-level, 8 nodes per (pseudo) row
        if (T.n_desc + 1 > MAX_LEVELS + 1) return set_error(EG_ERR_ARG, "too many levels");
        LevelDesc& d = T.desc[T.n_desc];
        d = LevelDesc{};
        d.base = 0; d.end = T.n_conn; d.side = 8; d.kind = KIND_CONN;
        T.n_desc += 1;
    }
    return EG_OK;
}

// ---------------------------------------------------------------- CSR build
// transposed = 0: rows are targets (A_hat);  1: rows are sources (A_hat^T, the backward's adjacency of a directed graph)
__global__ void k_edge_keys(const int64_t* __restrict__ ei, int64_t n_edges, int n_nodes, int transposed, int* __restrict__ keys,
                            int* __restrict__ vals, int* __restrict__ counts) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_edges) return;
    const int64_t src = ei[e], dst = ei[n_edges + e];
    const bool drop = (src == dst) || src < 0 || dst < 0 || src >= n_nodes || dst >= n_nodes;
    const int64_t row = transposed ? src : dst, col = transposed ? dst : src;
    keys[e] = drop ? n_nodes : (int)row;          // dropped edges sort behind every real row
    vals[e] = (int)col;
    if (!drop) atomicAdd(&counts[row], 1);        // integer: order-independent
}

__global__ void k_dis_from_counts(const int* __restrict__ counts, int n, float* __restrict__ dis) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dis[i] = 1.0f / sqrtf((float)(counts[i] + 1));
}

// ---------------------------------------------------------------- edge digest
__device__ inline unsigned long long mix64(unsigned long long r, unsigned long long c) {
    unsigned long long h = r * 0x9E3779B97F4A7C15ull + c * 0xC2B2AE3D27D4EB4Full + 0x165667B19E3779F9ull;
    h ^= h >> 29;
    h *= 0xBF58476D1CE4E5B9ull;
    h ^= h >> 32;
    return h;
}

__global__ void k_edge_hash(const int64_t* __restrict__ ei, int64_t n_edges, unsigned long long* __restrict__ out) {
    unsigned long long acc = 0;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n_edges; e += (int64_t)gridDim.x * blockDim.x)
        acc += mix64((unsigned long long)ei[e], (unsigned long long)ei[n_edges + e]);
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
    if ((threadIdx.x & 63) == 0) atomicAdd(&out[1], acc);
    if (blockIdx.x == 0 && threadIdx.x == 0) out[0] = (unsigned long long)n_edges;
}

// out[0] = sum mix64(src, dst), out[1] = sum mix64(dst, src) over the edges a CSR keeps: equal <=> (up to a 2^-64 collision)
// the edge multiset equals its own transpose <=> A_hat is symmetric
__global__ void k_edge_sym(const int64_t* __restrict__ ei, int64_t n_edges, int64_t n_nodes, unsigned long long* __restrict__ out) {
    unsigned long long a = 0, b = 0;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n_edges; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t src = ei[e], dst = ei[n_edges + e];
        if (src == dst || src < 0 || dst < 0 || src >= n_nodes || dst >= n_nodes) continue;
        a += mix64((unsigned long long)src, (unsigned long long)dst);
        b += mix64((unsigned long long)dst, (unsigned long long)src);
    }
    for (int off = 32; off > 0; off >>= 1) { a += __shfl_xor(a, off); b += __shfl_xor(b, off); }
    if ((threadIdx.x & 63) == 0) { atomicAdd(&out[0], a); atomicAdd(&out[1], b); }
}

// ---- CSR regrouped into tiles of 64 nodes that are close in the graph ------------------------------------------------------
// The layer kernel aggregates a tile of 64 rows per workgroup.  With 64 CONSECUTIVE node ids per tile almost every source row
// of an edge lies outside the tile and is loaded once per edge (2.8x the bytes of the implicit stencil at configs[1]); if the
// tile is a breadth-first ball of the graph most sources are the tile's own rows, which the kernel loads once into LDS.
// Greedy graph growing: a tile is filled breadth-first from a seed (taken from the frontier the previous tile left behind, so
// that consecutive tiles are neighbours too); what the queue still holds when the tile is full becomes that frontier.
// order[s] = node of tile slot s (tile = s / 64); `cluster` = false: identity order (consecutive rows).
static void csr_tile_order(int n, const std::vector<int>& rowptr, const std::vector<int>& colidx, bool cluster, std::vector<int>& order) {
    order.clear();
    order.reserve((size_t)n);
    if (!cluster) {
        for (int i = 0; i < n; ++i) order.push_back(i);
        return;
    }
    std::vector<unsigned char> seen((size_t)n, 0);      // 1: placed or in the current queue
    std::vector<int> frontier, q;
    size_t fhead = 0;
    int next_seed = 0;
    while ((int)order.size() < n) {
        int filled = 0;
        q.clear();
        size_t qhead = 0;
        while (filled < TILE && (int)order.size() < n) {
            if (qhead == q.size()) {                      // (the component is exhausted, or the tile starts): a new seed
                int seed = -1;
                while (fhead < frontier.size()) {
                    const int c = frontier[fhead++];
                    if (!seen[c]) { seed = c; break; }
                }
                if (seed < 0) {
                    while (seen[next_seed]) ++next_seed;
                    seed = next_seed;
                }
                seen[seed] = 1;
                q.push_back(seed);
            }
            const int u = q[qhead++];
            order.push_back(u);
            ++filled;
            for (int e = rowptr[u]; e < rowptr[u + 1]; ++e) {
                const int v = colidx[e];
                if (!seen[v]) { seen[v] = 1; q.push_back(v); }
            }
        }
        // queued but not placed: the next tiles' seeds, nearest first
        if (fhead > (1u << 20) && fhead * 2 > frontier.size()) { frontier.erase(frontier.begin(), frontier.begin() + (long)fhead); fhead = 0; }
        for (size_t k = qhead; k < q.size(); ++k) { seen[q[k]] = 0; frontier.push_back(q[k]); }
    }
}

// builds g->t_* from the handle's device CSR (host round trip: a set-up call); mode 1: consecutive rows, 2: clustered
static int csr_tiles(eg_graph* g, int mode, hipStream_t stream) {
    const int n = (int)g->n_nodes;
    const size_t nnz = (size_t)g->nnz;
    std::vector<int> rowptr((size_t)n + 1), colidx(nnz ? nnz : 1);
    std::vector<float> dis((size_t)n);
    EG_HIP_TRY(hipMemcpyAsync(rowptr.data(), g->rowptr, sizeof(int) * ((size_t)n + 1), hipMemcpyDeviceToHost, stream));
    if (nnz) EG_HIP_TRY(hipMemcpyAsync(colidx.data(), g->colidx, sizeof(int) * nnz, hipMemcpyDeviceToHost, stream));
    EG_HIP_TRY(hipMemcpyAsync(dis.data(), g->dis, sizeof(float) * (size_t)n, hipMemcpyDeviceToHost, stream));
    EG_HIP_TRY(hipStreamSynchronize(stream));
    std::vector<int> order;
    csr_tile_order(n, rowptr, colidx, mode >= 2, order);
    const int n_tiles = (n + TILE - 1) / TILE;
    const size_t slots = (size_t)n_tiles * TILE;
    std::vector<int> slot_of((size_t)n), t_rows(slots, -1), t_rowptr(slots + 1, 0), t_code(nnz ? nnz : 1), t_tgt(nnz ? nnz : 1);
    std::vector<float> t_w(nnz ? nnz : 1), t_dis(slots, 0.f);
    for (int s = 0; s < n; ++s) slot_of[(size_t)order[(size_t)s]] = s;
    size_t o = 0;
    for (size_t s = 0; s < slots; ++s) {
        t_rowptr[s] = (int)o;
        if (s >= (size_t)n) continue;
        const int u = order[s];
        t_rows[s] = u;
        t_dis[s] = dis[(size_t)u] * dis[(size_t)u];
        for (int e = rowptr[(size_t)u]; e < rowptr[(size_t)u + 1]; ++e) {      // (edge order kept: the sum of a row has the old order)
            const int v = colidx[(size_t)e];
            const int sv = slot_of[(size_t)v];
            t_code[o] = (sv / TILE == (int)(s / TILE)) ? -(sv % TILE + 1) : v;
            t_w[o] = dis[(size_t)v] * dis[(size_t)u];
            t_tgt[o] = (int)(s & 7);
            ++o;
        }
    }
    t_rowptr[slots] = (int)o;
    EG_HIP_TRY(hipMalloc((void**)&g->t_rows, sizeof(int) * slots));
    EG_HIP_TRY(hipMalloc((void**)&g->t_rowptr, sizeof(int) * (slots + 1)));
    EG_HIP_TRY(hipMalloc((void**)&g->t_code, sizeof(int) * t_code.size()));
    EG_HIP_TRY(hipMalloc((void**)&g->t_tgt, sizeof(int) * t_tgt.size()));
    EG_HIP_TRY(hipMalloc((void**)&g->t_w, sizeof(float) * t_w.size()));
    EG_HIP_TRY(hipMalloc((void**)&g->t_dis, sizeof(float) * slots));
    EG_HIP_TRY(hipMemcpy(g->t_rows, t_rows.data(), sizeof(int) * slots, hipMemcpyHostToDevice));
    EG_HIP_TRY(hipMemcpy(g->t_rowptr, t_rowptr.data(), sizeof(int) * (slots + 1), hipMemcpyHostToDevice));
    EG_HIP_TRY(hipMemcpy(g->t_code, t_code.data(), sizeof(int) * t_code.size(), hipMemcpyHostToDevice));
    EG_HIP_TRY(hipMemcpy(g->t_tgt, t_tgt.data(), sizeof(int) * t_tgt.size(), hipMemcpyHostToDevice));
    EG_HIP_TRY(hipMemcpy(g->t_w, t_w.data(), sizeof(float) * t_w.size(), hipMemcpyHostToDevice));
    EG_HIP_TRY(hipMemcpy(g->t_dis, t_dis.data(), sizeof(float) * slots, hipMemcpyHostToDevice));
    g->n_ctiles = n_tiles;
    return EG_OK;
}

static int csr_build(const int64_t* ei, int64_t n_nodes, int64_t n_edges, hipStream_t stream, const eg_graph* base, eg_graph** out);

}  // namespace eg

// ---- eg_debug_layer_timing_*: events around layer-kernel launches while armed (bench.py's in-step kernel durations) ----
namespace {
constexpr int TIMING_MAX = 256;
std::atomic<int> g_timing_armed{0};        // launches still to be timed
std::atomic<int> g_timing_count{0};
hipEvent_t g_timing_ev[2 * TIMING_MAX];
int g_timing_kind[TIMING_MAX];
bool g_timing_have_events = false;
}  // namespace

eg::LaunchTimer::LaunchTimer(int kind, hipStream_t s) : idx(-1), stream(s) {
    if (g_timing_armed.load(std::memory_order_relaxed) <= 0) return;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cs) != hipSuccess) { (void)hipGetLastError(); return; }
    if (cs != hipStreamCaptureStatusNone) return;
    if (g_timing_armed.fetch_sub(1, std::memory_order_relaxed) <= 0) return;
    const int i = g_timing_count.fetch_add(1, std::memory_order_relaxed);
    if (i >= TIMING_MAX) return;
    g_timing_kind[i] = kind;
    if (hipEventRecord(g_timing_ev[2 * i], s) == hipSuccess) idx = i;
}
eg::LaunchTimer::~LaunchTimer() {
    if (idx >= 0) (void)hipEventRecord(g_timing_ev[2 * idx + 1], stream);
}

using namespace eg;

extern "C" {

int eg_version(void) { return EG_ABI_VERSION; }

int eg_debug_layer_timing_begin(int max_launches) {
    if (max_launches < 1 || max_launches > TIMING_MAX) return eg::set_error(EG_ERR_ARG, "max_launches must be in [1, 256]");
    if (!g_timing_have_events) {
        for (int i = 0; i < 2 * TIMING_MAX; ++i) EG_HIP_TRY(hipEventCreate(&g_timing_ev[i]));
        g_timing_have_events = true;
    }
    g_timing_count.store(0, std::memory_order_relaxed);
    g_timing_armed.store(max_launches, std::memory_order_relaxed);
    return EG_OK;
}

int eg_debug_layer_timing_end(float* ms, int* kinds, int cap) {
    g_timing_armed.store(0, std::memory_order_relaxed);
    int n = g_timing_count.load(std::memory_order_relaxed);
    if (n > TIMING_MAX) n = TIMING_MAX;
    if (!ms || !kinds || cap < n) return eg::set_error(EG_ERR_ARG, "ms / kinds too small");
    for (int i = 0; i < n; ++i) {
        EG_HIP_TRY(hipEventSynchronize(g_timing_ev[2 * i + 1]));
        EG_HIP_TRY(hipEventElapsedTime(&ms[i], g_timing_ev[2 * i], g_timing_ev[2 * i + 1]));
        kinds[i] = g_timing_kind[i];
    }
    g_timing_count.store(0, std::memory_order_relaxed);
    return n;
}

const char* eg_last_error(void) { return g_last_error.c_str(); }

int eg_topo_create(int frame, int naux, int main_only, int coord_nodes, int conn_nodes, int diag_main, int diag_aux,
                   eg_graph** out) {
    if (!out) return set_error(EG_ERR_ARG, "out is NULL");
    *out = nullptr;
    Topo T;
    int rc = build_topo(frame, naux, main_only, coord_nodes, conn_nodes, diag_main, diag_aux, T);
    if (rc != EG_OK) return rc;
    const bool any_diag = T.diag_main || T.diag_aux || T.n_conn > 0;      // (any topology whose stencil lives in the producer/consumer kernel only)
    // connection node wired to every node of level l (datasets.py:1512-1515: node g - 1 <-> aux level g, g = 1 .. naux - 1), or -1
    auto hub_of_level = [&](int l) { return (T.n_conn > 0 && l <= T.n_aux - 2) ? l : -1; };
    // level / position of a grid node, and whether its level is 'grid-diagonal'
    auto level_diag = [&](int l) { return l == T.n_levels - 1 ? T.diag_main != 0 : T.diag_aux != 0; };
    auto diag_ids = [&](int n, int (&out)[4]) -> int {          // the up-to-4 diagonal neighbours of node n (datasets.py:1469-1475)
        if (n >= T.coord_base || n < T.n_conn) return 0;
        const int l = level_of(T, n);
        if (!level_diag(l)) return 0;
        const int side = T.side[l], idx = n - T.base[l], r = idx / side, c = idx - r * side;
        int k = 0;
        for (int dr = -1; dr <= 1; dr += 2)
            for (int dc = -1; dc <= 1; dc += 2)
                if (r + dr >= 0 && r + dr < side && c + dc >= 0 && c + dc < side) out[k++] = n + dr * side + dc;
        return k;
    };
    // every neighbour of node n (no self loop), sorted: grid stencil + diagonals + the level's connection node; connection nodes:
    // the other connection nodes + every node of their level
    Nbrs nb;
    auto full_row = [&](int n, std::vector<int>& row) {
        row.clear();
        if (n < T.n_conn) {
            for (int h = 0; h < T.n_conn; ++h) if (h != n) row.push_back(h);
            if (n <= T.n_aux - 2)
                for (int j = T.base[n]; j < T.base[n + 1]; ++j) row.push_back(j);
            return;
        }
        neighbours(T, n, nb);
        for (int sl = 1; sl < nb.count; ++sl)
            if (nb.valid[sl]) row.push_back(nb.id[sl]);
        int dg[4];
        const int nd = diag_ids(n, dg);
        row.insert(row.end(), dg, dg + nd);
        if (n < T.coord_base) {
            const int hub = hub_of_level(level_of(T, n));
            if (hub >= 0) row.push_back(hub);
        }
        std::sort(row.begin(), row.end());
    };
    std::vector<float> dis(T.n_nodes);
    {
        std::vector<int> row;
        for (int n = 0; n < T.n_nodes; ++n) {
            full_row(n, row);
            dis[n] = (float)(1.0 / std::sqrt((double)(row.size() + 1)));
        }
    }
    // 'grid-diagonal' handles also carry the CSR of one frame (sorted by target, then source): every path but the
    // producer/consumer kernel's stencil reads it (common.h eg_graph::hybrid)
    std::vector<int> h_rowptr, h_colidx;
    if (any_diag) {
        h_rowptr.assign((size_t)T.n_nodes + 1, 0);
        std::vector<int> row;
        for (int n = 0; n < T.n_nodes; ++n) {
            full_row(n, row);
            h_colidx.insert(h_colidx.end(), row.begin(), row.end());
            h_rowptr[(size_t)n + 1] = (int)h_colidx.size();
        }
    }
    // 2-D patch table.  Order = depth-first post-order over the pyramid of 8x8 patches: the patches under a
    // coarse patch are emitted (recursively, 2x2 blocks) before it, so vertical neighbours, parents and
    // children are worked on close in time by the workgroups of one XCD and meet in its L2.  Patches that
    // the pyramid does not reach (outside the centre crop, or a main-only graph) follow in 2x2-block order.
    std::vector<TileDesc> tiles;
    {
        std::vector<std::vector<char>> seen(T.n_levels);
        std::vector<int> tside(T.n_levels);
        for (int l = 0; l < T.n_levels; ++l) {
            tside[l] = (T.desc[l].side + 7) / 8;
            seen[l].assign((size_t)tside[l] * tside[l], 0);
        }
        auto push = [&](int l, int ty, int tx) {
            const LevelDesc& d = T.desc[l];
            const int r0 = ty * 8, c0 = tx * 8;
            tiles.push_back(TileDesc{l, r0, c0, d.side - r0 < 8 ? d.side - r0 : 8, d.side - c0 < 8 ? d.side - c0 : 8, 0, 0, 0});
        };
        // explicit stack: (level, ty, tx, state)
        struct Item { int l, ty, tx, expanded; };
        auto visit = [&](int l0, int ty0, int tx0) {
            std::vector<Item> st;
            st.push_back(Item{l0, ty0, tx0, 0});
            while (!st.empty()) {
                Item it = st.back();
                st.pop_back();
                if (it.l < 0 || it.l >= T.n_levels || it.ty < 0 || it.tx < 0 || it.ty >= tside[it.l] || it.tx >= tside[it.l]) continue;
                char& sn = seen[it.l][(size_t)it.ty * tside[it.l] + it.tx];
                if (it.expanded) { push(it.l, it.ty, it.tx); continue; }
                if (sn) continue;
                sn = 1;
                st.push_back(Item{it.l, it.ty, it.tx, 1});
                const LevelDesc& d = T.desc[it.l];
                if (d.kind != 0) continue;                        // main grid: leaf
                // node range of the children of this patch, in the child level's coordinates
                const int rlo = 2 * (it.ty * 8 - d.clo), rhi = 2 * (it.ty * 8 + 8 - d.clo);
                const int clo = 2 * (it.tx * 8 - d.clo), chi = 2 * (it.tx * 8 + 8 - d.clo);
                const int cl = it.l + 1;
                const int climit = 2 * (d.chi - d.clo);
                const int r_a = rlo < 0 ? 0 : rlo, r_b = rhi > climit ? climit : rhi;
                const int c_a = clo < 0 ? 0 : clo, c_b = chi > climit ? climit : chi;
                if (r_a >= r_b || c_a >= c_b) continue;
                for (int ty = (r_b - 1) / 8; ty >= r_a / 8; --ty)        // reversed: the stack pops them in order
                    for (int tx = (c_b - 1) / 8; tx >= c_a / 8; --tx) st.push_back(Item{cl, ty, tx, 0});
            }
        };
        if (T.n_aux > 0) visit(0, 0, 0);
        for (int l = 0; l < T.n_levels; ++l) {                    // whatever the pyramid did not reach: 2x2-block order
            const int ts = tside[l];
            for (int by = 0; by < ts; by += 2)
                for (int bx = 0; bx < ts; bx += 2)
                    for (int dy = 0; dy < 2; ++dy)
                        for (int dx = 0; dx < 2; ++dx) {
                            const int ty = by + dy, tx = bx + dx;
                            if (ty < ts && tx < ts && !seen[l][(size_t)ty * ts + tx]) { seen[l][(size_t)ty * ts + tx] = 1; push(l, ty, tx); }
                        }
        }
        if (T.coord_base < T.n_nodes) {
            const LevelDesc& d = T.desc[T.n_levels];
            tiles.push_back(TileDesc{T.n_levels, 0, 0, 1, d.end - d.base, 0, 0, 0});
        }
        for (int h0 = 0; h0 < T.n_conn; h0 += 8)           // connection nodes: 8 per pseudo-tile (one segment each)
            tiles.push_back(TileDesc{T.n_desc - 1, h0 / 8, 0, 1, T.n_conn - h0 < 8 ? T.n_conn - h0 : 8, 0, 0, 0});
    }
    // Per-segment descriptors (8 per patch) and the table of distinct weight patterns.  A pattern is the 64 + 64
    // lane weights of a segment: lane (u, s) -> valid(slot s of node u) ? (deg + 1)^-1/2 of that neighbour : 0.
    // Interior segments of a level all share one pattern, so the table stays at a few dozen entries.
    std::vector<SegDesc> segs(tiles.size() * 8);
    std::vector<float> pats;
    std::vector<float> pat_extra;                     // 5 floats per pattern (see `w` below)
    int kid_rows = 0;
    bool kidsum_ok = false;
    {
        kid_rows = (T.n_aux > 0 && T.n_levels > 1) ? T.base[T.n_levels - 1] : 0;
        kidsum_ok = kid_rows > 0;
        std::map<std::vector<float>, int> pat_index;
        auto clampi = [](int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); };
        const int n_frame = T.n_nodes, hi8 = n_frame - 8, last = n_frame - 1;
        std::vector<float> w(128 + 5);                // lane weights + {diagonal?, edge weights of the row above (l, r) and below (l, r)}
        for (size_t t = 0; t < tiles.size(); ++t) {
            const TileDesc& td = tiles[t];
            const LevelDesc& d = T.desc[td.level];
            for (int tr = 0; tr < 8; ++tr) {
                SegDesc sd{};
                sd.n_first = d.base + (td.r0 + tr) * d.side + td.c0;
                sd.cnt = tr < td.nrows ? td.ncols : 0;
                const int idx = sd.n_first - d.base;
                const bool grid = d.kind == KIND_AUX || d.kind == KIND_MAIN;       // (not the coordinate / connection pseudo-levels)
                const int r = grid ? idx / d.side : 0;
                const int c0 = idx - r * d.side;
                const int cb = d.cbase + 2 * (r - d.clo) * d.cside + 2 * (c0 - d.clo);
                const bool kids = d.kind == KIND_AUX && r >= d.clo && r < d.chi && c0 < d.chi && c0 + 8 > d.clo;   // some node of the segment has children
                // per-node scalar path: coordinate K4, and segments so close to the end of the frame that a run of
                // 8 rows (self / below / children) would have to be clamped while some of its rows are real neighbours
                // (the rows below the LAST grid row are no neighbours: their clamped run carries weight 0)
                const bool below = grid && r < d.side - 1;
                const bool ldiag = grid && level_diag(td.level);
                // (a 'grid-diagonal' segment takes the run path only when it is a whole 8-node run: seg_wide.h row_sum3)
                const bool slow = !grid || sd.n_first + 8 > n_frame || (below && sd.n_first + d.side + 8 > n_frame) ||
                                  (kids && (cb < 0 || cb + d.cside + 16 > n_frame)) || (ldiag && sd.cnt != 8);
                sd.mode = sd.cnt == 0 ? 0 : (d.kind == KIND_CONN ? 3 : (slow ? 2 : 1));
                sd.aux = (d.kind == KIND_AUX ? 1 : 0) | (ldiag ? 2 : 0) | (d.kind == KIND_AUX ? (hub_of_level(td.level) + 1) << 2 : 0);
                if (sd.mode == 1) {
                    sd.up0 = clampi(sd.n_first - d.side, 0, hi8);
                    sd.down0 = clampi(sd.n_first + d.side, 0, hi8);
                    sd.par0 = clampi(d.pbase + (d.poff + (r >> 1)) * d.pside + d.poff + (c0 >> 1), 0, hi8);
                    sd.left = clampi(sd.n_first - 1, 0, last);
                    sd.right = clampi(sd.n_first + 8, 0, last);
                    sd.c0 = clampi(cb, 0, hi8);
                    sd.c1 = clampi(cb + 8, 0, hi8);
                    sd.c2 = clampi(cb + d.cside, 0, hi8);
                    sd.c3 = clampi(cb + d.cside + 8, 0, hi8);
                    for (int lane = 0; lane < 64; ++lane) {
                        const int u = lane >> 3, sl = lane & 7;
                        const int n = sd.n_first + (u < sd.cnt ? u : sd.cnt - 1);
                        neighbours(T, n, nb);
                        w[lane] = nb.valid[sl] ? dis[nb.id[sl]] : 0.0f;
                        const int sb = 8 + (sl & 1);
                        w[64 + lane] = ((sd.aux & 1) && nb.valid[sb]) ? dis[nb.id[sb]] : 0.0f;
                    }
                    for (int e = 0; e < 5; ++e) w[128 + e] = 0.0f;
                    if (ldiag) {                                // the nodes left / right of the 8-node runs above and below
                        auto dnode = [&](int rr, int cc) { return (rr >= 0 && rr < d.side && cc >= 0 && cc < d.side) ? dis[d.base + rr * d.side + cc] : 0.0f; };
                        w[128] = 1.0f;
                        w[129] = dnode(r - 1, c0 - 1); w[130] = dnode(r - 1, c0 + 8);
                        w[131] = dnode(r + 1, c0 - 1); w[132] = dnode(r + 1, c0 + 8);
                    }
                    auto it = pat_index.find(w);
                    if (it == pat_index.end()) {
                        it = pat_index.emplace(w, (int)pat_index.size()).first;
                        pats.insert(pats.end(), w.begin(), w.begin() + 128);
                        pat_extra.insert(pat_extra.end(), w.begin() + 128, w.end());
                    }
                    sd.pat = it->second;
                }
                segs[t * 8 + tr] = sd;
            }
            for (int tr = 0; tr < 8; tr += 2) {       // pad0 of an even patch row: it and the next row form a pair (seg_wide.h)
                SegDesc& sa = segs[t * 8 + tr];
                const SegDesc& sb = segs[t * 8 + tr + 1];
                sa.pad0 = sa.mode == 1 && sb.mode == 1 && sa.aux == sb.aux && sa.par0 == sb.par0 && sa.down0 == sb.n_first &&
                          sb.up0 == sa.n_first && sa.cnt == sb.cnt;
                // pad1 = number of parents whose four children are exactly columns 2j, 2j+1 of this pair of rows
                // (child-sum side buffer, gcn_layer_ps.hip).  Anything irregular switches the side buffer off.
                if (sa.cnt > 0 && (d.kind == KIND_AUX || d.kind == KIND_MAIN)) {
                    const int idx = sa.n_first - d.base, r = idx / d.side, c0 = idx - r * d.side;
                    int npar = 0;
                    if (r < d.plim) {
                        const int cend = c0 + sa.cnt < d.plim ? c0 + sa.cnt : d.plim;
                        npar = cend > c0 ? (cend - c0) / 2 : 0;
                        if (cend > c0 && ((cend - c0) & 1)) { kidsum_ok = false; if (EG_DEBUG_TOPO_ON) fprintf(stderr, "kidsum off: odd t=%zu tr=%d\n", t, tr); }
                    }
                    const int par_raw = d.pbase + (d.poff + (r >> 1)) * d.pside + d.poff + (c0 >> 1);
                    // ('grid-diagonal' levels of fewer than 8 columns run node by node -- rows pulled through the CSR, children
                    //  included -- and never read the side buffer: child sums that nobody writes for THEIR rows are not missed)
                    const bool parent_slow = td.level > 0 && level_diag(td.level - 1) && T.side[td.level - 1] < 8 && d.kind == KIND_AUX;
                    if (npar > 0 && !parent_slow && (!sa.pad0 || par_raw != sa.par0 || par_raw + npar > kid_rows)) { kidsum_ok = false; if (EG_DEBUG_TOPO_ON) fprintf(stderr, "kidsum off: parent t=%zu tr=%d level=%d pad0=%d par_raw=%d par0=%d npar=%d kid_rows=%d modes %d %d\n", t, tr, td.level, sa.pad0, par_raw, sa.par0, npar, kid_rows, sa.mode, sb.mode); }
                    sa.pad1 = npar;
                    const bool kids = d.kind == KIND_AUX && ((r >= d.clo && r < d.chi) || (r + 1 >= d.clo && r + 1 < d.chi)) && c0 < d.chi && c0 + 8 > d.clo;
                    const bool self_slow = level_diag(td.level) && d.side < 8;
                    // the pair path reads runs of 8 child-sum rows from each segment's first node: they must stay inside the
                    // frame's slice of the side buffer (tiny pyramids only: a 2x2 or 4x4 level right at its end; an over-read
                    // past the LAST frame's slice left the allocation and aborted a test run once)
                    if (kids && !self_slow && (sa.n_first + 8 > kid_rows || sb.n_first + 8 > kid_rows)) { kidsum_ok = false; if (EG_DEBUG_TOPO_ON) fprintf(stderr, "kidsum off: 8-row run past the side buffer t=%zu tr=%d\n", t, tr); }
                    if (kids && !sa.pad0 && !self_slow) { kidsum_ok = false; if (EG_DEBUG_TOPO_ON) fprintf(stderr, "kidsum off: kids unpaired t=%zu tr=%d level=%d\n", t, tr, td.level); }      // a segment that would read the side buffer is not on the pair path
                }
            }
        }
        if (pats.empty()) { pats.assign(128, 0.0f); pat_extra.assign(5, 0.0f); }
        if (EG_DEBUG_TOPO_ON) fprintf(stderr, "topo: %zu tiles, %zu patterns, kidsum %d\n", tiles.size(), pats.size() / 128, (int)kidsum_ok);
    }
    eg_graph* g = new eg_graph{};
    g->only_stream.store(EG_NO_STREAM_YET, std::memory_order_relaxed);
    g->kind = GRAPH_TOPO;
    g->knobs = read_knobs();
    g->n_nodes = T.n_nodes;
    g->n_pats = (int)(pats.size() / 128);
    // The same patterns in "quad" layout for the producer/consumer kernel, which keeps them in LDS: [pattern][row parity
    // h][slot 0..7][k 0..3] = weight of (node 2k + h, slot); slot 6 = 1.0 when the node has children.  The outer
    // neighbours of a segment's first and last node travel apart from the inner ones (seg_wide.h, segw_rows): node 0's
    // left weight and node 7's right weight sit in slot 7 (k = 0 of h = 0, k = 3 of h = 1) and are zero in slots 3 / 4.
    std::vector<float> patsq((size_t)g->n_pats * 64, 0.0f);
    for (int pi = 0; pi < g->n_pats; ++pi)
        for (int h = 0; h < 2; ++h)
            for (int sl = 0; sl < 7; ++sl)
                for (int k = 0; k < 4; ++k) {
                    const float w = pats[(size_t)pi * 128 + (2 * k + h) * 8 + sl];
                    const bool outer = (sl == 3 && h == 0 && k == 0) || (sl == 4 && h == 1 && k == 3);
                    patsq[(size_t)pi * 64 + h * 32 + sl * 4 + k] = outer ? 0.0f : (sl < 6 ? w : (w != 0.0f ? 1.0f : 0.0f));
                    if (outer) patsq[(size_t)pi * 64 + h * 32 + 7 * 4 + k] = w;
                }
    for (int pi = 0; pi < g->n_pats && !pat_extra.empty(); ++pi) {
        const float* ex = &pat_extra[(size_t)pi * 5];
        if (ex[0] == 0.0f) continue;
        // diagonal segment: slots 3 / 4 = edge nodes of the row above / below (seg_wide.h SLOT_EDGE_U / SLOT_EDGE_D), laid out like
        // slot 7: the left edge belongs to node 0 (k = 0 of h = 0), the right edge to node 7 (k = 3 of h = 1)
        for (int h = 0; h < 2; ++h)
            for (int k = 0; k < 4; ++k) { patsq[(size_t)pi * 64 + h * 32 + 3 * 4 + k] = 0.0f; patsq[(size_t)pi * 64 + h * 32 + 4 * 4 + k] = 0.0f; }
        patsq[(size_t)pi * 64 + 0 * 32 + 3 * 4 + 0] = ex[1]; patsq[(size_t)pi * 64 + 1 * 32 + 3 * 4 + 3] = ex[2];
        patsq[(size_t)pi * 64 + 0 * 32 + 4 * 4 + 0] = ex[3]; patsq[(size_t)pi * 64 + 1 * 32 + 4 * 4 + 3] = ex[4];
    }
    // chained layers run the producer/consumer kernel, which keeps the pattern table in LDS beside its tile buffers
    const size_t ps_lds = (size_t)(4 * TILE * LDA + 16 + 64 + 2 * TILE + (pats.size() / 128) * 64 + 4 * C) * sizeof(float);   // incl. the fused-classifier tables
    g->kid_rows = (kidsum_ok && ps_lds <= 160 * 1024) ? kid_rows : 0;
    g->flat = (T.n_levels == 1 && T.n_desc == 1 && ps_lds <= 160 * 1024) ? 1 : 0;
    g->topo = T;
    g->n_tiles = (int)tiles.size();
    hipError_t e = hipMalloc((void**)&g->dis, sizeof(float) * T.n_nodes);
    if (e == hipSuccess) e = hipMemcpy(g->dis, dis.data(), sizeof(float) * T.n_nodes, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc((void**)&g->walk_counters, QUEUE_RING_BYTES);
    if (e == hipSuccess) e = hipMemset(g->walk_counters, 0, QUEUE_RING_BYTES);
    if (e == hipSuccess && create_slot_events(g) != EG_OK) e = hipErrorOutOfMemory;
    if (e == hipSuccess) e = hipMalloc((void**)&g->topo_dev, sizeof(Topo));
    if (e == hipSuccess) e = hipMemcpy(g->topo_dev, &T, sizeof(Topo), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc((void**)&g->tiles_dev, sizeof(TileDesc) * tiles.size());
    if (e == hipSuccess) e = hipMemcpy(g->tiles_dev, tiles.data(), sizeof(TileDesc) * tiles.size(), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc((void**)&g->segs_dev, sizeof(SegDesc) * segs.size());
    if (e == hipSuccess) e = hipMemcpy(g->segs_dev, segs.data(), sizeof(SegDesc) * segs.size(), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc((void**)&g->pats_dev, sizeof(float) * pats.size());
    if (e == hipSuccess) e = hipMemcpy(g->pats_dev, pats.data(), sizeof(float) * pats.size(), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc((void**)&g->patsq_dev, sizeof(float) * patsq.size());
    if (e == hipSuccess) e = hipMemcpy(g->patsq_dev, patsq.data(), sizeof(float) * patsq.size(), hipMemcpyHostToDevice);
    if (T.n_conn > 0) {
        // chunks of <= 256 rows over the levels that hang on a connection node (levels 0 .. naux - 2): the pre-pass (conn.hip) sums
        // (deg + 1)^-1/2 x over a chunk per workgroup, then over a level's chunks in order
        std::vector<int> table;
        for (int l = 0; l <= T.n_aux - 2; ++l)
            for (int r0 = T.base[l]; r0 < T.base[l + 1]; r0 += 256) {
                const int rows = T.base[l + 1] - r0 < 256 ? T.base[l + 1] - r0 : 256;
                table.insert(table.end(), {l, r0, rows, 0});
            }
        g->n_conn = T.n_conn;
        g->conn_chunks = (int)(table.size() / 4);
        g->conn_cap = 8;                                   // frames the scratch holds at first (it grows: conn.hip)
        if (g->conn_cap < 1) g->conn_cap = 1;
        const size_t per_frame = (size_t)(g->conn_chunks + 2 * T.n_conn) * C;
        if (e == hipSuccess) e = hipMalloc((void**)&g->conn_table, sizeof(int) * (table.empty() ? 4 : table.size()));
        if (e == hipSuccess && !table.empty()) e = hipMemcpy(g->conn_table, table.data(), sizeof(int) * table.size(), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMalloc((void**)&g->conn_scratch, sizeof(float) * per_frame * g->conn_cap * QUEUE_SLOTS);
    }
    if (any_diag) {
        g->hybrid = 1;
        g->nnz = (int64_t)h_colidx.size();
        g->symmetric = 1;
        if (e == hipSuccess) e = hipMalloc((void**)&g->rowptr, sizeof(int) * h_rowptr.size());
        if (e == hipSuccess) e = hipMemcpy(g->rowptr, h_rowptr.data(), sizeof(int) * h_rowptr.size(), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMalloc((void**)&g->colidx, sizeof(int) * (h_colidx.empty() ? 1 : h_colidx.size()));
        if (e == hipSuccess && !h_colidx.empty()) e = hipMemcpy(g->colidx, h_colidx.data(), sizeof(int) * h_colidx.size(), hipMemcpyHostToDevice);
    }
    if (e != hipSuccess) {
        if (g->segs_dev) (void)hipFree(g->segs_dev);
        if (g->pats_dev) (void)hipFree(g->pats_dev);
        if (g->patsq_dev) (void)hipFree(g->patsq_dev);
        if (g->dis) (void)hipFree(g->dis);
        if (g->topo_dev) (void)hipFree(g->topo_dev);
        if (g->tiles_dev) (void)hipFree(g->tiles_dev);
        if (g->walk_counters) (void)hipFree(g->walk_counters);
        if (g->rowptr) (void)hipFree(g->rowptr);
        if (g->colidx) (void)hipFree(g->colidx);
        if (g->conn_table) (void)hipFree(g->conn_table);
        if (g->conn_scratch) (void)hipFree(g->conn_scratch);
        delete g;
        return set_error(EG_ERR_HIP, std::string("eg_topo_create: ") + hipGetErrorString(e));
    }
    *out = g;
    return EG_OK;
}

int eg_csr_create(const int64_t* ei, int64_t n_nodes, int64_t n_edges, eg_stream_t stream, eg_graph** out) {
    return csr_build(ei, n_nodes, n_edges, (hipStream_t)stream, nullptr, out);
}

int eg_csr_create_transposed(const eg_graph* base, const int64_t* ei, int64_t n_edges, eg_stream_t stream, eg_graph** out) {
    if (!base || base->kind != GRAPH_CSR) return set_error(EG_ERR_ARG, "base must be a CSR handle");
    return csr_build(ei, base->n_nodes, n_edges, (hipStream_t)stream, base, out);
}

int eg_graph_is_symmetric(const eg_graph* g) { return g && (g->kind == GRAPH_TOPO || g->symmetric); }

}  // extern "C"

// base == NULL: rows = targets, (deg+1)^-1/2 from the in-degrees.  base != NULL: rows = sources (the transposed adjacency
// of the same edge_index), normalisation copied from base.
static int eg::csr_build(const int64_t* ei, int64_t n_nodes, int64_t n_edges, hipStream_t stream, const eg_graph* base, eg_graph** out) {
    if (!out) return set_error(EG_ERR_ARG, "out is NULL");
    *out = nullptr;
    if (n_nodes <= 0 || n_nodes >= (1ll << 31) - 1 || n_edges < 0 || n_edges >= (1ll << 31) - 1)
        return set_error(EG_ERR_ARG, "n_nodes / n_edges out of int32 range");
    if (n_edges > 0 && !ei) return set_error(EG_ERR_ARG, "edge_index is NULL");
    const int n = (int)n_nodes;
    const int m = (int)n_edges;
    eg_graph* g = new eg_graph{};
    g->only_stream.store(EG_NO_STREAM_YET, std::memory_order_relaxed);
    g->kind = GRAPH_CSR;
    g->knobs = read_knobs();
    g->n_nodes = n_nodes;
    int *keys = nullptr, *vals = nullptr, *keys_out = nullptr, *counts = nullptr;
    unsigned long long* sym = nullptr;
    void* tmp = nullptr;
    size_t tmp_bytes = 0, tmp2 = 0;
    auto cleanup = [&](bool all) {
        if (keys) (void)hipFree(keys);
        if (vals) (void)hipFree(vals);
        if (keys_out) (void)hipFree(keys_out);
        if (counts) (void)hipFree(counts);
        if (sym) (void)hipFree(sym);
        if (tmp) (void)hipFree(tmp);
        if (all) {
            if (g->dis) (void)hipFree(g->dis);
            if (g->rowptr) (void)hipFree(g->rowptr);
            if (g->colidx) (void)hipFree(g->colidx);
            if (g->walk_counters) (void)hipFree(g->walk_counters);
            delete g;
        }
    };
#define CSR_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess) {                                                                    \
            cleanup(true);                                                                         \
            return set_error(EG_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));       \
        }                                                                                          \
    } while (0)
    const size_t mm = (size_t)(m > 0 ? m : 1);
    CSR_TRY(hipMalloc((void**)&keys, sizeof(int) * mm));
    CSR_TRY(hipMalloc((void**)&vals, sizeof(int) * mm));
    CSR_TRY(hipMalloc((void**)&keys_out, sizeof(int) * mm));
    CSR_TRY(hipMalloc((void**)&counts, sizeof(int) * ((size_t)n + 1)));
    CSR_TRY(hipMalloc((void**)&g->colidx, sizeof(int) * mm));
    CSR_TRY(hipMalloc((void**)&g->rowptr, sizeof(int) * ((size_t)n + 1)));
    CSR_TRY(hipMalloc((void**)&g->dis, sizeof(float) * (size_t)n));
    CSR_TRY(hipMalloc((void**)&g->walk_counters, QUEUE_RING_BYTES));
    CSR_TRY(hipMemsetAsync(g->walk_counters, 0, QUEUE_RING_BYTES, stream));
    if (create_slot_events(g) != EG_OK) CSR_TRY(hipErrorOutOfMemory);
    CSR_TRY(hipMemsetAsync(counts, 0, sizeof(int) * ((size_t)n + 1), stream));
    if (m > 0) {
        hipLaunchKernelGGL(k_edge_keys, dim3((m + 255) / 256), dim3(256), 0, stream, ei, n_edges, n, base ? 1 : 0, keys, vals, counts);
        CSR_TRY(hipGetLastError());
        // stable LSD radix sort by target: neighbours keep their edge_index order
        int end_bit = 1;
        while ((1ll << end_bit) <= n_nodes && end_bit < 32) ++end_bit;
        CSR_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, keys, keys_out, vals, g->colidx, m, 0, end_bit, stream));
    }
    CSR_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp2, counts, g->rowptr, n + 1, stream));
    if (tmp2 > tmp_bytes) tmp_bytes = tmp2;
    CSR_TRY(hipMalloc(&tmp, tmp_bytes ? tmp_bytes : 16));
    if (m > 0) {
        size_t tb = tmp_bytes;
        int end_bit = 1;
        while ((1ll << end_bit) <= n_nodes && end_bit < 32) ++end_bit;
        CSR_TRY(hipcub::DeviceRadixSort::SortPairs(tmp, tb, keys, keys_out, vals, g->colidx, m, 0, end_bit, stream));
    }
    {
        size_t tb = tmp_bytes;
        CSR_TRY(hipcub::DeviceScan::ExclusiveSum(tmp, tb, counts, g->rowptr, n + 1, stream));
    }
    unsigned long long sym_host[2] = {0, 0};
    if (base) {
        CSR_TRY(hipMemcpyAsync(g->dis, base->dis, sizeof(float) * (size_t)n, hipMemcpyDeviceToDevice, stream));
    } else {
        hipLaunchKernelGGL(k_dis_from_counts, dim3((n + 255) / 256), dim3(256), 0, stream, counts, n, g->dis);
        CSR_TRY(hipGetLastError());
        CSR_TRY(hipMalloc((void**)&sym, 2 * sizeof(unsigned long long)));
        CSR_TRY(hipMemsetAsync(sym, 0, 2 * sizeof(unsigned long long), stream));
        if (m > 0) {
            int blocks = (m + 255) / 256;
            hipLaunchKernelGGL(k_edge_sym, dim3(blocks > 2048 ? 2048 : blocks), dim3(256), 0, stream, ei, n_edges, n_nodes, sym);
            CSR_TRY(hipGetLastError());
        }
        CSR_TRY(hipMemcpyAsync(sym_host, sym, sizeof(sym_host), hipMemcpyDeviceToHost, stream));
    }
    int nnz = 0;
    CSR_TRY(hipMemcpyAsync(&nnz, g->rowptr + n, sizeof(int), hipMemcpyDeviceToHost, stream));
    CSR_TRY(hipStreamSynchronize(stream));
    g->nnz = nnz;
    g->symmetric = base ? base->symmetric : (sym_host[0] == sym_host[1]);
    cleanup(false);
#undef CSR_TRY
    if (g->knobs.csr_tiles > 0) {
        const int rc = csr_tiles(g, g->knobs.csr_tiles, stream);
        if (rc != EG_OK) { eg_graph_destroy(g); return rc; }
    }
    *out = g;
    return EG_OK;
}

extern "C" {

int eg_graph_destroy(eg_graph* g) {
    if (!g) return EG_OK;
    if (g->dis) (void)hipFree(g->dis);
    if (g->topo_dev) (void)hipFree(g->topo_dev);
    if (g->segs_dev) (void)hipFree(g->segs_dev);
    if (g->pats_dev) (void)hipFree(g->pats_dev);
    if (g->patsq_dev) (void)hipFree(g->patsq_dev);
    if (g->tiles_dev) (void)hipFree(g->tiles_dev);
    if (g->walk_counters) (void)hipFree(g->walk_counters);
    for (int i = 0; i < eg::QUEUE_SLOTS; ++i)
        if (g->slot_event[i]) (void)hipEventDestroy(g->slot_event[i]);
    if (g->era_event) (void)hipEventDestroy(g->era_event);
    if (g->rowptr) (void)hipFree(g->rowptr);
    if (g->colidx) (void)hipFree(g->colidx);
    if (g->conn_table) (void)hipFree(g->conn_table);
    if (g->conn_scratch) (void)hipFree(g->conn_scratch);
    for (float* p : g->conn_retired) (void)hipFree(p);
    for (void* p : {(void*)g->t_rows, (void*)g->t_rowptr, (void*)g->t_code, (void*)g->t_tgt, (void*)g->t_w, (void*)g->t_dis})
        if (p) (void)hipFree(p);
    delete g;
    return EG_OK;
}

int64_t eg_graph_num_nodes(const eg_graph* g) { return g ? g->n_nodes : -1; }

int eg_graph_is_structured(const eg_graph* g) { return g && g->kind == GRAPH_TOPO; }

int64_t eg_graph_kidsum_rows(const eg_graph* g) { return (g && g->kind == GRAPH_TOPO) ? g->kid_rows : 0; }

int64_t eg_graph_num_tiles(const eg_graph* g) { return g ? (g->kind == GRAPH_TOPO ? g->n_tiles : (g->n_nodes + TILE - 1) / TILE) : -1; }

int eg_graph_deg_inv_sqrt(const eg_graph* g, float* out_dev, eg_stream_t stream) {
    if (!g || !out_dev) return set_error(EG_ERR_ARG, "NULL argument");
    EG_HIP_TRY(hipMemcpyAsync(out_dev, g->dis, sizeof(float) * g->n_nodes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return EG_OK;
}

int eg_debug_phase_cycles(eg_graph* g, uint64_t* out_host, int reset) {
    if (!g || !out_host) return set_error(EG_ERR_ARG, "NULL argument");
    EG_HIP_TRY(hipDeviceSynchronize());
    int* const tail = g->walk_counters + (size_t)QUEUE_SLOTS * QUEUE_SLICE_INTS;
    EG_HIP_TRY(hipMemcpy(out_host, tail, 11 * sizeof(uint64_t), hipMemcpyDeviceToHost));
    if (reset) EG_HIP_TRY(hipMemset(tail, 0, 16 * sizeof(uint64_t)));
    return EG_OK;
}

int eg_edge_hash(const int64_t* ei, int64_t n_edges, uint64_t* out_dev, eg_stream_t stream_) {
    if (!out_dev || (n_edges > 0 && !ei) || n_edges < 0) return set_error(EG_ERR_ARG, "bad argument");
    hipStream_t stream = (hipStream_t)stream_;
    EG_HIP_TRY(hipMemsetAsync(out_dev, 0, 2 * sizeof(uint64_t), stream));
    int blocks = (int)((n_edges + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_edge_hash, dim3(blocks), dim3(256), 0, stream, ei, n_edges, (unsigned long long*)out_dev);
    EG_HIP_TRY(hipGetLastError());
    return EG_OK;
}

}  // extern "C"
