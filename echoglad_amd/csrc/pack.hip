// Node-feature packing: per-level NCHW feature maps -> the node-major [B*N, 128] input of the GNN stack (gfx950).
// Replaces the per-sample Python loops at the tail of the reference's create_node_pixels
// (src/core/models.py:498-537, :590-636, :707-756): for every frame, `map[i].permute(1, 2, 0).reshape(-1, 128)` of
// every level followed by torch.cat, i.e. B*(naux+2) small permute / copy kernels and a re-copy of the whole batch.
// One launch moves every byte once: a workgroup takes 64 consecutive positions of one (frame, level), reads the 128
// channel rows (256 B contiguous each) into LDS and writes 64 node rows (512 B contiguous each).  The reverse
// direction (gradient of the packing) is the same tile walked the other way.
#include "common.h"

namespace eg {

constexpr int PK_MAX_LEVELS = 16;
constexpr int PK_THREADS = 256;
constexpr int PK_POS = 64;

struct PackArgs {
    const float* map[PK_MAX_LEVELS];      // [batch, 128, side, side]
    float* gmap[PK_MAX_LEVELS];           // reverse direction: destination maps
    int side[PK_MAX_LEVELS];
    int row0[PK_MAX_LEVELS];              // first row of the level inside a frame
    int tile0[PK_MAX_LEVELS + 1];         // prefix sum of 64-position tiles per level
    int n_levels, batch, n_rows;          // n_rows: rows per frame of the node-major tensor
};

template <bool REVERSE>
__global__ __launch_bounds__(PK_THREADS) void k_pack_levels(const PackArgs a, float* __restrict__ nodes) {
    __shared__ float s[C][PK_POS + 1];
    const int tiles_per_frame = a.tile0[a.n_levels];
    const int b = blockIdx.x / tiles_per_frame, t = blockIdx.x - b * tiles_per_frame;
    int l = 0;
    for (int k = 1; k < a.n_levels; ++k) l += t >= a.tile0[k] ? 1 : 0;
    const int P = a.side[l] * a.side[l];
    const int pos0 = (t - a.tile0[l]) * PK_POS;
    const int npos = min(PK_POS, P - pos0);
    const int tid = threadIdx.x, p = tid & 63, q = tid >> 6;
    float* rows = nodes + ((size_t)b * a.n_rows + a.row0[l] + pos0) * C;
    // 16-B accesses on both sides when the level's plane size keeps every channel row 16-B aligned (even sides)
    const bool vec = (P & 3) == 0;
    const int p4 = tid & 15, c16 = tid >> 4;                 // map side: 16 lanes x 16 B per channel row, 16 rows per pass
    const int h = tid >> 5 & 1, q4 = tid & 31, w2 = tid >> 6;  // node side: half-wave = one 512-B row, 2 rows per wave access
    if (!REVERSE) {
        const float* src = a.map[l] + (size_t)b * C * P + pos0;
        if (vec) {
#pragma unroll
            for (int k = 0; k < C / 16; ++k) {
                const int c = 16 * k + c16;
                float4 v = {0.f, 0.f, 0.f, 0.f};
                if (4 * p4 < npos) v = *reinterpret_cast<const float4*>(src + (size_t)c * P + 4 * p4);
                s[c][4 * p4] = v.x; s[c][4 * p4 + 1] = v.y; s[c][4 * p4 + 2] = v.z; s[c][4 * p4 + 3] = v.w;
            }
        } else {
#pragma unroll 8
            for (int k = 0; k < C / 4; ++k) {
                const int c = 4 * k + q;
                s[c][p] = p < npos ? src[(size_t)c * P + p] : 0.f;
            }
        }
        __syncthreads();
        for (int r = 2 * w2 + h; r < npos; r += 8)
            *reinterpret_cast<float4*>(rows + (size_t)r * C + 4 * q4) =
                float4{s[4 * q4][r], s[4 * q4 + 1][r], s[4 * q4 + 2][r], s[4 * q4 + 3][r]};
    } else {
        for (int r = 2 * w2 + h; r < npos; r += 8) {
            const float4 v = *reinterpret_cast<const float4*>(rows + (size_t)r * C + 4 * q4);
            s[4 * q4][r] = v.x; s[4 * q4 + 1][r] = v.y; s[4 * q4 + 2][r] = v.z; s[4 * q4 + 3][r] = v.w;
        }
        __syncthreads();
        float* dst = a.gmap[l] + (size_t)b * C * P + pos0;
        if (vec) {
#pragma unroll
            for (int k = 0; k < C / 16; ++k) {
                const int c = 16 * k + c16;
                if (4 * p4 < npos)
                    *reinterpret_cast<float4*>(dst + (size_t)c * P + 4 * p4) =
                        float4{s[c][4 * p4], s[c][4 * p4 + 1], s[c][4 * p4 + 2], s[c][4 * p4 + 3]};
            }
        } else {
#pragma unroll 8
            for (int k = 0; k < C / 4; ++k) {
                const int c = 4 * k + q;
                if (p < npos) dst[(size_t)c * P + p] = s[c][p];
            }
        }
    }
}

static int fill_pack(const void* const* maps, const int* side, int n_levels, int batch, int64_t n_rows, int64_t row_offset,
                     bool reverse, PackArgs& a) {
    if (!maps || !side || n_levels < 1 || n_levels > PK_MAX_LEVELS || batch < 1 || n_rows < 1 || row_offset < 0)
        return set_error(EG_ERR_ARG, "bad argument");
    if (n_rows * (int64_t)batch >= (1ll << 31)) return set_error(EG_ERR_ARG, "batch * rows exceeds int32");
    a.n_levels = n_levels; a.batch = batch; a.n_rows = (int)n_rows;
    int64_t row = row_offset;
    int tiles = 0;
    for (int l = 0; l < n_levels; ++l) {
        if (!maps[l] || side[l] < 1) return set_error(EG_ERR_ARG, "NULL level map or bad side");
        const int64_t P = (int64_t)side[l] * side[l];
        if (reverse) a.gmap[l] = (float*)maps[l]; else a.map[l] = (const float*)maps[l];
        a.side[l] = side[l]; a.row0[l] = (int)row; a.tile0[l] = tiles;
        row += P;
        tiles += (int)((P + PK_POS - 1) / PK_POS);
    }
    if (row > n_rows) return set_error(EG_ERR_ARG, "levels do not fit the frame's rows");
    a.tile0[n_levels] = tiles;
    return EG_OK;
}

}  // namespace eg

using namespace eg;

extern "C" {

int eg_pack_levels(const float* const* level_maps, const int* level_side, int n_levels, int batch, int64_t n_rows,
                   int64_t row_offset, float* nodes, eg_stream_t stream) {
    if (!nodes) return set_error(EG_ERR_ARG, "nodes is NULL");
    PackArgs a{};
    int rc = fill_pack((const void* const*)level_maps, level_side, n_levels, batch, n_rows, row_offset, false, a);
    if (rc != EG_OK) return rc;
    hipLaunchKernelGGL(k_pack_levels<false>, dim3((unsigned)(batch * a.tile0[n_levels])), dim3(PK_THREADS), 0,
                       (hipStream_t)stream, a, nodes);
    EG_HIP_TRY(hipGetLastError());
    return EG_OK;
}

int eg_unpack_levels(const float* nodes, float* const* level_maps, const int* level_side, int n_levels, int batch,
                     int64_t n_rows, int64_t row_offset, eg_stream_t stream) {
    if (!nodes) return set_error(EG_ERR_ARG, "nodes is NULL");
    PackArgs a{};
    int rc = fill_pack((const void* const*)level_maps, level_side, n_levels, batch, n_rows, row_offset, true, a);
    if (rc != EG_OK) return rc;
    hipLaunchKernelGGL(k_pack_levels<true>, dim3((unsigned)(batch * a.tile0[n_levels])), dim3(PK_THREADS), 0,
                       (hipStream_t)stream, a, const_cast<float*>(nodes));
    EG_HIP_TRY(hipGetLastError());
    return EG_OK;
}

}  // extern "C"
