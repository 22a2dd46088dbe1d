// Node-feature packing: per-level NCHW feature maps -> the node-major [B*N, 128] input of the GNN stack (gfx950).
// Replaces the per-sample Python loops at the tail of the reference's create_node_pixels
// (src/core/models.py:498-537, :590-636, :707-756): for every frame, `map[i].permute(1, 2, 0).reshape(-1, 128)` of
// every level followed by torch.cat, i.e. B*(naux+2) small permute / copy kernels and a re-copy of the whole batch.
// One launch moves every byte once: a workgroup takes 64 consecutive positions of one (frame, level), reads the 128
// channel rows (256 B contiguous each) into LDS and writes 64 node rows (512 B contiguous each).  The reverse
// direction (gradient of the packing) is the same tile walked the other way.
#include "common.h"

namespace eg {

typedef float f32x4 __attribute__((ext_vector_type(4)));

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


// ---- 1x1 convolution + ReLU fused into the packing (reference src/core/models.py:707-710 in front of :726-756) ----------
// The UNet variant turns every decoder map [B, C_l, p, p] into 128 channels with its own Conv2d(C_l, 128, kernel_size=1)
// + ReLU and THEN packs: 37 MB per frame written as NCHW and read again.  Here a workgroup takes 64 consecutive positions
// of one (frame, level), runs the [C_l -> 128] product on them (C_l = 4 .. 512: at most 118 MFLOP per frame over all levels,
// so plain FMAs from LDS; the launch is bound by the 512 B it writes per node) and stores node-major rows directly.
struct ConvPackArgs {
    const float* feat[PK_MAX_LEVELS];     // [batch, cin_l, side_l, side_l]
    const float* w[PK_MAX_LEVELS];        // [128, cin_l]  (Conv2d weight [128, cin_l, 1, 1])
    const float* bias[PK_MAX_LEVELS];     // [128] or NULL
    int cin[PK_MAX_LEVELS];
    int side[PK_MAX_LEVELS];
    int row0[PK_MAX_LEVELS];
    int tile0[PK_MAX_LEVELS + 1];
    int n_levels, batch, n_rows;
};
constexpr int CP_CC = 32;                  // input channels per pass through LDS

// thread (o4 = tid & 31, ps = tid >> 5): output channels 4 o4 .. 4 o4 + 3 of positions ps, ps + 8, .., ps + 56.  A half-wave
// holds one whole 512-B node row per position, so the result is stored as it stands; per input channel a thread reads its 4
// weights with one ds_read_b128 (weights staged transposed, [channel][128]) and the 8 positions as half-wave broadcasts.
__global__ __launch_bounds__(PK_THREADS) void k_conv1x1_relu_pack(const ConvPackArgs a, float* __restrict__ nodes) {
    __shared__ __attribute__((aligned(16))) float s_f[CP_CC][PK_POS];
    __shared__ __attribute__((aligned(16))) float s_w[CP_CC][C + 4];
    const int tiles_per_frame = a.tile0[a.n_levels];
    const int b = blockIdx.x / tiles_per_frame, t = blockIdx.x - b * tiles_per_frame;
    int l = 0;
    for (int k = 1; k < a.n_levels; ++k) l += t >= a.tile0[k] ? 1 : 0;
    const int P = a.side[l] * a.side[l], cin = a.cin[l];
    const int pos0 = (t - a.tile0[l]) * PK_POS;
    const int npos = min(PK_POS, P - pos0);
    const int tid = threadIdx.x, o4 = tid & 31, ps = tid >> 5;
    const float* __restrict__ src = a.feat[l] + (size_t)b * cin * P + pos0;
    const float* __restrict__ wl = a.w[l];
    const bool vec = (P & 3) == 0;                                        // channel planes 16-B aligned: 16-B loads
    f32x4 acc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int c0 = 0; c0 < cin; c0 += CP_CC) {
        const int cc = min(CP_CC, cin - c0);
        if (vec) {                                                        // 16 lanes x 16 B per channel plane run, 16 planes per pass
            const int p4 = tid & 15, cr = tid >> 4;
            for (int c = cr; c < cc; c += 16) {
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (4 * p4 < npos) v = *reinterpret_cast<const f32x4*>(src + (size_t)(c0 + c) * P + 4 * p4);
                *reinterpret_cast<f32x4*>(&s_f[c][4 * p4]) = v;
            }
        } else {
            for (int i = tid; i < cc * PK_POS; i += PK_THREADS) {
                const int c = i >> 6, pp = i & 63;
                s_f[c][pp] = pp < npos ? src[(size_t)(c0 + c) * P + pp] : 0.f;
            }
        }
        for (int i = tid; i < C * cc; i += PK_THREADS) {                  // W[o][c0 + c] -> s_w[c][o]
            const int o = i / cc, c = i - o * cc;
            s_w[c][o] = wl[(size_t)o * cin + c0 + c];
        }
        __syncthreads();
#pragma unroll 4
        for (int c = 0; c < cc; ++c) {
            const f32x4 w = *reinterpret_cast<const f32x4*>(&s_w[c][4 * o4]);
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[k] += w * s_f[c][ps + 8 * k];
        }
        __syncthreads();
    }
    const float* __restrict__ bl = a.bias[l];
    const f32x4 bv = bl ? *reinterpret_cast<const f32x4*>(bl + 4 * o4) : f32x4{0.f, 0.f, 0.f, 0.f};
    float* rows = nodes + ((size_t)b * a.n_rows + a.row0[l] + pos0) * C + 4 * o4;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int r = ps + 8 * k;
        f32x4 v = acc[k] + bv;
        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        if (r < npos) *reinterpret_cast<f32x4*>(rows + (size_t)r * C) = v;
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

int eg_conv1x1_relu_pack_levels(const float* const* level_feats, const float* const* level_weights, const float* const* level_biases,
                                const int* level_channels, const int* level_side, int n_levels, int batch, int64_t n_rows,
                                int64_t row_offset, float* nodes, eg_stream_t stream) {
    if (!nodes || !level_feats || !level_weights || !level_channels || !level_side) return set_error(EG_ERR_ARG, "NULL argument");
    if (n_levels < 1 || n_levels > PK_MAX_LEVELS || batch < 1 || n_rows < 1 || row_offset < 0) return set_error(EG_ERR_ARG, "bad argument");
    if (n_rows * (int64_t)batch >= (1ll << 31)) return set_error(EG_ERR_ARG, "batch * rows exceeds int32");
    ConvPackArgs a{};
    a.n_levels = n_levels; a.batch = batch; a.n_rows = (int)n_rows;
    int64_t row = row_offset;
    int tiles = 0;
    for (int l = 0; l < n_levels; ++l) {
        if (!level_feats[l] || !level_weights[l] || level_side[l] < 1 || level_channels[l] < 1)
            return set_error(EG_ERR_ARG, "NULL level map / weight or bad side / channel count");
        const int64_t P = (int64_t)level_side[l] * level_side[l];
        a.feat[l] = level_feats[l]; a.w[l] = level_weights[l]; a.bias[l] = level_biases ? level_biases[l] : nullptr;
        a.cin[l] = level_channels[l]; a.side[l] = level_side[l]; a.row0[l] = (int)row; a.tile0[l] = tiles;
        row += P;
        tiles += (int)((P + PK_POS - 1) / PK_POS);
    }
    if (row > n_rows) return set_error(EG_ERR_ARG, "levels do not fit the frame's rows");
    a.tile0[n_levels] = tiles;
    hipLaunchKernelGGL(k_conv1x1_relu_pack, dim3((unsigned)(batch * tiles)), dim3(PK_THREADS), 0, (hipStream_t)stream, a, nodes);
    EG_HIP_TRY(hipGetLastError());
    return EG_OK;
}

}  // extern "C"
