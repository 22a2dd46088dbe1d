// Host-side check of the C ABI under AddressSanitizer + UBSan (CPU only; built and run by tests/test_abi_sanitized.py).
// Drives what the library does on the HOST: argument validation of the entry points and the topology-table construction of
// eg_topo_create (closed-form level / tile / segment / weight-pattern tables, graph.hip), whose indexing is where a host-side
// out-of-bounds would hide.  Without a GPU the device allocation at the end of eg_topo_create fails and the error path frees
// everything; with one the handle is created and destroyed.  Exit code 0 = every call returned what it should and the
// sanitizers stayed silent (they abort the process otherwise).
#include <cstdio>
#include <cstring>
#include <cstdint>

#include "../../include/echoglad_hip.h"

static int failures = 0;
#define EXPECT(cond, what) do { if (!(cond)) { std::printf("FAIL: %s (%s)\n", what, eg_last_error()); ++failures; } } while (0)

int main() {
    EXPECT(eg_version() > 0, "eg_version");
    // ---- argument validation: NULL / out-of-range arguments must come back as EG_ERR_ARG, never crash
    eg_graph* g = nullptr;
    EXPECT(eg_topo_create(224, 7, 0, 0, 0, 0, 0, nullptr) == EG_ERR_ARG, "eg_topo_create(out = NULL)");
    EXPECT(eg_topo_create(0, 7, 0, 0, 0, 0, 0, &g) != EG_OK && g == nullptr, "eg_topo_create(frame = 0)");
    EXPECT(eg_topo_create(224, 99, 0, 0, 0, 0, 0, &g) != EG_OK && g == nullptr, "eg_topo_create(naux = 99)");
    EXPECT(eg_topo_create(-5, 3, 0, 0, 0, 0, 0, &g) != EG_OK && g == nullptr, "eg_topo_create(frame < 0)");
    EXPECT(eg_graph_destroy(nullptr) == EG_OK, "eg_graph_destroy(NULL)");
    EXPECT(eg_graph_num_nodes(nullptr) == -1, "eg_graph_num_nodes(NULL)");
    EXPECT(eg_graph_kidsum_rows(nullptr) == 0, "eg_graph_kidsum_rows(NULL)");
    EXPECT(eg_graph_fused_classifier_ok(nullptr) == 0, "eg_graph_fused_classifier_ok(NULL)");
    float dummy[4] = {0, 0, 0, 0};
    EXPECT(eg_gcn_layer_fwd(nullptr, 1, dummy, dummy, nullptr, nullptr, nullptr, 0, 0, dummy + 1, nullptr) == EG_ERR_ARG, "eg_gcn_layer_fwd(g = NULL)");
    EXPECT(eg_gcn_layer_fwd(nullptr, 1, nullptr, dummy, nullptr, nullptr, nullptr, 0, 0, dummy, nullptr) == EG_ERR_ARG, "eg_gcn_layer_fwd(x = NULL)");
    EXPECT(eg_linear128_fwd(dummy, -1, dummy, nullptr, nullptr, nullptr, 0, 0, dummy + 1, nullptr) == EG_ERR_ARG, "eg_linear128_fwd(rows < 0)");
    EXPECT(eg_linear128_fwd(dummy, 4, dummy, nullptr, nullptr, nullptr, 0, 0, dummy, nullptr) == EG_ERR_ARG, "eg_linear128_fwd(out aliases x)");
    EXPECT(eg_gcn_aggregate(nullptr, 1, dummy, dummy + 1, nullptr) == EG_ERR_ARG, "eg_gcn_aggregate(g = NULL)");
    EXPECT(eg_colsum128(nullptr, 4, dummy, dummy, nullptr) == EG_ERR_ARG, "eg_colsum128(x = NULL)");
    EXPECT(eg_bn_act_fwd(dummy, 4, dummy, dummy, nullptr, 0, 1.5f, 0, dummy, nullptr) == EG_ERR_ARG, "eg_bn_act_fwd(p = 1.5)");
    EXPECT(eg_pack_levels(nullptr, nullptr, 0, 1, 1, 0, dummy, nullptr) == EG_ERR_ARG, "eg_pack_levels(no levels)");
    {
        const float* maps[1] = {dummy};
        const int side17[1] = {17};
        EXPECT(eg_pack_levels(maps, side17, 1, 1, 16, 0, dummy, nullptr) == EG_ERR_ARG, "eg_pack_levels(level larger than the frame's rows)");
        const int chans[1] = {0};
        EXPECT(eg_conv1x1_relu_pack_levels(maps, maps, nullptr, chans, side17, 1, 1, 289, 0, dummy, nullptr) == EG_ERR_ARG,
               "eg_conv1x1_relu_pack_levels(0 input channels)");
    }
    {   // round 6's entry points: everything that can be refused on the host is refused before a launch
        int sides[3] = {2, 4, 8}, bad_order[2] = {8, 4}, too_big[1] = {64};
        float* outs[3] = {dummy, dummy, dummy};
        const float* grads[3] = {dummy, dummy, dummy};
        EXPECT(eg_avg_pool_pyramid_fwd(nullptr, 4, 16, sides, 3, outs, nullptr) == EG_ERR_ARG, "eg_avg_pool_pyramid_fwd(x = NULL)");
        EXPECT(eg_avg_pool_pyramid_fwd(dummy, 4, 16, bad_order, 2, outs, nullptr) == EG_ERR_ARG, "eg_avg_pool_pyramid_fwd(sides not ascending)");
        EXPECT(eg_avg_pool_pyramid_fwd(dummy, 4, 16, too_big, 1, outs, nullptr) == EG_ERR_ARG, "eg_avg_pool_pyramid_fwd(side > frame)");
        EXPECT(eg_avg_pool_pyramid_fwd(dummy, 0, 16, sides, 3, outs, nullptr) == EG_ERR_ARG, "eg_avg_pool_pyramid_fwd(no planes)");
        EXPECT(eg_avg_pool_pyramid_bwd(grads, nullptr, 4, 1024, sides, 3, dummy, nullptr) == EG_ERR_ARG, "eg_avg_pool_pyramid_bwd(frame > 512)");
        EXPECT(eg_avg_pool_pyramid_bwd(grads, nullptr, 4, 16, sides, 3, nullptr, nullptr) == EG_ERR_ARG, "eg_avg_pool_pyramid_bwd(dx = NULL)");
        EXPECT(eg_criteria_workspace_bytes(2, sides, 3) > 0 && eg_criteria_workspace_bytes(0, sides, 3) == 0, "eg_criteria_workspace_bytes");
        int start[3] = {0, 4, 20};
        EXPECT(eg_criteria_fwd(dummy, dummy, nullptr, 1, 84, start, sides, 3, dummy, 9000.f, 1.f, 10.f, nullptr, nullptr, 0, 1.f, dummy, dummy, dummy,
                               dummy, nullptr, dummy, dummy, dummy, dummy, nullptr, nullptr) == EG_ERR_ARG, "eg_criteria_fwd(valid = NULL)");
        EXPECT(eg_criteria_fwd(dummy, dummy, dummy, 1, 84, start, sides, 3, dummy, 9000.f, 1.f, 10.f, dummy, nullptr, 8, 1.f, dummy, dummy, dummy,
                               dummy, dummy, dummy, dummy, dummy, dummy, dummy, nullptr) == EG_ERR_ARG, "eg_criteria_fwd(coord_pred without coord_y)");
        EXPECT(eg_criteria_bwd(dummy, dummy, dummy, 1, 10, start, sides, 3, 9000.f, dummy, dummy, dummy, dummy, nullptr, 0, nullptr, nullptr, nullptr,
                               nullptr, dummy, nullptr, nullptr) == EG_ERR_ARG, "eg_criteria_bwd(levels outside the frame's rows)");
        eg_lower_sums ls{};
        EXPECT(eg_gcn_layer_bwd_lower(nullptr, 1, dummy, dummy, dummy, dummy, dummy, dummy, dummy, 1, 0.f, 0, 1, dummy, dummy, dummy, dummy, dummy, dummy,
                                      dummy, nullptr, &ls, nullptr) == EG_ERR_ARG, "eg_gcn_layer_bwd_lower(g = NULL)");
        EXPECT(eg_bilinear4_bwd_rows_sums(dummy, 512, dummy, dummy, 1, 4, 100, 0, 8, dummy, dummy, &ls, dummy, nullptr) == EG_ERR_ARG,
               "eg_bilinear4_bwd_rows_sums(incomplete eg_lower_sums)");
        eg_cls_train_params cp{};
        EXPECT(eg_coord_update_fwd(nullptr, 100, 96, 0, dummy, 1, &cp, 1, 8, 1, dummy, dummy, dummy, dummy, dummy, dummy, dummy, nullptr) == EG_ERR_ARG,
               "eg_coord_update_fwd(h = NULL)");
        EXPECT(eg_coord_update_fwd(dummy, 100, 98, 0, dummy, 1, &cp, 1, 8, 1, dummy, dummy, dummy, dummy, dummy, dummy, dummy, nullptr) == EG_ERR_ARG,
               "eg_coord_update_fwd(coordinate rows outside the frame)");
        EXPECT(eg_coord_update_fwd(dummy, 100, 62, 0, dummy, 1, &cp, 1, 8, 1, dummy, dummy, dummy, dummy, dummy, dummy, dummy, nullptr) == EG_ERR_ARG,
               "eg_coord_update_fwd(coordinate rows inside the main grid)");
        EXPECT(eg_coord_update_bwd(dummy, 100, 96, 0, dummy, dummy, nullptr, dummy, dummy, 1, &cp, 8, dummy, dummy, dummy, dummy, dummy, nullptr, nullptr, nullptr,
                                   dummy, dummy, nullptr) == EG_ERR_ARG, "eg_coord_update_bwd(dbil = NULL)");
        EXPECT(eg_coord_update_bwd(dummy, 100, 96, 0, dummy, dummy, nullptr, dummy, dummy, 1, &cp, 8, dummy, dummy, dummy, dummy, dummy, dummy, &ls, nullptr,
                                   dummy, dummy, nullptr) == EG_ERR_ARG, "eg_coord_update_bwd(incomplete eg_lower_sums)");
        eg_adam_tensor at{dummy, dummy, dummy, nullptr, 16};
        EXPECT(eg_adam_step(&at, 1, dummy, 1e-3f, nullptr, 0.9f, 0.999f, 1e-8f, 0.f, 0, nullptr) == EG_ERR_ARG, "eg_adam_step(exp_avg_sq = NULL)");
        at.exp_avg_sq = dummy;
        EXPECT(eg_adam_step(&at, 1, dummy, 1e-3f, nullptr, 1.0f, 0.999f, 1e-8f, 0.f, 0, nullptr) == EG_ERR_ARG, "eg_adam_step(beta1 = 1)");
        EXPECT(eg_adam_step(&at, 97, dummy, 1e-3f, nullptr, 0.9f, 0.999f, 1e-8f, 0.f, 0, nullptr) == EG_ERR_UNSUPPORTED, "eg_adam_step(97 tensors)");
        EXPECT(eg_adam_step(&at, 1, nullptr, 1e-3f, nullptr, 0.9f, 0.999f, 1e-8f, 0.f, 0, nullptr) == EG_ERR_ARG, "eg_adam_step(steps = NULL)");
    }
    EXPECT(eg_workspace_bytes() > 0, "eg_workspace_bytes");
    EXPECT(std::strlen(eg_last_error()) > 0, "eg_last_error carries the last message");
    // ---- topology tables: every BASELINE shape and the odd ones of the parity tests (host tables are complete before the first
    // device call; a box without a GPU then reports the HIP error and must have freed everything)
    // {frame, naux, main_only, coordinate nodes, grid-diagonal main, grid-diagonal aux, connection nodes}
    const int cfgs[][7] = {{224, 7, 0, 0, 0, 0, 0}, {224, 7, 0, 1, 0, 0, 0}, {224, 7, 1, 0, 0, 0, 0}, {448, 8, 0, 0, 0, 0, 0}, {448, 7, 0, 0, 0, 0, 0},
                           {64, 2, 0, 0, 0, 0, 0}, {64, 6, 0, 0, 0, 0, 0}, {30, 3, 0, 0, 0, 0, 0}, {17, 3, 0, 0, 0, 0, 0}, {8, 1, 0, 0, 0, 0, 0},
                           {8, 2, 0, 0, 0, 0, 0}, {16, 2, 1, 0, 0, 0, 0}, {32, 4, 0, 1, 0, 0, 0}, {16, 3, 0, 1, 0, 0, 0},
                           // 'grid-diagonal' levels (stencil tables + the per-frame CSR of such handles)
                           {224, 7, 0, 0, 1, 1, 0}, {224, 7, 0, 1, 1, 0, 0}, {224, 7, 1, 0, 1, 0, 0}, {64, 5, 0, 0, 0, 1, 0}, {30, 3, 0, 0, 1, 1, 0},
                           {17, 3, 0, 1, 1, 1, 0}, {8, 2, 0, 0, 1, 1, 0}, {448, 8, 0, 0, 1, 1, 0},
                           // connection nodes (pre-pass tables, hub pseudo-tiles), alone and with everything else
                           {224, 7, 0, 0, 0, 0, 1}, {224, 7, 0, 1, 1, 1, 1}, {64, 5, 0, 0, 0, 0, 1}, {16, 3, 0, 1, 0, 0, 1}, {448, 8, 0, 0, 0, 0, 1},
                           {8, 2, 0, 0, 0, 0, 1}, {16, 2, 1, 0, 0, 0, 1}};
    int created = 0;
    for (const auto& c : cfgs) {
        g = nullptr;
        const int rc = eg_topo_create(c[0], c[1], c[2], c[3], c[6], c[4], c[5], &g);
        if (rc == EG_OK) {
            ++created;
            EXPECT(g != nullptr && eg_graph_num_nodes(g) > 0 && eg_graph_num_tiles(g) > 0, "eg_topo_create handle");
            EXPECT(eg_graph_is_structured(g) == 1 && eg_graph_is_symmetric(g) == 1, "topology handle flags");
            EXPECT(eg_graph_destroy(g) == EG_OK, "eg_graph_destroy");
        } else {
            EXPECT(rc == EG_ERR_HIP && g == nullptr, "eg_topo_create without a device: EG_ERR_HIP and no handle");
        }
    }
    std::printf("abi_host_check: %d failure(s), %d of %zu topology handles created (0 without a GPU)\n", failures, created,
                sizeof(cfgs) / sizeof(cfgs[0]));
    return failures ? 1 : 0;
}
