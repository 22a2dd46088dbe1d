"""-m gpu: 'grid-diagonal' levels (8-neighbour grids, reference src/core/datasets.py:1469-1475, :1494-1500) and connection nodes
(:1450-1456, :1512-1515: a node per aux level wired to every node of it; level sums by a pre-pass, conn.hip) on the implicit-stencil
path.  A topology handle with diagonal levels runs its fused layers on the producer/consumer kernel (seg_wide.h segp_diag_rows)
and everything else on the CSR of one frame it carries; every entry point is compared with a CSR handle built from the oracle's
edge_index for the same graph (the arbitrary-graph path, itself checked against the dense fp64 form in test_gpu_layer.py)."""
import numpy as np
import pytest
import torch

from fixtures_util import synthetic_node_feats
from gpu_util import DEV, assert_param_grads_close, dense_ahat, graph_tensors, model_pair, rand_rows
from oracle import gnn_oracle as O
from echoglad_amd import _lib, ops
from echoglad_amd.topology import HierTopology, TopologySpec

pytestmark = pytest.mark.gpu

# (frame, naux, main_only, coord, diag_main, diag_aux[, connection nodes])
DIAG_CASES = [(16, 3, False, False, True, True), (64, 5, False, False, True, True), (64, 6, False, False, False, True),
              (64, 6, False, False, True, False), (30, 3, False, False, True, True), (17, 3, False, True, True, True),
              (32, 4, False, True, True, True), (16, 2, True, False, True, False), (8, 2, False, False, True, True),
              (100, 5, False, False, True, True)]
CONN = {}          # cases with connection nodes: the flag travels beside the tuple (keeps the parametrisation ids readable)
for _c in ((16, 3, False, False, False, False), (64, 5, False, False, False, False), (64, 5, False, False, True, True),
           (32, 4, False, True, False, False), (8, 2, False, False, False, False), (30, 3, False, True, True, False)):
    CONN[_c] = True


def _types(dm, da):
    return ("grid-diagonal" if dm else "grid"), ("grid-diagonal" if da else "grid")


def _graphs(frame, naux, main_only, coord, dm, da, B, conn=False):
    mt, at = _types(dm, da)
    topo, ei, nt, bi = graph_tensors(frame, naux, B, coord=coord, main_only=main_only, conn=conn, main_type=mt, aux_type=at)
    g = ops.Graph.topo(frame, naux, main_only, coord, use_connection_nodes=conn, diag_main=dm, diag_aux=da)
    c = ops.Graph.csr(ei.to(DEV), B * topo.num_nodes)
    return topo, ei, g, c


@pytest.mark.parametrize("frame,naux,main_only,coord,dm,da", DIAG_CASES + [(224, 7, False, False, True, True), (224, 7, False, True, True, False)])
def test_diagonal_degree_table_and_handle(frame, naux, main_only, coord, dm, da):
    mt, at = _types(dm, da)
    topo = HierTopology(TopologySpec(frame, naux, main_only, coord, False, mt, at))
    assert topo.is_structured()
    g = ops.Graph.topo(frame, naux, main_only, coord, diag_main=dm, diag_aux=da)
    assert g.structured and g.hybrid and g.num_nodes == topo.num_nodes
    assert np.allclose(g.deg_inv_sqrt().cpu().numpy(), topo.deg_inv_sqrt(), rtol=1e-7, atol=0)
    if (frame, naux) in ((224, 7), (64, 5), (64, 6)) and not main_only:
        assert g.kidsum_rows > 0                                   # diagonal graphs keep the chained path
        assert g.fused_classifier_ok == (not coord)


@pytest.mark.parametrize("frame,naux,main_only,coord,dm,da", DIAG_CASES)
def test_diagonal_aggregate_and_fused_layer_vs_csr(frame, naux, main_only, coord, dm, da):
    B = 2
    topo, ei, g, c = _graphs(frame, naux, main_only, coord, dm, da, B)
    rows = B * topo.num_nodes
    x = rand_rows(rows, seed=frame + naux).to(DEV)
    # A_hat x: per-frame CSR of the handle vs the CSR of the whole batch vs (small graphs) the dense fp64 matrix
    a_g, a_c = ops.gcn_aggregate(g, B, x), ops.gcn_aggregate(c, 1, x)
    assert float((a_g - a_c).abs().max()) < 2e-5
    if topo.num_nodes <= 4200:
        A = dense_ahat(topo)
        want = torch.cat([A @ x[b * topo.num_nodes:(b + 1) * topo.num_nodes].double().cpu() for b in range(B)])
        assert float((a_g.double().cpu() - want).abs().max()) < 2e-5
    w = (rand_rows(128, seed=21) * 0.08).to(DEV)
    sc = rand_rows(1, seed=30).to(DEV).reshape(128) * 0.1 + 1.0
    sh = rand_rows(1, seed=31).to(DEV).reshape(128) * 0.1
    for relu, res in ((True, True), (False, True), (True, False)):
        before = g.ps_launches
        got = ops.gcn_layer_fwd(g, B, x, w, sc, sh, x if res else None, relu=relu)
        assert g.ps_launches == before + 1                          # the stencil kernel, not the CSR aggregator
        want = ops.gcn_layer_fwd(c, 1, x, w, sc, sh, x if res else None, relu=relu)
        err = float((got - want).abs().max())
        assert err < 3e-5 * max(1.0, float(want.abs().max())), (relu, res, err)
    assert torch.equal(got, ops.gcn_layer_fwd(g, B, x, w, sc, sh, None, relu=True))              # deterministic
    # transposed weight (the backward's dX form) with a residual of its own
    dy = rand_rows(rows, seed=9).to(DEV)
    got = ops.gcn_layer_fwd(g, B, x, w, None, None, dy, relu=False, transpose_w=True)
    want = ops.gcn_layer_fwd(c, 1, x, w, None, None, dy, relu=False, transpose_w=True)
    assert float((got - want).abs().max()) < 3e-5 * max(1.0, float(want.abs().max()))


@pytest.mark.parametrize("frame,naux,main_only,coord,dm,da", [c for c in DIAG_CASES if not c[2]])
def test_diagonal_chained_layers_and_fused_heads(frame, naux, main_only, coord, dm, da):
    B = 2
    topo, ei, g, c = _graphs(frame, naux, main_only, coord, dm, da, B)
    rows = B * topo.num_nodes
    x = rand_rows(rows, seed=5).to(DEV)
    ws = [rand_rows(128, seed=20 + i).to(DEV) * 0.08 for i in range(3)]
    sc = rand_rows(1, seed=30).to(DEV).reshape(128) * 0.1 + 1.0
    sh = rand_rows(1, seed=31).to(DEV).reshape(128) * 0.1
    want = x
    for i, w in enumerate(ws):
        want = ops.gcn_layer_fwd(c, 1, want, w, sc, sh, want, relu=i < 2)
    if g.kidsum_rows == 0:
        return
    ka, kb = ops.new_kidsum(g, B), ops.new_kidsum(g, B)
    h1 = ops.gcn_layer_fwd(g, B, x, ws[0], sc, sh, x, relu=True, kidsum_out=ka)
    h2 = ops.gcn_layer_fwd(g, B, h1, ws[1], sc, sh, h1, relu=True, kidsum_in=ka, kidsum_out=kb)
    h3 = ops.gcn_layer_fwd(g, B, h2, ws[2], sc, sh, h2, relu=False, kidsum_in=kb)
    assert float((h3 - want).abs().max()) < 3e-5 * max(1.0, float(want.abs().max()))
    if not g.fused_classifier_ok:
        return
    rs = np.random.RandomState(frame)
    f = lambda *shape: torch.from_numpy(rs.uniform(-0.3, 0.3, shape).astype(np.float32)).to(DEV)
    packed = {"w1": f(128, 128), "s1": f(128) + 1.0, "t1": f(128), "w2": f(4, 16, 32), "s2": f(64) + 1.0, "t2": f(64),
              "w3": f(4, 16), "b3": f(4)}
    wl = ops.classifier_fwd(want, B, g.num_nodes, 0, g.num_nodes, packed)
    gl = ops.gcn_layer_cls_fwd(g, B, h2, ws[2], sc, sh, h2, False, packed, False, kidsum_in=kb)
    assert float((gl - wl).abs().max()) < 5e-5 * max(1.0, float(wl.abs().max()))


@pytest.mark.parametrize("frame,naux,coord,dm,da", [(32, 4, True, True, True), (64, 5, False, True, True), (30, 3, False, False, True)])
def test_diagonal_train_composites_vs_csr(frame, naux, coord, dm, da):
    """eg_gcn_layer_train_fwd / eg_gcn_layer_bwd on a diagonal handle (train forms of the stencil kernel, tile-order activation
    pass with child sums) against the same composites on the CSR of the same graph."""
    B = 2
    topo, ei, g, c = _graphs(frame, naux, False, coord, dm, da, B)
    rows = B * topo.num_nodes
    rs = np.random.RandomState(frame)
    W = torch.from_numpy(rs.uniform(-0.15, 0.15, (128, 128)).astype(np.float32)).to(DEV)
    bias = torch.from_numpy(rs.standard_normal(128).astype(np.float32) * 0.1).to(DEV)
    gamma = torch.from_numpy(1 + 0.3 * rs.standard_normal(128).astype(np.float32)).to(DEV)
    beta = torch.from_numpy(0.1 * rs.standard_normal(128).astype(np.float32)).to(DEV)
    x = rand_rows(rows, seed=3).to(DEV)
    args = (W, bias, gamma, beta, None, None, None, 1e-5, True, 0.3, 77, True)
    ka = ops.new_kidsum(g, B) if g.kidsum_rows else None
    out_g, z_g, agg_g, bn_g = ops.gcn_layer_train_fwd(g, B, x, *args, kidsum_out=ka)
    out_c, z_c, agg_c, bn_c = ops.gcn_layer_train_fwd(c, 1, x, *args)
    assert float((agg_g - agg_c).abs().max()) < 2e-5 and float((z_g - z_c).abs().max()) < 1e-4
    assert torch.allclose(bn_g, bn_c, rtol=1e-4, atol=1e-6)
    keep = (z_g - z_c).abs() < 1e-5                                # (ReLU kinks aside)
    assert float(((out_g - out_c).abs() * keep).max()) < 2e-4
    if ka is not None:
        o2g, z2g, a2g, _ = ops.gcn_layer_train_fwd(g, B, out_g, *args, kidsum_in=ka)
        o2c, z2c, a2c, _ = ops.gcn_layer_train_fwd(c, 1, out_g, *args)
        assert float((a2g - a2c).abs().max()) < 3e-5 * max(1.0, float(a2c.abs().max()))
    dy = rand_rows(rows, seed=4).to(DEV)
    gg = ops.gcn_layer_bwd(g.bwd, B, dy, z_g, agg_g, W, gamma, beta, bn_g, True, 0.3, 77, True, True, True)
    gc = ops.gcn_layer_bwd(c.bwd, 1, dy, z_g, agg_g, W, gamma, beta, bn_g, True, 0.3, 77, True, True, True)
    for a, b, name in zip(gg, gc, ("dx", "dw", "db", "dgamma", "dbeta")):
        assert float((a - b).abs().max()) <= 2e-4 * float(b.abs().max()) + 1e-6, name


@pytest.mark.parametrize("frame,naux,coord,dm,da,B", [(16, 3, False, True, True, 2), (64, 5, False, True, True, 2), (32, 4, True, True, False, 2),
                                                     (224, 7, False, True, True, 2)])
def test_model_on_a_diagonal_graph_takes_the_stencil_and_matches_the_oracle(frame, naux, coord, dm, da, B):
    """The model meets the graph type only in the edge_index (it is dataset configuration, datasets.py:1441): the resolver
    recognises the closed form with 'grid-diagonal' levels and routes to the stencil handle; logits equal the oracle's."""
    mt, at = _types(dm, da)
    hip, ref = model_pair(frame, naux, 3, coord=coord, seed=frame + 1)
    topo, ei, nt, bi = graph_tensors(frame, naux, B, coord=coord, main_type=mt, aux_type=at)
    feats = synthetic_node_feats(B * topo.num_nodes, 128, seed=9)
    from fixtures_util import initial_coords
    c0 = initial_coords(B, frame) if coord else None
    with torch.no_grad():
        want, wc = ref.forward_nodes(feats, ei, nt, B, None if c0 is None else c0.clone())
        got, gc = hip.forward_nodes(feats.to(DEV), ei.to(DEV), B, None if c0 is None else c0.clone().to(DEV))
    graph, gb = hip._resolver.resolve(ei.to(DEV), feats.shape[0])
    assert graph.structured and graph.hybrid and gb == B
    assert float((got.cpu() - want).abs().max()) < 1e-4
    assert torch.equal(O.landmark_argmax(got.cpu(), B, frame), O.landmark_argmax(want, B, frame))
    if coord:
        assert float((gc.cpu() - wc).abs().max()) < 2e-4
    # a train step on the same graph: parameter gradients against the oracle's autograd (dropout off)
    for m in list(hip.modules()) + list(ref.modules()):
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    if frame > 64:
        return
    hip.train(); ref.train()
    want, wc = ref.forward_nodes(feats, ei, nt, B, None if c0 is None else c0.clone())
    got, gc = hip.forward_nodes(feats.to(DEV), ei.to(DEV), B, None if c0 is None else c0.clone().to(DEV))
    assert float((got.detach().cpu() - want.detach()).abs().max()) < 2e-4
    ((want ** 2).mean() + (0 if wc is None else (wc ** 2).mean() * 1e-3)).backward()
    ((got ** 2).mean() + (0 if gc is None else (gc ** 2).mean() * 1e-3)).backward()
    assert_param_grads_close(hip, ref)


# ---------------------------------------------------------------------------------------------------------------------------------
# connection nodes
# ---------------------------------------------------------------------------------------------------------------------------------
CONN_CASES = sorted(CONN)


@pytest.mark.parametrize("frame,naux,main_only,coord,dm,da", CONN_CASES + [(224, 7, False, False, False, False)])
def test_connection_node_handle_and_degrees(frame, naux, main_only, coord, dm, da):
    mt, at = _types(dm, da)
    topo = HierTopology(TopologySpec(frame, naux, main_only, coord, True, mt, at))
    assert topo.is_structured() and topo.n_conn == naux + 1
    g = ops.Graph.topo(frame, naux, main_only, coord, use_connection_nodes=True, diag_main=dm, diag_aux=da)
    assert g.structured and g.hybrid and g.num_nodes == topo.num_nodes and g.num_conn == topo.n_conn
    assert g.fused_classifier_ok == (not coord and g.kidsum_rows > 0)           # (the heads' row filter drops the connection rows inside the fused kernel)
    assert np.allclose(g.deg_inv_sqrt().cpu().numpy(), topo.deg_inv_sqrt(), rtol=1e-7, atol=0)
    want = sum(((lv.side + 7) // 8) ** 2 for lv in topo.aux_levels + [topo.main]) + (1 if topo.n_coord else 0) + (topo.n_conn + 7) // 8
    assert _lib.load().eg_graph_num_tiles(g._h) == want


@pytest.mark.parametrize("frame,naux,main_only,coord,dm,da", CONN_CASES)
def test_connection_nodes_fused_layer_and_chain_vs_csr(frame, naux, main_only, coord, dm, da):
    B = 3
    topo, ei, g, c = _graphs(frame, naux, main_only, coord, dm, da, B, conn=True)
    rows = B * topo.num_nodes
    x = rand_rows(rows, seed=frame + naux).to(DEV)
    assert float((ops.gcn_aggregate(g, B, x) - ops.gcn_aggregate(c, 1, x)).abs().max()) < 5e-5        # per-frame CSR of the handle
    ws = [rand_rows(128, seed=20 + i).to(DEV) * 0.08 for i in range(3)]
    sc = rand_rows(1, seed=30).to(DEV).reshape(128) * 0.1 + 1.0
    sh = rand_rows(1, seed=31).to(DEV).reshape(128) * 0.1
    before = g.ps_launches
    got = ops.gcn_layer_fwd(g, B, x, ws[0], sc, sh, x, relu=True)
    assert g.ps_launches == before + 1
    want = ops.gcn_layer_fwd(c, 1, x, ws[0], sc, sh, x, relu=True)
    scale = max(1.0, float(want.abs().max()))
    assert float((got - want).abs().max()) < 5e-5 * scale
    # the connection nodes' own rows (the first naux + 1 of every frame: sums over whole levels) and the levels they feed
    n = topo.num_nodes
    assert float((got.view(B, n, 128)[:, :topo.n_conn] - want.view(B, n, 128)[:, :topo.n_conn]).abs().max()) < 5e-5 * scale
    assert torch.equal(got, ops.gcn_layer_fwd(g, B, x, ws[0], sc, sh, x, relu=True))                    # deterministic
    # frames are independent: frame 1 alone
    one = ops.gcn_layer_fwd(g, 1, x[n:2 * n].contiguous(), ws[0], sc, sh, x[n:2 * n].contiguous(), relu=True)
    assert torch.equal(one, got[n:2 * n])
    want3 = x
    for i, w in enumerate(ws):
        want3 = ops.gcn_layer_fwd(c, 1, want3, w, sc, sh, want3, relu=i < 2)
    if g.kidsum_rows:
        ka, kb = ops.new_kidsum(g, B), ops.new_kidsum(g, B)
        h1 = ops.gcn_layer_fwd(g, B, x, ws[0], sc, sh, x, relu=True, kidsum_out=ka)
        h2 = ops.gcn_layer_fwd(g, B, h1, ws[1], sc, sh, h1, relu=True, kidsum_in=ka, kidsum_out=kb)
        h3 = ops.gcn_layer_fwd(g, B, h2, ws[2], sc, sh, h2, relu=False, kidsum_in=kb)
        assert float((h3 - want3).abs().max()) < 1e-4 * max(1.0, float(want3.abs().max()))
    # the backward's form: transposed weight, a residual of its own
    dy = rand_rows(rows, seed=9).to(DEV)
    gt = ops.gcn_layer_fwd(g, B, x, ws[0], None, None, dy, relu=False, transpose_w=True)
    wt = ops.gcn_layer_fwd(c, 1, x, ws[0], None, None, dy, relu=False, transpose_w=True)
    assert float((gt - wt).abs().max()) < 5e-5 * max(1.0, float(wt.abs().max()))


def test_connection_node_scratch_grows_with_the_batch():
    """The handle's scratch for the level sums starts at 8 frames per launch and grows when a launch brings more (conn.hip)."""
    g = ops.Graph.topo(32, 4, use_connection_nodes=True)
    topo = HierTopology(TopologySpec(32, 4, use_connection_nodes=True))
    w = rand_rows(128, seed=2).to(DEV) * 0.08
    ref = None
    for B in (2, 11, 40, 3):
        x = rand_rows(B * topo.num_nodes, seed=4).to(DEV)
        got = ops.gcn_layer_fwd(g, B, x, w, None, None, x, relu=True)
        one = ops.gcn_layer_fwd(g, 1, x[:topo.num_nodes].contiguous(), w, None, None, x[:topo.num_nodes].contiguous(), relu=True)
        assert torch.equal(got[:topo.num_nodes], one)
        ref = one if ref is None else ref
        assert torch.equal(one, ref)                                   # (rand_rows(seed) starts every batch with the same frame)


@pytest.mark.parametrize("frame,naux,coord,dm,da,B", [(16, 3, False, False, False, 2), (64, 5, True, False, False, 2), (64, 5, False, True, True, 2),
                                                     (224, 7, False, False, False, 2)])
def test_model_with_connection_nodes_takes_the_stencil_and_matches_the_oracle(frame, naux, coord, dm, da, B):
    mt, at = _types(dm, da)
    hip, ref = model_pair(frame, naux, 3, coord=coord, seed=frame + 2, use_connection_nodes=True)
    topo, ei, nt, bi = graph_tensors(frame, naux, B, coord=coord, conn=True, main_type=mt, aux_type=at)
    feats = synthetic_node_feats(B * topo.num_nodes, 128, seed=9)
    from fixtures_util import initial_coords
    c0 = initial_coords(B, frame) if coord else None
    with torch.no_grad():
        want, wc = ref.forward_nodes(feats, ei, nt, B, None if c0 is None else c0.clone())
        got, gc = hip.forward_nodes(feats.to(DEV), ei.to(DEV), B, None if c0 is None else c0.clone().to(DEV))
    graph, gb = hip._resolver.resolve(ei.to(DEV), feats.shape[0])
    assert graph.structured and graph.hybrid and gb == B
    assert got.shape == want.shape == (B * topo.num_valid_nodes, 4)
    assert float((got.cpu() - want).abs().max()) < 1e-4
    assert torch.equal(O.landmark_argmax(got.cpu(), B, frame), O.landmark_argmax(want, B, frame))
    if frame > 64:
        return
    for m in list(hip.modules()) + list(ref.modules()):
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    hip.train(); ref.train()
    want, wc = ref.forward_nodes(feats, ei, nt, B, None if c0 is None else c0.clone())
    got, gc = hip.forward_nodes(feats.to(DEV), ei.to(DEV), B, None if c0 is None else c0.clone().to(DEV))
    assert float((got.detach().cpu() - want.detach()).abs().max()) < 2e-4
    ((want ** 2).mean() + (0 if wc is None else (wc ** 2).mean() * 1e-3)).backward()
    ((got ** 2).mean() + (0 if gc is None else (gc ** 2).mean() * 1e-3)).backward()
    assert_param_grads_close(hip, ref)


@pytest.mark.parametrize("frame,naux,dm,da,B", [(64, 5, False, False, 3), (224, 7, False, False, 2), (64, 5, True, True, 2)])
def test_fused_classifier_on_a_connection_node_handle(frame, naux, dm, da, B):
    """eg_gcn_layer_cls_fwd on a handle with connection nodes: the heads' node-type filter (the first naux + 1 rows of a frame
    have no logits row) inside the fused last-layer kernel == the layer kernel + eg_classifier_fwd with the row range."""
    mt, at = _types(dm, da)
    hip, _ = model_pair(frame, naux, 3, seed=frame + 5, use_connection_nodes=True)
    topo, ei, nt, bi = graph_tensors(frame, naux, B, conn=True, main_type=mt, aux_type=at)
    feats = synthetic_node_feats(B * topo.num_nodes, 128, seed=4).to(DEV)
    graph, _ = hip._resolver.resolve(ei.to(DEV), feats.shape[0])
    assert graph.num_conn == naux + 1 and graph.fused_classifier_ok
    outs = {}
    for fuse in (True, False):
        hip.fuse_classifier = fuse
        with torch.no_grad():
            outs[fuse] = hip.forward_nodes(feats, ei.to(DEV), B)[0]
    assert outs[True].shape == outs[False].shape == (B * topo.num_valid_nodes, 4)
    assert float((outs[True] - outs[False]).abs().max()) < 2e-5 * max(1.0, float(outs[False].abs().max()))
