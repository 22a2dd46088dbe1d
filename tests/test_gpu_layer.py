"""-m gpu: the HIP kernels, called through the C-ABI, against the CPU oracle."""
import numpy as np
import pytest
import torch

from gpu_util import DEV, dense_ahat, graph_tensors, rand_rows
from oracle import gnn_oracle as O
from echoglad_amd import _lib, ops
from echoglad_amd.topology import HierTopology, TopologySpec, commutative_edge_hash

pytestmark = pytest.mark.gpu

TOPO_CASES = [  # frame, naux, main_only, coord
    (8, 2, False, False), (16, 3, False, False), (16, 3, False, True), (16, 2, True, False),
    (64, 2, False, False), (30, 3, False, False), (17, 3, False, False), (8, 1, False, False),
    (32, 4, False, True), (64, 6, False, False),
]


def test_library_is_loaded_and_not_a_fallback():
    lib = _lib.load()
    assert lib.eg_version() >= 100
    assert torch.cuda.is_available()
    with pytest.raises(RuntimeError):
        ops.gcn_aggregate(ops.Graph.topo(8, 2), 1, torch.zeros(84, 128))      # CPU tensor: loud failure


@pytest.mark.parametrize("frame,naux,main_only,coord", TOPO_CASES + [(224, 7, False, False), (224, 7, True, False),
                                                                      (224, 7, False, True)])
def test_closed_form_degree_table(frame, naux, main_only, coord):
    topo = HierTopology(TopologySpec(frame, naux, main_only, coord))
    g = ops.Graph.topo(frame, naux, main_only, coord)
    assert g.num_nodes == topo.num_nodes
    dis = g.deg_inv_sqrt().cpu().numpy()
    assert np.allclose(dis, topo.deg_inv_sqrt(), rtol=1e-7, atol=0)
    # the 8x8 patch table must hold every patch of every level exactly once (+1 for the coordinate nodes)
    want = sum(((lv.side + 7) // 8) ** 2 for lv in topo.aux_levels + [topo.main]) + (1 if topo.n_coord else 0)
    assert _lib.load().eg_graph_num_tiles(g._h) == want


@pytest.mark.parametrize("frame,naux,main_only,coord", TOPO_CASES)
def test_csr_degree_table_and_edge_hash(frame, naux, main_only, coord):
    topo = HierTopology(TopologySpec(frame, naux, main_only, coord))
    ei = torch.from_numpy(topo.batched_edge_index(2)).to(DEV)
    g = ops.Graph.csr(ei, 2 * topo.num_nodes)
    dis = g.deg_inv_sqrt().cpu().numpy()
    assert np.allclose(dis, np.tile(topo.deg_inv_sqrt(), 2), rtol=1e-6)
    assert ops.edge_hash(ei) == commutative_edge_hash(topo.batched_edge_index(2))


@pytest.mark.parametrize("frame,naux,main_only,coord", TOPO_CASES)
def test_aggregate_stencil_vs_csr_vs_dense(frame, naux, main_only, coord):
    B = 3
    topo, ei, _, _ = graph_tensors(frame, naux, B, coord, main_only)
    x = rand_rows(B * topo.num_nodes, seed=frame + naux)
    xg = x.to(DEV)
    s = ops.gcn_aggregate(ops.Graph.topo(frame, naux, main_only, coord), B, xg).cpu()
    c = ops.gcn_aggregate(ops.Graph.csr(ei.to(DEV), B * topo.num_nodes), 1, xg).cpu()
    # per-frame CSR handle replicated over the batch
    c1 = ops.gcn_aggregate(ops.Graph.csr(torch.from_numpy(topo.edge_index()).to(DEV), topo.num_nodes), B, xg).cpu()
    assert torch.isfinite(s).all()
    assert (s - c).abs().max() < 2e-6
    assert (c1 - c).abs().max() < 2e-6
    if topo.num_nodes <= 4200:
        a = dense_ahat(topo)
        want = torch.cat([a @ x[b * topo.num_nodes:(b + 1) * topo.num_nodes].double() for b in range(B)])
        assert (s.double() - want).abs().max() < 5e-6


def test_aggregate_csr_generic_graphs():
    """connection nodes (huge degree), diagonal grids, self loops and duplicate-free random graphs."""
    for spec in (TopologySpec(16, 3, use_connection_nodes=True), TopologySpec(16, 3, main_graph_type="grid-diagonal",
                                                                              aux_graph_type="grid-diagonal")):
        topo = HierTopology(spec)
        ei = torch.from_numpy(topo.edge_index())
        x = rand_rows(topo.num_nodes, seed=5)
        got = ops.gcn_aggregate(ops.Graph.csr(ei.to(DEV), topo.num_nodes), 1, x.to(DEV)).cpu()
        want = O.gcn_conv_sparse(x, ei, torch.eye(128), None)
        assert (got - want).abs().max() < 5e-6
    # explicit self loops must be replaced, isolated nodes keep x_i
    ei = torch.tensor([[0, 0, 1, 2, 2], [0, 1, 0, 2, 1]])
    x = rand_rows(5, seed=6)
    got = ops.gcn_aggregate(ops.Graph.csr(ei.to(DEV), 5), 1, x.to(DEV)).cpu()
    want = O.gcn_conv_sparse(x, ei, torch.eye(128), None)
    assert (got - want).abs().max() < 1e-6
    assert torch.allclose(got[3:], x[3:])


def test_gcn_conv_literal_known_answer_on_the_csr_kernel():
    """The hand-computed 6-node case (fixtures_util.gcn_known_answer: existing self loop, isolated node, one-directional
    edge, duplicate edge) through eg_csr_create + the fused layer kernel, the aggregation kernel and nn.GCNConv."""
    from fixtures_util import gcn_known_answer
    from echoglad_amd import nn as egnn
    ei, x, w, b, want, dis = gcn_known_answer()
    g = ops.Graph.csr(ei.to(DEV), 6)
    assert (g.deg_inv_sqrt().cpu().double() - dis).abs().max() < 1e-7
    got = ops.gcn_layer_fwd(g, 1, x.to(DEV), w.to(DEV), None, b.to(DEV), None, False)
    assert (got.cpu().double() - want).abs().max() < 1e-6
    agg = ops.gcn_aggregate(g, 1, x.to(DEV)).cpu().double()          # A_hat x: channel 0 of x is n + 1
    a_hat_x0 = torch.tensor([0.5 * 1 + 0.35355339 * 2, 0.70710678 * 1 + 0.25 * 2 + 0.35355339 * 3, 0.35355339 * 2 + 0.5 * 3,
                             4.0, 0.70710678 * 4 + 0.5 * 5, 6.0], dtype=torch.float64)
    assert (agg[:, 0] - a_hat_x0).abs().max() < 1e-6
    conv = egnn.GCNConv(128, 128)
    with torch.no_grad():
        conv.lin.weight.copy_(w); conv.bias.copy_(b)
    conv = conv.to(DEV)
    with torch.no_grad():
        got2 = conv(x.to(DEV), ei.to(DEV))
    assert (got2.cpu().double() - want).abs().max() < 1e-6


@pytest.mark.parametrize("n", [3, 7, 64, 130, 1000])
def test_gcn_conv_closed_form_families_on_the_csr_kernels(n):
    """Complete graph, cycle, star (fixtures_util.gcn_closed_form_families: A_hat written down in closed form, no aggregation
    code involved) through eg_csr_create + the fused layer kernel, the aggregation kernel, nn.GCNConv -- and, 2 frames batched,
    the same graph repeated per frame.  n = 130 / 1000: several 64-row tiles, rows of degree n - 1 next to rows of degree 1."""
    from fixtures_util import gcn_closed_form_families
    from echoglad_amd import nn as egnn
    rs = np.random.RandomState(n)
    x = torch.from_numpy(rs.standard_normal((n, 128)).astype(np.float32))
    w = torch.from_numpy(rs.uniform(-0.2, 0.2, (128, 128)).astype(np.float32))
    b = torch.from_numpy(rs.standard_normal(128).astype(np.float32))
    tol = 2e-5 * max(1.0, float(n) ** 0.5 / 8)
    for name, ei, a_hat in gcn_closed_form_families(n, seed=n):
        if name == "complete" and n > 200:
            continue                                                  # (10^6 edges of K_1000: nothing a smaller K_n does not show)
        want = a_hat(x.double() @ w.double().t()) + b.double()
        g = ops.Graph.csr(ei.to(DEV), n)
        got = ops.gcn_layer_fwd(g, 1, x.to(DEV), w.to(DEV), None, b.to(DEV), None, False)
        assert (got.cpu().double() - want).abs().max() < tol, name
        agg = ops.gcn_aggregate(g, 1, x.to(DEV)).cpu().double()
        assert (agg - a_hat(x.double())).abs().max() < tol, name
        conv = egnn.GCNConv(128, 128)
        with torch.no_grad():
            conv.lin.weight.copy_(w); conv.bias.copy_(b)
        conv = conv.to(DEV)
        with torch.no_grad():
            got2 = conv(x.to(DEV), ei.to(DEV))
        assert (got2.cpu().double() - want).abs().max() < tol, name
        x2 = torch.cat([x, -2.0 * x])                               # two frames on one handle: frame 1 = -2 x frame 0
        got3 = ops.gcn_layer_fwd(g, 2, x2.to(DEV), w.to(DEV), None, None, None, False).cpu().double()
        base = a_hat(x.double() @ w.double().t())
        assert (got3[:n] - base).abs().max() < tol and (got3[n:] + 2.0 * base).abs().max() < 2 * tol, name


@pytest.mark.parametrize("rows", [1, 31, 32, 33, 64, 65, 200, 4116])
@pytest.mark.parametrize("transpose", [False, True])
def test_linear128(rows, transpose):
    rs = np.random.RandomState(rows)
    x = rand_rows(rows, seed=rows)
    w = torch.from_numpy(rs.uniform(-0.2, 0.2, (128, 128)).astype(np.float32))
    scale = torch.from_numpy(rs.uniform(0.5, 1.5, 128).astype(np.float32))
    shift = torch.from_numpy(rs.standard_normal(128).astype(np.float32))
    res = rand_rows(rows, seed=rows + 1)
    got = ops.linear128_fwd(x.to(DEV), w.to(DEV), scale.to(DEV), shift.to(DEV), res.to(DEV), relu=True,
                            transpose_w=transpose).cpu()
    wm = w if transpose else w.t()
    want = torch.relu((x.double() @ wm.double()) * scale.double() + shift.double()) + res.double()
    assert (got.double() - want).abs().max() < 2e-5
    plain = ops.linear128_fwd(x.to(DEV), w.to(DEV)).cpu()
    assert (plain.double() - x.double() @ (w.double() if transpose is None else w.t().double())).abs().max() < 2e-5


def test_linear128_distinguishes_rows_and_columns():
    """A = I check with an asymmetric weight (catches a transposed accumulator map)."""
    w = torch.arange(128 * 128, dtype=torch.float32).reshape(128, 128) / 1000.0
    x = torch.eye(128)
    got = ops.linear128_fwd(x.to(DEV), w.to(DEV)).cpu()
    assert torch.equal(got, w.t())
    got_t = ops.linear128_fwd(x.to(DEV), w.to(DEV), transpose_w=True).cpu()
    assert torch.equal(got_t, w)


@pytest.mark.parametrize("frame,naux,main_only,coord", TOPO_CASES)
@pytest.mark.parametrize("relu", [False, True])
def test_fused_layer_vs_oracle(frame, naux, main_only, coord, relu):
    B = 2
    topo, ei, _, _ = graph_tensors(frame, naux, B, coord, main_only)
    rows = B * topo.num_nodes
    rs = np.random.RandomState(frame * 7 + naux)
    x = rand_rows(rows, seed=11)
    w = torch.from_numpy(rs.uniform(-0.15, 0.15, (128, 128)).astype(np.float32))
    b = torch.from_numpy((0.1 * rs.standard_normal(128)).astype(np.float32))
    bn = torch.nn.BatchNorm1d(128).eval()
    O.randomize_bn_stats(bn, seed=frame)
    with torch.no_grad():
        want = bn(O.gcn_conv_sparse(x, ei, w, b))
        want = (torch.relu(want) if relu else want) + x
        scale = bn.weight / torch.sqrt(bn.running_var + bn.eps)
        shift = bn.bias - bn.running_mean * scale + b * scale
    xg = x.to(DEV)
    for graph, gb in ((ops.Graph.topo(frame, naux, main_only, coord), B),
                      (ops.Graph.csr(ei.to(DEV), rows), 1)):
        got = ops.gcn_layer_fwd(graph, gb, xg, w.to(DEV), scale.to(DEV), shift.to(DEV), xg, relu=relu).cpu()
        assert (got - want).abs().max() < 3e-5, (graph.structured, float((got - want).abs().max()))


CHAIN_CASES = TOPO_CASES + [(224, 7, False, False), (128, 5, False, False), (100, 5, False, False), (200, 6, False, False),
                            (112, 6, False, True)]


@pytest.mark.parametrize("frame,naux,main_only,coord", CHAIN_CASES)
def test_chained_layers_match_separate_layers(frame, naux, main_only, coord):
    """eg_gcn_layer_fwd_chain (child sums handed from layer to layer) == the same layers run one by one."""
    B = 2
    g = ops.Graph.topo(frame, naux, main_only, coord)
    if (frame, naux) in ((224, 7), (64, 6), (128, 5)):
        assert g.kidsum_rows > 0                                   # the benchmark topology must take the chained path
    rows = B * g.num_nodes
    x = rand_rows(rows, seed=5).to(DEV)
    ws = [rand_rows(128, seed=20 + i).to(DEV) * 0.08 for i in range(3)]
    sc = rand_rows(1, seed=30).to(DEV).reshape(128) * 0.1 + 1.0
    sh = rand_rows(1, seed=31).to(DEV).reshape(128) * 0.1
    want = x
    for i, w in enumerate(ws):
        want = ops.gcn_layer_fwd(g, B, want, w, sc, sh, want, relu=i < 2)
    if g.kidsum_rows == 0:
        with pytest.raises(RuntimeError):
            ops.gcn_layer_fwd(g, B, x, ws[0], sc, sh, x, relu=True, kidsum_out=torch.zeros(8, 128, device=DEV))
        return
    ka, kb = ops.new_kidsum(g, B), ops.new_kidsum(g, B)
    for rep in range(2):                                           # second pass: buffers are reused as they are
        h1 = ops.gcn_layer_fwd(g, B, x, ws[0], sc, sh, x, relu=True, kidsum_out=ka)
        h2 = ops.gcn_layer_fwd(g, B, h1, ws[1], sc, sh, h1, relu=True, kidsum_in=ka, kidsum_out=kb)
        h3 = ops.gcn_layer_fwd(g, B, h2, ws[2], sc, sh, h2, relu=False, kidsum_in=kb)
        err = float((h3 - want).abs().max())
        assert err < 2e-5 * max(1.0, float(want.abs().max())), err
    # the side buffer itself: sum over the four children of dis[c] * h1[c]
    topo = HierTopology(TopologySpec(frame, naux, main_only, coord))
    dis = g.deg_inv_sqrt().double()
    ei = torch.from_numpy(topo.edge_index()).to(DEV)
    bases = torch.from_numpy(topo.level_table()[:, 0].astype(np.int64)).to(DEV)
    lvl = torch.bucketize(torch.arange(g.num_nodes, device=DEV), bases, right=True)       # 1-based level of each grid node
    src, dst = ei[0], ei[1]
    child_edges = (lvl[src] == lvl[dst] + 1) & (src < topo.coord_base)                   # src one level below dst
    want_k = torch.zeros(B, g.num_nodes, 128, dtype=torch.float64, device=DEV)
    h1v = h1.view(B, g.num_nodes, 128).double()
    want_k.index_add_(1, dst[child_edges], h1v[:, src[child_edges], :] * dis[src[child_edges]][None, :, None])
    got_k = ka.view(B, g.kidsum_rows, 128).double()
    assert float((got_k - want_k[:, :g.kidsum_rows]).abs().max()) < 1e-5


@pytest.mark.parametrize("frame,naux,main_only,coord", CHAIN_CASES)
@pytest.mark.parametrize("sigmoid", [False, True])
def test_last_layer_with_fused_classifier_matches_separate_kernels(frame, naux, main_only, coord, sigmoid):
    """eg_gcn_layer_cls_fwd == eg_gcn_layer_fwd followed by eg_classifier_fwd (both oracle-checked elsewhere)."""
    B = 2
    g = ops.Graph.topo(frame, naux, main_only, coord)
    rs = np.random.RandomState(frame + naux)
    rows = B * g.num_nodes
    x = rand_rows(rows, seed=9).to(DEV)
    w = rand_rows(128, seed=21).to(DEV) * 0.08
    sc = rand_rows(1, seed=30).to(DEV).reshape(128) * 0.1 + 1.0
    sh = rand_rows(1, seed=31).to(DEV).reshape(128) * 0.1
    f = lambda *shape: torch.from_numpy(rs.uniform(-0.3, 0.3, shape).astype(np.float32)).to(DEV)
    packed = {"w1": f(128, 128), "s1": f(128) + 1.0, "t1": f(128), "w2": f(4, 16, 32), "s2": f(64) + 1.0, "t2": f(64),
              "w3": f(4, 16), "b3": f(4)}
    if (frame, naux, main_only, coord) in ((224, 7, False, False), (64, 6, False, False), (16, 2, True, False)):
        assert g.fused_classifier_ok                      # benchmark topologies (hierarchical and main-grid-only) must qualify
    if not g.fused_classifier_ok:
        with pytest.raises(RuntimeError):
            ops.gcn_layer_cls_fwd(g, B, x, w, sc, sh, x, False, packed, sigmoid)
        return
    h = ops.gcn_layer_fwd(g, B, x, w, sc, sh, x, relu=False)
    want = ops.classifier_fwd(h, B, g.num_nodes, 0, g.num_nodes, packed, sigmoid=sigmoid)
    got = ops.gcn_layer_cls_fwd(g, B, x, w, sc, sh, x, False, packed, sigmoid)
    assert float((got - want).abs().max()) < 2e-5 * max(1.0, float(want.abs().max()))
    if g.kidsum_rows == 0:
        return
    # chained form: child sums of x from a previous layer
    ka = ops.new_kidsum(g, B)
    h0 = ops.gcn_layer_fwd(g, B, x, w, sc, sh, x, relu=True, kidsum_out=ka)
    want2 = ops.classifier_fwd(ops.gcn_layer_fwd(g, B, h0, w, sc, sh, h0, relu=False), B, g.num_nodes, 0, g.num_nodes, packed,
                               sigmoid=sigmoid)
    got2 = ops.gcn_layer_cls_fwd(g, B, h0, w, sc, sh, h0, False, packed, sigmoid, kidsum_in=ka)
    assert float((got2 - want2).abs().max()) < 2e-5 * max(1.0, float(want2.abs().max()))
    assert torch.equal(got2, ops.gcn_layer_cls_fwd(g, B, h0, w, sc, sh, h0, False, packed, sigmoid, kidsum_in=ka))   # deterministic


def test_fused_layer_is_run_to_run_deterministic():
    topo, ei, _, _ = graph_tensors(64, 6, 2)
    x = rand_rows(2 * topo.num_nodes, seed=1).to(DEV)
    w = rand_rows(128, seed=2).to(DEV) * 0.1
    g = ops.Graph.topo(64, 6)
    a = ops.gcn_layer_fwd(g, 2, x, w, None, None, x, relu=True)
    for _ in range(3):
        assert torch.equal(a, ops.gcn_layer_fwd(g, 2, x, w, None, None, x, relu=True))


@pytest.mark.parametrize("rows_per_frame,row_lo,n_valid,batch", [(340, 0, 340, 2), (344, 0, 340, 3), (348, 4, 340, 2),
                                                                  (100, 3, 1, 1), (4116, 0, 4116, 1)])
@pytest.mark.parametrize("sigmoid", [False, True])
def test_classifier_heads(rows_per_frame, row_lo, n_valid, batch, sigmoid):
    from gpu_util import model_pair
    hip, ref = model_pair(16, 3, 1, seed=21, output_activation="sigmoid" if sigmoid else "logit")
    h = rand_rows(batch * rows_per_frame, seed=3)
    got = ops.classifier_fwd(h.to(DEV), batch, rows_per_frame, row_lo, n_valid, hip._packed_classifier(),
                             sigmoid=sigmoid).cpu()
    hv = h.view(batch, rows_per_frame, 128)[:, row_lo:row_lo + n_valid].reshape(-1, 128)
    with torch.no_grad():
        want = torch.cat([c(hv) for c in ref.node_classifiers], dim=1)
    assert got.shape == want.shape
    assert (got - want).abs().max() < 2e-5


@pytest.mark.parametrize("kind", ["hierarchy", "multigraph", "directed", "tiny", "isolated"])
def test_csr_layer_in_clustered_tiles_equals_the_row_by_row_aggregator(kind, monkeypatch):
    """CSR handles regroup their rows into breadth-first tiles of 64 nodes, keep a tile's raw rows in LDS and walk a wave's 8 rows
    as one lane-parallel edge list (graph.hip csr_tiles, k_gcn_layer<AGG_CSRT>).  A row's sum has a fixed order (self, sources
    inside the tile, sources outside, each group in edge_index order): launches are bitwise reproducible, and the output agrees
    with the row-by-row aggregator (EG_CSR_TILES=0) and with the same walk over consecutive rows (EG_CSR_TILES=1) to rounding --
    forward, transposed-weight form and with a residual -- on the reference's hierarchy given as a plain edge_index, a random
    multigraph with duplicate edges and self loops, a directed graph (the backward's transposed handle too), graphs smaller than
    a tile, isolated nodes."""
    rs = np.random.RandomState(11)
    if kind == "hierarchy":
        topo = HierTopology(TopologySpec(64, 6, False, False))
        ei = torch.from_numpy(topo.batched_edge_index(3))
        n = 3 * topo.num_nodes
    elif kind == "multigraph":
        n = 5000
        src, dst = rs.randint(0, n, 30000), rs.randint(0, n, 30000)
        e = np.stack([np.concatenate([src, dst, src[:500], np.arange(0, n, 7)]), np.concatenate([dst, src, dst[:500], np.arange(0, n, 7)])])
        ei = torch.from_numpy(e.astype(np.int64))
    elif kind == "directed":
        n = 3000
        ei = torch.from_numpy(np.stack([rs.randint(0, n, 20000), rs.randint(0, n, 20000)]).astype(np.int64))
    elif kind == "tiny":
        n = 37
        ei = torch.from_numpy(np.stack([rs.randint(0, n, 90), rs.randint(0, n, 90)]).astype(np.int64))
    else:
        n = 1000                                                     # only the first 100 nodes have edges
        ei = torch.from_numpy(np.stack([rs.randint(0, 100, 600), rs.randint(0, 100, 600)]).astype(np.int64))
    x = rand_rows(n, seed=3).to(DEV)
    res = rand_rows(n, seed=4).to(DEV)
    w = torch.from_numpy(rs.uniform(-0.1, 0.1, (128, 128)).astype(np.float32)).to(DEV)
    sc, sh = torch.from_numpy(rs.uniform(0.5, 1.5, 128).astype(np.float32)).to(DEV), torch.from_numpy(rs.uniform(-0.2, 0.2, 128).astype(np.float32)).to(DEV)
    outs = {}
    for mode in ("0", "1", "2"):
        monkeypatch.setenv("EG_CSR_TILES", mode)                     # (read when the handle is created)
        g = ops.Graph.csr(ei.to(DEV), n)
        outs[mode] = (ops.gcn_layer_fwd(g, 1, x, w, sc, sh, x, relu=True), ops.gcn_layer_fwd(g, 1, x, w, None, None, res, relu=False),
                      ops.gcn_layer_fwd(g.bwd, 1, x, w, None, None, None, relu=False, transpose_w=True))
        torch.cuda.synchronize()
        again = ops.gcn_layer_fwd(g, 1, x, w, sc, sh, x, relu=True)
        assert torch.equal(again, outs[mode][0]), (kind, mode)          # fixed summation order: bitwise run to run
    for mode in ("1", "2"):
        for a, b in zip(outs["0"], outs[mode]):
            assert (a - b).abs().max() <= 2e-5 * max(1.0, float(a.abs().max())), (kind, mode, float((a - b).abs().max()))
    # and against the sparse oracle
    want = O.gcn_conv_sparse(x.cpu(), ei, w.cpu(), torch.zeros(128))
    got = ops.gcn_layer_fwd(ops.Graph.csr(ei.to(DEV), n), 1, x, w, None, None, None, relu=False)
    assert (got.cpu() - want).abs().max() < 2e-4 * max(1.0, float(want.abs().max()))
