"""-m gpu: the torch_geometric-shaped operator surface — the call shapes the reference's unmodified model code uses
(src/core/models.py:5, :329-335 construction, :431 ``layer(x, edge_index)``, :434-435 residual, :479-482 JumpingKnowledge)
— forward AND backward through the HIP kernels against the CPU oracle."""
import os

import numpy as np
import pytest
import torch
import torch.nn as nn

from fixtures_util import fill_state_dict, synthetic_frames, synthetic_node_feats
from gpu_util import DEV, graph_tensors, model_pair
from oracle import gnn_oracle as O
from echoglad_amd import nn as egnn
from echoglad_amd import ops

pytestmark = pytest.mark.gpu


def _stack(conv_cls, seq_cls, jk_cls, layers, jk):
    """The reference's constructor loop (models.py:328-335, :380-382) with the given torch_geometric.nn classes."""
    m = nn.Module()
    m.gnn_layers = nn.ModuleList()
    for i in range(layers):
        m.gnn_layers.append(seq_cls("x, edge_index", [
            (conv_cls(in_channels=128, out_channels=128), "x, edge_index -> x"),
            nn.BatchNorm1d(128), nn.Dropout(p=0.0), nn.Identity() if i == layers - 1 else nn.ReLU(inplace=True)]))
    m.jk = jk_cls(jk) if jk != "last" else None
    return m


def _loop(m, x, edge_index):
    """models.py:426-435, :476-482."""
    hidden = [x]
    for i in range(len(m.gnn_layers)):
        h = m.gnn_layers[i](hidden[i], edge_index)
        h = h + hidden[i]
        hidden.append(h)
    return m.jk(hidden) if m.jk is not None else hidden[-1]


def _random_multigraph(n, e, seed):
    rs = np.random.RandomState(seed)
    src, dst = rs.randint(0, n, e), rs.randint(0, n, e)
    ei = np.stack([np.concatenate([src, dst, src[:50]]), np.concatenate([dst, src, dst[:50]])])    # + 50 duplicate edges
    ei[:, 3] = [7, 7]                                                                              # + an explicit self loop
    return torch.from_numpy(ei.astype(np.int64))


@pytest.mark.parametrize("graph_kind", ["closed_form", "diagonal", "multigraph"])
@pytest.mark.parametrize("jk", ["last", "max"])
@pytest.mark.parametrize("train", [False, True])
def test_reference_shaped_loop_forward_backward(graph_kind, jk, train):
    L = 3
    hip = _stack(egnn.GCNConv, egnn.Sequential, egnn.JumpingKnowledge, L, jk)
    ref = _stack(O.OracleGCNConv, O.OracleSequential, O.OracleJumpingKnowledge, L, jk)
    fill_state_dict(ref, seed=17)
    hip.load_state_dict(ref.state_dict(), strict=True)
    # state_dict keys are the reference's: gnn_layers.{i}.module_0.lin.weight / .bias, module_1.*
    assert "gnn_layers.0.module_0.lin.weight" in hip.state_dict() and "gnn_layers.2.module_1.running_var" in hip.state_dict()
    hip = hip.to(DEV)
    hip.train(train); ref.train(train)
    if graph_kind == "multigraph":
        n = 777
        ei = _random_multigraph(n, 3000, 5)
    else:
        diag = "grid-diagonal" if graph_kind == "diagonal" else "grid"
        topo, ei, nt, bi = graph_tensors(16, 3, 2, main_type=diag, aux_type=diag)
        n = 2 * topo.num_nodes
    x = synthetic_node_feats(n, 128, seed=3)
    xr = x.clone().requires_grad_(True)
    xh = x.clone().to(DEV).requires_grad_(True)
    eih = ei.to(DEV)
    want = _loop(ref, xr, ei)
    got = _loop(hip, xh, eih)
    structured = egnn._SHARED_RESOLVER.resolve(eih, n)[0].structured
    assert structured == (graph_kind in ("closed_form", "diagonal"))      # a stand-alone GCNConv recognises the reference's topologies, 'grid-diagonal' included
    assert (got.detach().cpu() - want.detach()).abs().max() < 2e-4
    (want ** 2).mean().backward()
    (got ** 2).mean().backward()
    gx = xr.grad
    assert (xh.grad.cpu() - gx).abs().max() < 5e-3 * gx.abs().max() + 1e-7
    ref_grads = dict(ref.named_parameters())
    for name, p in hip.named_parameters():
        rg = ref_grads[name].grad
        assert p.grad is not None, name
        err = (p.grad.cpu() - rg).abs().max().item()
        assert err < 5e-3 * rg.abs().max().item() + 1e-6, (name, err, rg.abs().max().item())


def test_gcnconv_alone_matches_dense_fp64():
    topo, ei, nt, bi = graph_tensors(30, 3, 1)
    conv = egnn.GCNConv(128, 128)
    fill_state_dict(conv, seed=2)
    x = synthetic_node_feats(topo.num_nodes, 128, seed=8)
    want = O.gcn_conv_dense64(x, ei, conv.lin.weight.detach(), conv.bias.detach())
    conv = conv.to(DEV)
    with torch.no_grad():
        got = conv(x.to(DEV), ei.to(DEV))
    assert (got.cpu().double() - want).abs().max() < 2e-5
    with pytest.raises(NotImplementedError):
        egnn.GCNConv(128, 256)                              # (wider than the kernels' 128-channel rows)
    with pytest.raises(NotImplementedError):
        egnn.GCNConv(128, 128, improved=True)
    # narrower layers -- the reference's signature defaults are 128 -> 64 -> 64 -- run zero-padded on the same kernels
    for cin, cout in ((128, 64), (64, 64), (40, 100)):
        narrow = egnn.GCNConv(cin, cout)
        fill_state_dict(narrow, seed=cin + cout)
        xn = synthetic_node_feats(topo.num_nodes, 128, seed=9)[:, :cin].contiguous()
        want = O.gcn_conv_dense64(xn, ei, narrow.lin.weight.detach(), narrow.bias.detach())
        narrow = narrow.to(DEV)
        xg = xn.to(DEV).requires_grad_(True)
        got = narrow(xg, ei.to(DEV))
        assert got.shape == (topo.num_nodes, cout) and (got.detach().cpu().double() - want).abs().max() < 2e-5
        got.sum().backward()
        assert xg.grad.shape == xn.shape and narrow.lin.weight.grad.shape == (cout, cin) and narrow.bias.grad.shape == (cout,)


@pytest.mark.parametrize("frame,naux,batch", [(16, 3, 2), (64, 6, 2)])
def test_model_jumping_knowledge_max(frame, naux, batch):
    """gnn_jk_mode='max' (models.py:380-382, :479-482) through the whole model."""
    hip, ref = model_pair(frame, naux, 3, seed=21, gnn_jk_mode="max")
    topo, ei, nt, bi = graph_tensors(frame, naux, batch)
    frames = synthetic_frames(batch, 128, frame, 9)
    with torch.no_grad():
        want, _ = ref(x=frames, edge_index=ei, node_type=nt, batch_idx=bi)
        got, _ = hip(x=frames.to(DEV), edge_index=ei.to(DEV), node_type=nt.to(DEV), batch_idx=bi.to(DEV))
    assert (got.cpu() - want).abs().max() < 1e-4
    assert torch.equal(O.landmark_argmax(got.cpu(), batch, frame), O.landmark_argmax(want, batch, frame))
    # the maximum is carried through the FUSED stack (eg_gcn_layer_fwd_jk / eg_gcn_layer_cls_fwd with jk_in): the chained
    # producer/consumer kernel ran once per layer, and the unfused route (layers one by one + torch.stack().max()) agrees
    feats = hip.create_node_pixels(frames.to(DEV), batch)
    graph, gb = hip._resolver.resolve(ei.to(DEV), feats.shape[0])
    assert graph.fused_classifier_ok
    before = graph.ps_launches
    with torch.no_grad():
        fused, _ = hip.forward_nodes(feats, ei.to(DEV), batch)
    assert graph.ps_launches - before == 3
    from echoglad_amd import nn as egnn
    egnn.ROUTES.jk_fused = False
    try:
        before = graph.layer_launches
        with torch.no_grad():
            unfused, _ = hip.forward_nodes(feats, ei.to(DEV), batch)
        assert graph.layer_launches - before == 3      # (three plain layer launches; torch takes the maximum and the heads run separately)
    finally:
        egnn.ROUTES.jk_fused = True
    assert (fused - unfused).abs().max() < 2e-5
    with pytest.raises(NotImplementedError):
        model_pair(frame, naux, 3, gnn_jk_mode="cat")


def test_jumping_knowledge_max_stays_within_1p3x_of_the_last_path_at_full_size():
    """configs[1] shape (224 / 7, batch 8): gnn_jk_mode='max' on the fused path against gnn_jk_mode='last', HIP-graph replays."""
    frame, naux, B = 224, 7, 8
    topo, ei, nt, bi = graph_tensors(frame, naux, B)
    x = synthetic_node_feats(B * topo.num_nodes, 128, seed=3).to(DEV)
    eid = ei.to(DEV)
    times = {}
    for mode in ("last", "max"):
        hip, _ = model_pair(frame, naux, 3, seed=5, gnn_jk_mode=mode)
        hip.enable_hip_graph(True)
        with torch.no_grad():
            for _ in range(5):
                hip.forward_nodes(x, eid, B)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(30):
                hip.forward_nodes(x, eid, B)
            e1.record()
            torch.cuda.synchronize()
        times[mode] = e0.elapsed_time(e1) / 30
    assert times["max"] <= 1.3 * times["last"], times


@pytest.mark.parametrize("coord", [False, True])
def test_model_with_connection_nodes(coord):
    """use_connection_nodes=True (datasets.py:1450-1456, :1512-1515; models.py:520-524 connection-node features,
    :485 node-type filter dropping the LEADING naux+1 rows of every frame): eval logits, then a train step (p = 0)."""
    frame, naux, B, L = 16, 3, 2, 3
    hip, ref = model_pair(frame, naux, L, coord=coord, seed=13, use_connection_nodes=True)
    topo, ei, nt, bi = graph_tensors(frame, naux, B, coord=coord, conn=True)
    assert topo.n_conn == naux + 1
    frames = synthetic_frames(B, 128, frame, 4)
    from fixtures_util import initial_coords
    c0 = initial_coords(B, frame) if coord else None

    def run(m, dev):
        return m(x=frames.to(dev), node_coords=None if c0 is None else c0.clone().to(dev), edge_index=ei.to(dev),
                 node_type=nt.to(dev), batch_idx=bi.to(dev))

    with torch.no_grad():
        want, wc = run(ref, "cpu")
        got, gc = run(hip, DEV)
    assert got.shape == want.shape == (B * topo.num_valid_nodes, 4)
    assert (got.cpu() - want).abs().max() < 1e-4
    if coord:
        assert (gc.cpu() - wc).abs().max() < 1e-4
    for m in list(hip.modules()) + list(ref.modules()):
        if isinstance(m, nn.Dropout):
            m.p = 0.0
    hip.train(); ref.train()
    want, wc = run(ref, "cpu")
    got, gc = run(hip, DEV)
    assert (got.detach().cpu() - want.detach()).abs().max() < 2e-4
    (want ** 2).mean().backward(); (got ** 2).mean().backward()
    ref_grads = dict(ref.named_parameters())
    for name, p in hip.named_parameters():
        rg = ref_grads[name].grad
        if rg is None:
            continue
        err = (p.grad.cpu() - rg).abs().max().item()
        assert err < 5e-3 * rg.abs().max().item() + 1e-6, (name, err, rg.abs().max().item())


@pytest.mark.parametrize("coord", [False, True])
def test_eval_mode_with_gradients(coord):
    """model.eval() with autograd on (e.g. saliency / fine-tuning with frozen statistics): the un-fused branch gives the
    fused branch's logits and the oracle's gradients."""
    frame, naux, B = 32, 4, 2
    hip, ref = model_pair(frame, naux, 3, coord=coord, seed=3)
    topo, ei, nt, bi = graph_tensors(frame, naux, B, coord=coord)
    feats = synthetic_node_feats(B * topo.num_nodes, 128, seed=6)
    from fixtures_util import initial_coords
    c0 = initial_coords(B, frame) if coord else None
    with torch.no_grad():
        fused, _ = hip.forward_nodes(feats.to(DEV), ei.to(DEV), B, None if c0 is None else c0.clone().to(DEV))
    xh = feats.clone().to(DEV).requires_grad_(True)
    xr = feats.clone().requires_grad_(True)
    got, _ = hip.forward_nodes(xh, ei.to(DEV), B, None if c0 is None else c0.clone().to(DEV))
    want, _ = ref.forward_nodes(xr, ei, nt, B, None if c0 is None else c0.clone())
    assert (got.detach() - fused).abs().max() < 2e-5
    assert (got.detach().cpu() - want.detach()).abs().max() < 1e-4
    (got ** 2).mean().backward(); (want ** 2).mean().backward()
    assert (xh.grad.cpu() - xr.grad).abs().max() < 5e-3 * xr.grad.abs().max()
    ref_grads = dict(ref.named_parameters())
    for name, p in hip.named_parameters():
        rg = ref_grads[name].grad
        err = (p.grad.cpu() - rg).abs().max().item()
        assert err < 5e-3 * rg.abs().max().item() + 1e-6, (name, err)


def test_resolver_never_serves_a_stale_graph():
    """A loader that frees and re-allocates edge_index every batch gets the same address back from the caching
    allocator; a different graph of the same shape must not hit the old handle (nn.py:65)."""
    conv = egnn.GCNConv(128, 128)
    fill_state_dict(conv, seed=1)
    w, b = conv.lin.weight.detach().clone(), conv.bias.detach().clone()
    conv = conv.to(DEV)
    n = 500
    x = synthetic_node_feats(n, 128, seed=2)
    ptrs = []
    for seed in (1, 2, 3, 1):
        ei = _random_multigraph(n, 2000, seed)
        eid = ei.to(DEV)
        ptrs.append(eid.data_ptr())
        with torch.no_grad():
            got = conv(x.to(DEV), eid).cpu()
        want = O.gcn_conv_sparse(x, ei, w, b)
        assert (got - want).abs().max() < 1e-4, seed
        del eid
    assert len(set(ptrs)) < len(ptrs), "the test did not exercise address reuse"
    # in-place modification of a live tensor is seen too (_version)
    ei = _random_multigraph(n, 2000, 9).to(DEV)
    with torch.no_grad():
        a = conv(x.to(DEV), ei)
        ei[:, :100] = 0
        bb = conv(x.to(DEV), ei)
    want = O.gcn_conv_sparse(x, ei.cpu(), w, b)
    assert (bb.cpu() - want).abs().max() < 1e-4 and not torch.equal(a, bb)


def test_shared_handle_on_two_streams():
    """One graph handle, launches on two streams at once (include/echoglad_hip.h: re-entrant): every launch has its own
    slice of the handle's tile-queue ring, so the results are the serial results, bit for bit."""
    B = 4
    g = ops.Graph.topo(64, 6)
    n = g.num_nodes
    rs = np.random.RandomState(0)
    w = torch.from_numpy(rs.uniform(-0.1, 0.1, (128, 128)).astype(np.float32)).to(DEV)
    xs = [synthetic_node_feats(B * n, 128, seed=s).to(DEV) for s in (1, 2)]
    serial = [ops.gcn_layer_fwd(g, B, x, w, residual=x, relu=True) for x in xs]
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    for _ in range(20):
        outs = []
        for st, x in ((s1, xs[0]), (s2, xs[1])):
            with torch.cuda.stream(st):
                outs.append(ops.gcn_layer_fwd(g, B, x, w, residual=x, relu=True))
        torch.cuda.synchronize()
        assert torch.equal(outs[0], serial[0]) and torch.equal(outs[1], serial[1])


def test_queue_ring_refuses_a_slice_still_in_flight_on_another_stream():
    """include/echoglad_hip.h: a launch takes the next of 64 tile-queue slices of its handle; the slice of the launch 64 calls
    ago must be free.  If that launch is still in flight on ANOTHER stream the call is refused (unsupported) instead of sharing
    live counters -- and it works again once that stream has drained.  Same-stream reuse is ordered and always fine."""
    B = 2
    g = ops.Graph.topo(32, 4)
    x = synthetic_node_feats(B * g.num_nodes, 128, seed=1).to(DEV)
    w = torch.eye(128, device=DEV)
    want = ops.gcn_layer_fwd(g, B, x, w, residual=x, relu=True)
    torch.cuda.synchronize()
    for _ in range(200):                                             # one stream: the ring wraps three times, no refusal
        ops.gcn_layer_fwd(g, B, x, w, residual=x, relu=True)
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    with torch.cuda.stream(s1):
        torch.cuda._sleep(int(4e9))                                  # ~2 s of GPU time in front of the launch that takes a slice
        held = ops.gcn_layer_fwd(g, B, x, w, residual=x, relu=True)
    refused = None
    with torch.cuda.stream(s2):
        for k in range(70):
            try:
                ops.gcn_layer_fwd(g, B, x, w, residual=x, relu=True)
            except RuntimeError as ex:
                refused = (k, str(ex))
                break
    assert refused is not None and refused[0] == 63 and "unsupported" in refused[1] and "in flight" in refused[1], refused
    torch.cuda.synchronize()
    assert torch.equal(held, want)
    with torch.cuda.stream(s2):
        got = ops.gcn_layer_fwd(g, B, x, w, residual=x, relu=True)   # the held slice is free again
    torch.cuda.synchronize()
    assert torch.equal(got, want)


def test_queue_slices_are_left_zeroed_by_the_kernels_themselves(monkeypatch):
    """Both layer kernels zero their slice of the tile-queue ring on the way out (no memset in front of a launch, no host flag):
    the ring wraps many times with identical results, also at the full persistent grid, with launches of the symmetric kernel
    (plain eg_gcn_layer_fwd on a hierarchical handle: queue walk) taking slices in between."""
    monkeypatch.setenv("EG_LAYER_IMPL", "0")            # plain calls on the symmetric kernel (read when the handle is created)
    for frame, naux, B in ((64, 6, 2), (224, 7, 4)):
        g = ops.Graph.topo(frame, naux)
        x = synthetic_node_feats(B * g.num_nodes, 128, seed=1).to(DEV)
        rs = np.random.RandomState(3)
        w = torch.from_numpy(rs.uniform(-0.1, 0.1, (128, 128)).astype(np.float32)).to(DEV)
        ka, kb = ops.new_kidsum(g, B), ops.new_kidsum(g, B)                  # chained calls run on the producer / consumer kernel
        one, zero = torch.ones(128, device=DEV), torch.zeros(128, device=DEV)
        h = ops.gcn_layer_fwd(g, B, x, w, one, zero, x, relu=True, kidsum_out=ka)
        want = ops.gcn_layer_fwd(g, B, h, w, one, zero, h, relu=True, kidsum_in=ka, kidsum_out=kb)
        want_sym = ops.gcn_layer_fwd(g, B, x, w, one, zero, x, relu=True)   # symmetric kernel
        before, before_all = g.ps_launches, g.layer_launches
        n_sym = 0
        for k in range(300):
            got = ops.gcn_layer_fwd(g, B, h, w, one, zero, h, relu=True, kidsum_in=ka, kidsum_out=kb)
            if k % 37 == 0:
                assert torch.equal(got, want), k
            if k % 7 == 6:                                           # (7 and 64 are coprime: the symmetric kernel visits every slice)
                got_sym = ops.gcn_layer_fwd(g, B, x, w, one, zero, x, relu=True)
                n_sym += 1
                if k % 49 == 48:
                    assert torch.equal(got_sym, want_sym), k
        assert torch.equal(got, want) and g.ps_launches - before == 300 and g.layer_launches - before_all == 300 + n_sym


@pytest.mark.parametrize("captured", ["producer_consumer", "symmetric"])
def test_hip_graph_replay_is_immune_to_eager_launches_on_the_same_handle(captured, monkeypatch):
    """a launch recorded into a HIP graph keeps its queue slice for every replay, and whether that slice is clean
    must not depend on host state frozen at capture time.  Capture one launch, let 70 eager launches of the OTHER kernel wrap
    the 64-slice ring (they use the captured slice too), replay on a changed input: the replay equals an eager launch."""
    B = 2
    monkeypatch.setenv("EG_LAYER_IMPL", "0")            # plain calls on the symmetric kernel, chained ones on the producer / consumer kernel
    g = ops.Graph.topo(64, 6)
    rs = np.random.RandomState(5)
    w = torch.from_numpy(rs.uniform(-0.1, 0.1, (128, 128)).astype(np.float32)).to(DEV)
    x = synthetic_node_feats(B * g.num_nodes, 128, seed=1).to(DEV)
    ka = ops.new_kidsum(g, B)
    kw_cap = dict(kidsum_out=ka) if captured == "producer_consumer" else {}
    kw_other = {} if captured == "producer_consumer" else dict(kidsum_out=ops.new_kidsum(g, B))
    out = torch.empty_like(x)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        ops.gcn_layer_fwd(g, B, x, w, residual=x, relu=True, out=out, **kw_cap)
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        ops.gcn_layer_fwd(g, B, x, w, residual=x, relu=True, out=out, **kw_cap)
    for rnd in range(3):
        for _ in range(70):
            ops.gcn_layer_fwd(g, B, x, w, residual=x, relu=True, **kw_other)
        x.copy_(synthetic_node_feats(B * g.num_nodes, 128, seed=10 + rnd).to(DEV))
        graph.replay()
        torch.cuda.synchronize()
        want = ops.gcn_layer_fwd(g, B, x, w, residual=x, relu=True, **kw_cap)
        assert torch.equal(out, want), rnd


def test_connection_node_scratch_growth_keeps_captured_graphs_valid():
    """a HIP graph captured at a small batch holds its slice of the connection-node scratch in its kernel
    nodes; a later launch with more frames than the scratch holds allocates a larger one and must NOT free the old one."""
    g = ops.Graph.topo(32, 4, use_connection_nodes=True)
    rs = np.random.RandomState(7)
    w = torch.from_numpy(rs.uniform(-0.1, 0.1, (128, 128)).astype(np.float32)).to(DEV)
    B = 4
    x = synthetic_node_feats(B * g.num_nodes, 128, seed=1).to(DEV)
    out = torch.empty_like(x)
    before = g.ps_launches
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        ops.gcn_layer_fwd(g, B, x, w, residual=x, relu=True, out=out)
    torch.cuda.current_stream().wait_stream(side)
    assert g.ps_launches == before + 1                                # (the connection-node stencil lives in the producer/consumer kernel)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        ops.gcn_layer_fwd(g, B, x, w, residual=x, relu=True, out=out)
    for big in (16, 40):                                              # default scratch: 8 frames -> grows twice
        xb = synthetic_node_feats(big * g.num_nodes, 128, seed=big).to(DEV)
        yb = ops.gcn_layer_fwd(g, big, xb, w, residual=xb, relu=True)
        # frames are independent: frame 3 of the big batch alone gives the same rows
        n = g.num_nodes
        alone = ops.gcn_layer_fwd(g, 1, xb[3 * n:4 * n].contiguous(), w, residual=xb[3 * n:4 * n].contiguous(), relu=True)
        assert torch.equal(yb[3 * n:4 * n], alone)
        # fill fresh allocations (what a freed scratch would have become) and replay the old graph on new input
        junk = [torch.full((1 << 20,), float("nan"), device=DEV) for _ in range(8)]
        x.copy_(synthetic_node_feats(B * g.num_nodes, 128, seed=100 + big).to(DEV))
        graph.replay()
        torch.cuda.synchronize()
        want = ops.gcn_layer_fwd(g, B, x, w, residual=x, relu=True)
        assert torch.equal(out, want), big
        del junk


@pytest.mark.parametrize("train", [False, True])
def test_reference_loop_at_full_frame_size_is_one_launch_per_layer(train):
    """INTEGRATION route B at the default frame: the reference's own constructor loop and forward loop (models.py:328-335,
    :426-435) over this package's torch_geometric-shaped classes, 224x224 / 7 aux levels, 2 frames.  Sequential recognises the
    reference's layer and runs it as ONE kernel launch per layer (eg_graph_layer_launches) -- eval: bias + BatchNorm + ReLU folded
    into eg_gcn_layer_fwd; train (p = 0): the eg_gcn_layer_train_fwd / eg_gcn_layer_bwd composites -- with the module-by-module
    route's values (EG_SEQ_FUSED=0) and the oracle's."""
    L, B = 3, 2
    hip = _stack(egnn.GCNConv, egnn.Sequential, egnn.JumpingKnowledge, L, "last")
    ref = _stack(O.OracleGCNConv, O.OracleSequential, O.OracleJumpingKnowledge, L, "last")
    fill_state_dict(ref, seed=23)
    hip.load_state_dict(ref.state_dict(), strict=True)
    hip = hip.to(DEV)
    hip.train(train); ref.train(train)
    topo, ei, nt, bi = graph_tensors(224, 7, B)
    n = B * topo.num_nodes
    x = synthetic_node_feats(n, 128, seed=4)
    eih = ei.to(DEV)
    graph, gb = egnn._SHARED_RESOLVER.resolve(eih, n)
    assert graph.structured and gb == B
    state = {k: v.clone() for k, v in hip.state_dict().items()}
    xh = x.clone().to(DEV).requires_grad_(train)
    before = graph.layer_launches
    with torch.set_grad_enabled(train):
        got = _loop(hip, xh, eih)
    assert graph.layer_launches - before == L                         # one fused launch per layer, nothing else on the handle
    if train:
        (got ** 2).mean().backward()
        fused_grads = {k: p.grad.clone() for k, p in hip.named_parameters()}
        fused_xgrad = xh.grad.clone()
    # the module-by-module route (GCNConv kernel, then torch BatchNorm1d / Dropout / ReLU) from the same state
    hip.load_state_dict(state)
    hip.zero_grad(set_to_none=True)
    os.environ["EG_SEQ_FUSED"] = "0"
    try:
        xm = x.clone().to(DEV).requires_grad_(train)
        with torch.set_grad_enabled(train):
            unfused = _loop(hip, xm, eih)
        if train:
            (unfused ** 2).mean().backward()
    finally:
        del os.environ["EG_SEQ_FUSED"]
    assert (got.detach() - unfused.detach()).abs().max() < 5e-5
    # ... and the oracle
    xr = x.clone().requires_grad_(train)
    with torch.set_grad_enabled(train):
        want = _loop(ref, xr, ei)
    assert (got.detach().cpu() - want.detach()).abs().max() < 2e-4
    if train:
        (want ** 2).mean().backward()

        # 18 M activations: a handful of pre-activations sit within rounding of the ReLU kink and two fp32 evaluations put them on
        # different sides, which moves single gradient entries by a whole term -- so: the bulk tightly (Frobenius),
        # no entry grossly, and per channel of a parameter as gpu_util.assert_param_grads_close does
        def close(a, b, what):
            a, b = a.detach().cpu().double(), b.detach().cpu().double()
            assert float((a - b).norm()) <= 2e-3 * float(b.norm()) + 1e-12, (what, float((a - b).norm()), float(b.norm()))
            assert float((a - b).abs().max()) <= 5e-2 * float(b.abs().max()) + 1e-12, what

        close(fused_xgrad, xr.grad, "dx vs oracle")
        close(fused_xgrad, xm.grad, "dx vs module-by-module")
        ref_grads = dict(ref.named_parameters())
        gmax = max(float(q.grad.abs().max()) for q in ref_grads.values())
        for name, p in hip.named_parameters():
            rg = ref_grads[name].grad
            if float(rg.abs().max()) < 1e-5 * gmax:          # the conv bias in front of a train-mode BatchNorm: analytically zero
                assert float(fused_grads[name].abs().max()) <= 1e-4 * gmax, name
                continue
            close(fused_grads[name], rg, name)
        # running statistics moved exactly as nn.BatchNorm1d moves them
        for k, v in ref.state_dict().items():
            if "running" in k or "num_batches" in k:
                assert (hip.state_dict()[k].cpu().double() - v.double()).abs().max() < 1e-4 * max(1.0, float(v.abs().max())), k


def test_sequential_keeps_the_module_route_where_it_must():
    """Forward hooks on a child, a frozen BatchNorm inside a training layer, eval with gradients, and module lists other than
    the reference's: Sequential runs module by module (and gives the same values)."""
    topo, ei, nt, bi = graph_tensors(16, 3, 2)
    n = 2 * topo.num_nodes
    x = synthetic_node_feats(n, 128, seed=2).to(DEV)
    eih = ei.to(DEV)
    seq = egnn.Sequential("x, edge_index", [(egnn.GCNConv(128, 128), "x, edge_index -> x"), nn.BatchNorm1d(128), nn.Dropout(p=0.0),
                                             nn.ReLU(inplace=True)])
    fill_state_dict(seq, seed=3)
    seq = seq.to(DEV).eval()
    with torch.no_grad():
        fused = seq(x, eih)
    seen = []
    hook = seq.module_1.register_forward_hook(lambda m, i, o: seen.append(o.detach().clone()))
    with torch.no_grad():
        hooked = seq(x, eih)
    hook.remove()
    assert len(seen) == 1 and (fused - hooked).abs().max() < 2e-5 and (torch.relu(seen[0]) - hooked).abs().max() == 0
    xg = x.clone().requires_grad_(True)
    with_grad = seq(xg, eih)                                          # eval + autograd: module by module, differentiable
    with_grad.sum().backward()
    assert xg.grad is not None and (with_grad.detach() - fused).abs().max() < 2e-5
    other = egnn.Sequential("x, edge_index", [(egnn.GCNConv(128, 128), "x, edge_index -> x"), nn.ReLU()])
    assert other._reference_layer() is None
