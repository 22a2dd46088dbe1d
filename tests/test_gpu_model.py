"""-m gpu: the whole GNN stack through the reference-shaped model API vs golden fixtures and the oracle.
Bar (BASELINE.json north_star): logits within 1e-4 fp32, landmark argmax indices bit-exact."""
import os

import numpy as np
import pytest
import torch

from fixtures_util import initial_coords, synthetic_frames, synthetic_node_feats
from gpu_util import DEV, graph_tensors, model_pair
from oracle import gnn_oracle as O

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _run_both(hip, ref, frames, ei, nt, bi, coords=None):
    with torch.no_grad():
        want, wc = ref(x=frames, node_coords=None if coords is None else coords.clone(), edge_index=ei,
                       node_type=nt, batch_idx=bi)
        got, gc = hip(x=frames.to(DEV), node_coords=None if coords is None else coords.clone().to(DEV),
                      edge_index=ei.to(DEV), node_type=nt.to(DEV), batch_idx=bi.to(DEV))
    return got.cpu(), want, (None if gc is None else gc.cpu()), wc


def test_kat_fixture(golden_dir):
    g = np.load(os.path.join(golden_dir, "kat_f16_a3.npz"))
    hip, ref = model_pair(16, 3, 3, seed=int(g["weight_seed"]))
    topo, ei, nt, bi = graph_tensors(16, 3, int(g["batch"]))
    frames = synthetic_frames(int(g["batch"]), 128, 16, int(g["frame_seed"]))
    got, want, _, _ = _run_both(hip, ref, frames, ei, nt, bi)
    assert np.abs(got.numpy() - g["logits"]).max() < TOL
    assert (got - want).abs().max() < TOL
    # the reference's own (from_networkx) edge order must resolve to the same implicit topology
    with torch.no_grad():
        got2, _ = hip(x=frames.to(DEV), edge_index=torch.from_numpy(g["edge_index"]).to(DEV), node_type=nt.to(DEV),
                      batch_idx=bi.to(DEV))
    assert torch.equal(got2.cpu(), got)
    assert hip._resolver.resolve(torch.from_numpy(g["edge_index"]).to(DEV), nt.numel())[0].structured


def test_cfg1_fixture(golden_dir):
    g = np.load(os.path.join(golden_dir, "cfg1_f64_a2.npz"))
    hip, ref = model_pair(64, 2, 2, seed=int(g["weight_seed"]))
    topo, ei, nt, bi = graph_tensors(64, 2, 1)
    frames = synthetic_frames(1, 128, 64, int(g["frame_seed"]))
    got, want, _, _ = _run_both(hip, ref, frames, ei, nt, bi)
    assert np.abs(got.numpy()[g["sample_rows"]] - g["logits_rows"]).max() < TOL
    assert (got - want).abs().max() < TOL
    assert np.array_equal(O.landmark_argmax(got, 1, 64).numpy(), g["argmax"])


def test_main_only_fixture(golden_dir):
    g = np.load(os.path.join(golden_dir, "mainonly_f16.npz"))
    hip, ref = model_pair(16, 2, 3, main_only=True, seed=int(g["weight_seed"]))
    topo, ei, nt, bi = graph_tensors(16, 2, int(g["batch"]), main_only=True)
    frames = synthetic_frames(int(g["batch"]), 128, 16, int(g["frame_seed"]))
    got, want, _, _ = _run_both(hip, ref, frames, ei, nt, bi)
    assert np.abs(got.numpy() - g["logits"]).max() < TOL


@pytest.mark.parametrize("frame,naux,layers,batch,main_only", [
    (224, 7, 3, 2, False),      # BASELINE cfg 2 shape (default.yml), reduced batch so the oracle takes seconds
    (224, 7, 3, 2, True),       # BASELINE cfg 3 shape
    (448, 8, 3, 1, False),      # BASELINE cfg 5 shape (448x448, 8 aux levels, N = 288,084): crop offset 16, 9-level pyramid
    (448, 7, 2, 1, False),      # the degenerate 448 / naux = 7 wiring (a 48^2 corner of level 7 links to the frame)
    (64, 6, 3, 3, False), (30, 3, 2, 2, False), (17, 3, 1, 2, False),
])
def test_stack_vs_oracle_from_node_features(frame, naux, layers, batch, main_only):
    hip, ref = model_pair(frame, naux, layers, main_only=main_only, seed=frame + layers)
    topo, ei, nt, bi = graph_tensors(frame, naux, batch, main_only=main_only)
    feats = synthetic_node_feats(batch * topo.num_nodes, 128, seed=200)
    with torch.no_grad():
        want, _ = ref.forward_nodes(feats, ei, nt, batch)
        got, _ = hip.forward_nodes(feats.to(DEV), ei.to(DEV), batch)
    got = got.cpu()
    err = (got - want).abs().max().item()
    assert err < TOL, err
    assert torch.equal(O.landmark_argmax(got, batch, frame), O.landmark_argmax(want, batch, frame))
    assert (O.landmark_expected_coords(got, batch, frame) - O.landmark_expected_coords(want, batch, frame)).abs().max() < 1e-3


def test_generic_edge_index_falls_back_to_csr_kernel():
    """A graph that is no closed form (here: the model's grid with 40 random extra edges) -> CSR path, same answer as the oracle.
    ('grid-diagonal' graphs, the fallback's customers until round 3, are closed forms now: tests/test_gpu_diag.py.)"""
    hip, ref = model_pair(16, 3, 2, seed=8)
    topo, ei, nt, bi = graph_tensors(16, 3, 2)
    rs = np.random.RandomState(5)
    extra = torch.from_numpy(rs.randint(0, 2 * topo.num_nodes, (2, 40)).astype(np.int64))
    extra = extra[:, extra[0] != extra[1]]
    ei = torch.cat([ei, extra, extra.flip(0)], dim=1)
    feats = synthetic_node_feats(2 * topo.num_nodes, 128, seed=9)
    with torch.no_grad():
        want, _ = ref.forward_nodes(feats, ei, nt, 2)
        got, _ = hip.forward_nodes(feats.to(DEV), ei.to(DEV), 2)
    assert not hip._resolver.resolve(ei.to(DEV), feats.shape[0])[0].structured
    assert (got.cpu() - want).abs().max() < TOL


def test_empty_and_ragged_batches():
    hip, ref = model_pair(8, 2, 2, seed=4)
    topo, ei, nt, bi = graph_tensors(8, 2, 1)
    feats = synthetic_node_feats(topo.num_nodes, 128, seed=1)
    with torch.no_grad():
        got, _ = hip.forward_nodes(feats.to(DEV), ei.to(DEV), 1)
        want, _ = ref.forward_nodes(feats, ei, nt, 1)
    assert (got.cpu() - want).abs().max() < TOL
    with pytest.raises(RuntimeError):
        hip.forward_nodes(feats[:-1].to(DEV), ei.to(DEV), 1)


def test_round_trip_properties_at_full_size():
    """BASELINE cfg 2 at full batch 8: size-independent checks (linearity of the aggregation,
    frame independence, determinism) instead of an oracle run."""
    from echoglad_amd import ops
    B = 8
    g = ops.Graph.topo(224, 7)
    n = g.num_nodes
    x = synthetic_node_feats(B * n, 128, seed=3).to(DEV)
    y = synthetic_node_feats(B * n, 128, seed=4).to(DEV)
    ax, ay = ops.gcn_aggregate(g, B, x), ops.gcn_aggregate(g, B, y)
    axy = ops.gcn_aggregate(g, B, 2.0 * x - 0.5 * y)
    assert (axy - (2.0 * ax - 0.5 * ay)).abs().max() < 1e-4
    # A_hat is symmetric: <A x, y> == <x, A y>
    lhs, rhs = (ax.double() * y.double()).sum(), (x.double() * ay.double()).sum()
    assert abs(lhs - rhs) / abs(lhs) < 1e-6
    # frames are independent units: frame 3 alone == frame 3 inside the batch
    one = ops.gcn_aggregate(g, 1, x[3 * n:4 * n].contiguous())
    assert torch.equal(one, ax[3 * n:4 * n])
    hip, _ = model_pair(224, 7, 3, seed=5)
    topo, ei, nt, bi = graph_tensors(224, 7, B)
    with torch.no_grad():
        a, _ = hip.forward_nodes(x, ei.to(DEV), B)
        b, _ = hip.forward_nodes(x, ei.to(DEV), B)
        c, _ = hip.forward_nodes(x[3 * n:4 * n].contiguous(), torch.from_numpy(topo.edge_index()).to(DEV), 1)
    assert torch.equal(a, b)
    nv = topo.num_valid_nodes
    assert torch.equal(a[3 * nv:4 * nv], c)


def _full_size_properties(frame, naux, B, main_only=False, layers=3, seed=5):
    """Size-independent checks at a BASELINE config's full per-GPU batch: linearity and symmetry of the aggregation,
    frame independence, determinism of the whole stack, and one frame of the batch against the CPU oracle."""
    from echoglad_amd import ops
    g = ops.Graph.topo(frame, naux, main_only)
    n = g.num_nodes
    x = synthetic_node_feats(B * n, 128, seed=3).to(DEV)
    y = synthetic_node_feats(B * n, 128, seed=4).to(DEV)
    ax, ay = ops.gcn_aggregate(g, B, x), ops.gcn_aggregate(g, B, y)
    axy = ops.gcn_aggregate(g, B, 2.0 * x - 0.5 * y)
    assert (axy - (2.0 * ax - 0.5 * ay)).abs().max() < 1e-4
    lhs, rhs = (ax.double() * y.double()).sum(), (x.double() * ay.double()).sum()
    assert abs(lhs - rhs) / abs(lhs) < 1e-6
    f = B - 2
    one = ops.gcn_aggregate(g, 1, x[f * n:(f + 1) * n].contiguous())
    assert torch.equal(one, ax[f * n:(f + 1) * n])
    del ax, ay, axy, y
    hip, ref = model_pair(frame, naux, layers, main_only=main_only, seed=seed)
    topo, ei, nt, bi = graph_tensors(frame, naux, B, main_only=main_only)
    ei1 = torch.from_numpy(topo.edge_index())
    with torch.no_grad():
        a, _ = hip.forward_nodes(x, ei.to(DEV), B)
        b, _ = hip.forward_nodes(x, ei.to(DEV), B)
        c, _ = hip.forward_nodes(x[f * n:(f + 1) * n].contiguous(), ei1.to(DEV), 1)
        want, _ = ref.forward_nodes(x[f * n:(f + 1) * n].cpu(), ei1, nt[:n], 1)
    assert torch.equal(a, b)
    nv = topo.num_valid_nodes
    assert torch.equal(a[f * nv:(f + 1) * nv], c)
    assert (c.cpu() - want).abs().max() < TOL
    assert torch.equal(O.landmark_argmax(c.cpu(), 1, frame), O.landmark_argmax(want, 1, frame))
    # the HIP-graph replay path (what bench.py times) gives the same bits
    hip.enable_hip_graph(True)
    with torch.no_grad():
        d = hip.forward_nodes(x, ei.to(DEV), B)[0].clone()
        e = hip.forward_nodes(x, ei.to(DEV), B)[0].clone()
    assert torch.equal(d, a) and torch.equal(e, a)
    # ... and at batch 1, the reference's own setting (configs/default.yml:27): the replayed single-frame step equals the eager
    # single-frame step, i.e. that frame of the batch, bit for bit (what other_configs.cfg2_b1 of the bench line times)
    x1 = x[f * n:(f + 1) * n].contiguous()
    e1 = ei1.to(DEV)
    with torch.no_grad():
        r1 = hip.forward_nodes(x1, e1, 1)[0].clone()
        r2 = hip.forward_nodes(x1, e1, 1)[0].clone()
    assert torch.equal(r1, c) and torch.equal(r2, c)


def test_cfg2_default_full_batch_8_graph_replay_vs_oracle():
    """BASELINE configs[1] at the batch the metric is quoted on (8 per GPU), on the path bench.py times: the HIP-graph replay
    of the chained layers + fused heads gives the eager bits, and frame 6 of the batch equals the CPU oracle on that frame."""
    _full_size_properties(224, 7, 8)


def test_cfg3_main_only_full_batch_32():
    """BASELINE configs[2]: use_main_graph_only, 224x224, batch 32 per GPU."""
    _full_size_properties(224, 7, 32, main_only=True)


def test_cfg5_448_full_batch_8():
    """BASELINE configs[4]: 448x448, 8 aux levels, 8 frames per GPU (N = 288,084 per frame, 1.18 GB of node features)."""
    _full_size_properties(448, 8, 8)


@pytest.mark.parametrize("frame,naux,batch,hidden,cls_hidden", [(16, 3, 2, 64, 16), (64, 6, 1, 64, 16), (16, 3, 2, 96, 32), (16, 3, 2, 128, 16)])
def test_signature_default_widths_take_the_compatibility_route(frame, naux, batch, hidden, cls_hidden):
    """The reference's SIGNATURE defaults are node_hidden_dim = 64, classifier_hidden_dim = 16 (models.py:286-301; default.yml and
    every BASELINE config use 128 / 32, which is what the fused kernels are built for).  Such a model is constructed with the
    reference's parameter shapes (same state_dict) and runs GCNConv on the 128-channel kernels with zero padding, everything
    else as the torch modules it is: eval logits and one train step (p = 0) against the oracle."""
    hip, ref = model_pair(frame, naux, 3, seed=23, node_hidden_dim=hidden, classifier_hidden_dim=cls_hidden)
    assert hip.gnn_layers[0].module_0.lin.weight.shape == (hidden, 128) and hip.node_classifiers[0][0].weight.shape == (cls_hidden, hidden)
    assert set(hip.state_dict()) == set(ref.state_dict())
    topo, ei, nt, bi = graph_tensors(frame, naux, batch)
    frames = synthetic_frames(batch, 128, frame, 5)
    with torch.no_grad():
        want, _ = ref(x=frames, edge_index=ei, node_type=nt, batch_idx=bi)
        got, _ = hip(x=frames.to(DEV), edge_index=ei.to(DEV), node_type=nt.to(DEV), batch_idx=bi.to(DEV))
    assert got.shape == want.shape and (got.cpu() - want).abs().max() < TOL
    assert torch.equal(O.landmark_argmax(got.cpu(), batch, frame), O.landmark_argmax(want, batch, frame))
    for m in list(hip.modules()) + list(ref.modules()):
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    hip.train(); ref.train()
    want, _ = ref(x=frames, edge_index=ei, node_type=nt, batch_idx=bi)
    got, _ = hip(x=frames.to(DEV), edge_index=ei.to(DEV), node_type=nt.to(DEV), batch_idx=bi.to(DEV))
    assert (got.detach().cpu() - want.detach()).abs().max() < 2e-4
    (want ** 2).mean().backward(); (got ** 2).mean().backward()
    rg = dict(ref.named_parameters())
    for name, p in hip.named_parameters():
        g = rg[name].grad
        assert p.grad is not None and p.grad.shape == g.shape, name
        assert (p.grad.cpu() - g).abs().max() < 5e-3 * g.abs().max() + 1e-6, name


@pytest.mark.parametrize("frame,naux,batch,hidden,cls_hidden", [(16, 3, 2, 64, 16), (32, 4, 2, 96, 32), (16, 3, 2, 128, 16)])
def test_signature_default_widths_with_the_coordinate_graph(frame, naux, batch, hidden, cls_hidden):
    """HierarchicalPatchModel(node_hidden_dim=64, classifier_hidden_dim=16, use_coordinate_graph=True) is a legal reference constructor
    call (models.py:286-301, :339-351: the landmark MLP is Linear(hidden + 8, cls_hidden) ...): the compatibility route runs it as its
    torch modules and resamples the coordinate rows with eg_bilinear4_* on zero-padded node rows -- eval logits and coordinates, and
    one train step (p = 0) against the oracle."""
    hip, ref = model_pair(frame, naux, 3, coord=True, seed=29, node_hidden_dim=hidden, classifier_hidden_dim=cls_hidden)
    assert hip.node_coordinate_mlp[0][0].weight.shape == (cls_hidden, hidden + 8) and set(hip.state_dict()) == set(ref.state_dict())
    topo, ei, nt, bi = graph_tensors(frame, naux, batch, coord=True)
    frames = synthetic_frames(batch, 128, frame, 6)
    c0 = initial_coords(batch, frame)
    with torch.no_grad():
        want, wc = ref(x=frames, node_coords=c0.clone(), edge_index=ei, node_type=nt, batch_idx=bi)
        got, gc = hip(x=frames.to(DEV), node_coords=c0.clone().to(DEV), edge_index=ei.to(DEV), node_type=nt.to(DEV), batch_idx=bi.to(DEV))
    assert got.shape == want.shape and (got.cpu() - want).abs().max() < TOL and (gc.cpu() - wc).abs().max() < 1e-4
    assert torch.equal(O.landmark_argmax(got.cpu(), batch, frame), O.landmark_argmax(want, batch, frame))
    for m in list(hip.modules()) + list(ref.modules()):
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    hip.train(); ref.train()
    want, wc = ref(x=frames, node_coords=c0.clone(), edge_index=ei, node_type=nt, batch_idx=bi)
    got, gc = hip(x=frames.to(DEV), node_coords=c0.clone().to(DEV), edge_index=ei.to(DEV), node_type=nt.to(DEV), batch_idx=bi.to(DEV))
    assert (got.detach().cpu() - want.detach()).abs().max() < 2e-4 and (gc.detach().cpu() - wc.detach()).abs().max() < 2e-4
    ((want ** 2).mean() + (wc ** 2).mean() * 1e-3).backward()
    ((got ** 2).mean() + (gc ** 2).mean() * 1e-3).backward()
    rg = dict(ref.named_parameters())
    for name, p in hip.named_parameters():
        g = rg[name].grad
        assert p.grad is not None and p.grad.shape == g.shape, name
        assert (p.grad.cpu() - g).abs().max() < 5e-3 * g.abs().max() + 1e-6, name



@pytest.mark.parametrize("frame,naux,batch,conn", [(64, 6, 2, False), (224, 7, 1, False), (32, 4, 2, True)])
def test_hip_graph_replay_is_reached_through_forward_with_new_frames_every_call(frame, naux, batch, conn):
    """engine.py:251-255 / :394-398 call ``model(x=frames, edge_index=...)`` with NEW frames (and, from a DataLoader, a new
    edge_index tensor of the same content) every step.  With enable_hip_graph(True) the packing in front of the stack writes into
    the model's static node-feature buffer and every call replays ONE captured graph: 5 different frame tensors + a fresh copy of
    the edge_index -> exactly 1 capture, every output bit-equal to the eager forward of the same frames; ``model(data_batch)``
    takes the same route; a parameter update captures again (the folded parameters changed) and follows it."""
    hip, ref = model_pair(frame, naux, 3, seed=31, use_connection_nodes=conn)
    topo, ei, nt, bi = graph_tensors(frame, naux, batch, conn=conn)
    ei_d, nt_d, bi_d = ei.to(DEV), nt.to(DEV), bi.to(DEV)
    frames = [synthetic_frames(batch, 128, frame, 40 + k).to(DEV) for k in range(5)]
    with torch.no_grad():
        eager = [hip(x=f, edge_index=ei_d, node_type=nt_d, batch_idx=bi_d)[0].clone() for f in frames]
        want, _ = ref(x=frames[2].cpu(), edge_index=ei, node_type=nt, batch_idx=bi)
    assert (eager[2].cpu() - want).abs().max() < TOL
    hip.enable_hip_graph(True)
    assert hip.hip_graph_captures == 0
    with torch.no_grad():
        for k, f in enumerate(frames):
            e = ei_d if k % 2 == 0 else ei_d.clone()             # (a fresh tensor with the same edges resolves to the same handle)
            got = hip(x=f, edge_index=e, node_type=nt_d, batch_idx=bi_d)[0]
            assert torch.equal(got, eager[k]), k
        assert hip.hip_graph_captures == 1

        class _Batch:
            pass
        db = _Batch()
        db.x, db.edge_index, db.batch, db.node_type = frames[3], ei_d, bi_d, nt_d
        assert torch.equal(hip(db)[0], eager[3]) and hip.hip_graph_captures == 1
        # the caller's own buffer through forward_nodes: captured once per buffer ADDRESS, whatever tensor object wraps it
        feats = hip.create_node_pixels(frames[1], batch).clone()
        a = hip.forward_nodes(feats, ei_d, batch)[0].clone()
        b = hip.forward_nodes(feats.view(-1, 128), ei_d.clone(), batch)[0].clone()
        assert torch.equal(a, eager[1]) and torch.equal(b, eager[1]) and hip.hip_graph_captures == 2
        # new weights: the next call follows them (one more capture), bit-equal to eager again
        hip.gnn_layers[0].module_0.lin.weight.mul_(1.5)
        got = hip(x=frames[4], edge_index=ei_d, node_type=nt_d, batch_idx=bi_d)[0].clone()
        assert hip.hip_graph_captures == 3
        hip.enable_hip_graph(False)
        assert torch.equal(got, hip(x=frames[4], edge_index=ei_d, node_type=nt_d, batch_idx=bi_d)[0])
    # a call that wants gradients never lands in the static buffer
    hip.enable_hip_graph(True)
    n0 = hip.hip_graph_captures
    x = frames[0].clone().requires_grad_(True)
    out, _ = hip(x=x, edge_index=ei_d, node_type=nt_d, batch_idx=bi_d)
    out.sum().backward()
    assert x.grad is not None and hip.hip_graph_captures == n0
