#!/usr/bin/env python3
"""Generate the committed golden fixtures under tests/golden/ by RUNNING THE
REFERENCE'S OWN CODE in this container.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

/root/reference is imported read-only with its missing third-party modules
stubbed (torch_geometric, torchvision, torchsummary, imageio, cv2).  The stub
for ``torch_geometric.nn.GCNConv`` / ``Sequential`` is the oracle's restatement
(oracle/gnn_oracle.py) because PyG 2.0.2 is not installable here; everything
else that executes — ``DummyDataset.create_graphs`` (datasets.py:1441-1584),
``HierarchicalPatchModel.__init__/forward/create_node_pixels/
bilinear_interpolation`` (models.py:286-553), the criteria (criterion.py) — is
the reference's own code.  Only OUTPUT DATA is written to the repo; no
reference source text or bytecode is copied.

Outputs
  topology.json        per config: N, E, sha256 of sorted undirected edge list, degree histogram
  topo_f8_a2_edges.npy full undirected edge list for F=8 / naux=2
  kat_f16_a3.npz       layer KAT: inputs, weights seed, per-layer outputs, logits (eval)
  cfg1_f64_a2.npz      BASELINE config 1 (64x64, 2 aux, L=2, B=1): sampled rows, digests, argmax
  coord_f32_a4.npz     coordinate-graph path (B=2): coords per layer, logits, train-mode grad norms
  cfg4_f224_a7_coord.npz  BASELINE configs[3] shape (224x224, 7 aux, coordinate graph): eval at B=2 (sampled rows, digests,
                       coordinates per layer, argmax) and a train-mode (p=0) step at B=1 (loss, gradient norms); `cfg4` target
  mainonly_f16.npz     use_main_graph_only ablation
  losses_f16_a3.npz    WeightedBCEWithLogits + ExpectedLandmarkMSE values on the KAT logits
  labels.npz           DummyDataset.create_node_labels for hand-picked coordinates on 5 configs + one full sample
  decode_f16_a3.npz    seeded logits / labels / valid masks (B=3): both losses and their gradients w.r.t. the logits
  decode_f30_a3.npz    (criterion.py, autograd of the reference's own classes) and every number that
                       LandmarkExpectedCoordiantesEvaluator.update records (evaluators.py:291-391)
"""
import hashlib
import json
import os
import sys
import types

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, REPO)
sys.path.insert(0, HERE)

import numpy as np
import torch
import torch.nn as nn

from oracle import gnn_oracle as O
from fixtures_util import fill_state_dict, synthetic_frames, initial_coords


# ----------------------------------------------------------------- stubs
def _from_networkx(G):
    """PyG 2.0.2 ``from_networkx`` semantics: relabel to 0..N-1 in node order,
    directed copy, edges in adjacency order."""
    import networkx as nx
    G = nx.convert_node_labels_to_integers(G)
    G = G.to_directed() if not nx.is_directed(G) else G
    ei = torch.tensor(list(G.edges), dtype=torch.long).t().contiguous().view(2, -1)
    data = types.SimpleNamespace()
    data.edge_index = ei
    data.num_nodes = G.number_of_nodes()
    return data


def install_stubs():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class _Dataset:
        def __init__(self, *a, **k):
            pass

    tg = mod("torch_geometric")
    tg.nn = mod("torch_geometric.nn", GCNConv=O.OracleGCNConv, Sequential=O.OracleSequential,
                global_add_pool=lambda *a, **k: None, JumpingKnowledge=O.OracleJumpingKnowledge)
    tg.data = mod("torch_geometric.data", Dataset=_Dataset)
    tg.utils = mod("torch_geometric.utils", from_networkx=_from_networkx)
    tv = mod("torchvision")
    tv.models = mod("torchvision.models")
    tv.models._utils = mod("torchvision.models._utils", IntermediateLayerGetter=object)
    tv.transforms = mod("torchvision.transforms")
    tv.transforms.functional = mod("torchvision.transforms.functional", hflip=lambda x: x)
    mod("torchsummary", summary=lambda *a, **k: None)
    mod("imageio")
    mod("cv2")
    try:
        import matplotlib  # noqa: F401
    except Exception:
        mp = mod("matplotlib")
        mp.pyplot = mod("matplotlib.pyplot", new_figure_manager=None)


install_stubs()
sys.path.insert(0, REF)
from src.core import models as RM            # noqa: E402  (reference code, executed not copied)
from src.core import datasets as RD          # noqa: E402
from src.core import criterion as RC         # noqa: E402

from echoglad_amd.topology import TopologySpec, HierTopology   # noqa: E402


def digest(arr: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(arr).tobytes()).hexdigest()


# ------------------------------------------------------------ topology
TOPO_CONFIGS = [
    # (frame, naux, main_only, coord, conn, main_type, aux_type)
    (8, 2, False, False, False, "grid", "grid"),
    (16, 3, False, False, False, "grid", "grid"),
    (16, 3, False, True, False, "grid", "grid"),
    (16, 3, False, False, True, "grid", "grid"),
    (16, 3, False, True, True, "grid", "grid"),
    (16, 3, False, False, False, "grid-diagonal", "grid-diagonal"),
    (16, 2, True, False, False, "grid", "grid"),
    (32, 4, False, True, False, "grid", "grid"),
    (64, 2, False, False, False, "grid", "grid"),          # BASELINE cfg 1 (slice-clamp quirk)
    (30, 3, False, False, False, "grid", "grid"),          # odd crop
    (17, 3, False, False, False, "grid", "grid"),          # odd frame
    (8, 1, False, False, False, "grid", "grid"),           # negative centre -> wraps to last row
    (48, 5, False, False, False, "grid", "grid-diagonal"),
    (64, 6, False, False, False, "grid", "grid"),
    (224, 7, False, False, False, "grid", "grid"),         # BASELINE cfg 2 (default.yml)
    (224, 7, True, False, False, "grid", "grid"),          # BASELINE cfg 3
    (224, 7, False, True, False, "grid", "grid"),          # BASELINE cfg 4
    (448, 8, False, False, False, "grid", "grid"),         # BASELINE cfg 5 (c0 = 16, 9-level pyramid; ~3 min in the reference's O(n^2) builder)
    (448, 7, False, False, False, "grid", "grid"),         # the degenerate naux for 448: only a 48^2 corner of level 7 links to the frame
]


def reference_graph(frame, naux, main_only, coord, conn, main_type, aux_type):
    ds = RD.DummyDataset(data_dir=None, data_info_file=None, mode="train", num_aux_graphs=naux,
                         frame_size=frame, main_graph_type=main_type, aux_graph_type=aux_type,
                         use_coordinate_graph=coord, use_connection_nodes=conn, use_main_graph_only=main_only)
    g = _from_networkx(ds.graphs)
    return g.edge_index.numpy(), np.asarray(ds.node_type, dtype=np.float64), g.num_nodes


def topo_entry(ei, node_type, n):
    lo, hi = np.minimum(ei[0], ei[1]), np.maximum(ei[0], ei[1])
    und = np.unique(np.stack([lo, hi], axis=1), axis=0)          # sorted lexicographically
    deg = np.bincount(ei[1], minlength=n)
    vals, cnt = np.unique(deg, return_counts=True)
    return {
        "num_nodes": int(n),
        "num_directed_edges": int(ei.shape[1]),
        "num_undirected_edges": int(und.shape[0]),
        "edge_sha256": digest(und.astype("<i8")),
        "degree_hist": {str(int(v)): int(c) for v, c in zip(vals, cnt)},
        "node_type_sha256": digest(node_type.astype("<f8")),
        "num_valid": int((node_type == 0).sum()),
    }, und


def make_topology():
    out = {}
    for cfg in TOPO_CONFIGS:
        frame, naux, main_only, coord, conn, mt, at = cfg
        if frame >= 224 and os.environ.get("GOLDEN_SKIP_224"):
            continue
        if frame >= 448 and os.environ.get("GOLDEN_SKIP_448"):
            continue
        ei, nt, n = reference_graph(*cfg)
        entry, und = topo_entry(ei, nt, n)
        key = f"F{frame}_A{naux}_mo{int(main_only)}_co{int(coord)}_cn{int(conn)}_{mt}_{at}"
        out[key] = entry
        # cross-check the closed form right here
        topo = HierTopology(TopologySpec(frame, naux, main_only, coord, conn, mt, at))
        assert topo.num_nodes == n, (key, topo.num_nodes, n)
        assert topo.edge_set_digest() == entry["edge_sha256"], key
        assert np.array_equal(topo.node_type(), nt), key
        print("topology ok", key, entry["num_nodes"], entry["num_undirected_edges"], flush=True)
        if (frame, naux) == (8, 2) and not (coord or conn or main_only) and mt == "grid":
            np.save(os.path.join(HERE, "topo_f8_a2_edges.npy"), und.astype(np.int64))
    with open(os.path.join(HERE, "topology.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)


# ------------------------------------------------------------ model fixtures
def build_ref_model(frame, naux, layers, coord=False, main_only=False, C=128, hidden=128, clf=32):
    m = RM.HierarchicalPatchModel(frame_size=frame, gnn_dropout_p=0.5, classifier_dropout_p=0.5,
                                  node_embedding_dim=C, node_hidden_dim=hidden, num_output_channels=4,
                                  num_gnn_layers=layers, num_aux_graphs=naux, gnn_jk_mode="last",
                                  classifier_hidden_dim=clf, residual=True, use_coordinate_graph=coord,
                                  output_activation="logit", use_connection_nodes=False,
                                  use_main_graph_only=main_only)
    return m


def collate(ei, nt, n, batch):
    """PyG Batch collate: per-sample node offset on edge_index, cat on dim 0."""
    eis = [torch.from_numpy(ei) + b * n for b in range(batch)]
    edge_index = torch.cat(eis, dim=1)
    node_type = torch.from_numpy(np.tile(nt, batch))
    batch_idx = torch.arange(batch).repeat_interleave(n)
    return edge_index, node_type, batch_idx


def run_ref(model, frames, edge_index, node_type, batch_idx, coords=None):
    layer_out = []
    hooks = [l.register_forward_hook(lambda m, i, o: layer_out.append(o.detach().clone())) for l in model.gnn_layers]
    node_feats = []
    orig = model.create_node_pixels

    def wrapped(*a, **k):
        r = orig(*a, **k)
        node_feats.append(r.detach().clone())
        return r
    model.create_node_pixels = wrapped
    c = coords.clone() if coords is not None else None       # reference mutates it in place (models.py:450)
    logits, out_coords = model(x=frames, node_coords=c, edge_index=edge_index, node_type=node_type,
                               batch_idx=batch_idx)
    for h in hooks:
        h.remove()
    model.create_node_pixels = orig
    return logits, out_coords, node_feats[0], layer_out


def make_kat():
    frame, naux, L, B = 16, 3, 3, 2
    ei, nt, n = reference_graph(frame, naux, False, False, False, "grid", "grid")
    edge_index, node_type, batch_idx = collate(ei, nt, n, B)
    model = build_ref_model(frame, naux, L)
    fill_state_dict(model, seed=1234)
    model.eval()
    frames = synthetic_frames(B, 128, frame, seed=200)
    with torch.no_grad():
        logits, _, node_feats, layer_out = run_ref(model, frames, edge_index, node_type, batch_idx)
        # dense fp64 cross-check of the first GCN layer output (pre-BN)
        conv = model.gnn_layers[0].module_0
        d64 = O.gcn_conv_dense64(node_feats, edge_index, conv.lin.weight, conv.bias)
        s32 = O.gcn_conv_sparse(node_feats, edge_index, conv.lin.weight, conv.bias)
        assert (d64 - s32.double()).abs().max() < 1e-5
    np.savez_compressed(os.path.join(HERE, "kat_f16_a3.npz"),
                        frame=frame, naux=naux, layers=L, batch=B, weight_seed=1234, frame_seed=200,
                        edge_index=edge_index.numpy(), node_type=node_type.numpy(),
                        node_feats=node_feats.numpy(),
                        layer0=layer_out[0].numpy(), layer1=layer_out[1].numpy(), layer2=layer_out[2].numpy(),
                        gcn0_dense64=d64.numpy(),
                        logits=logits.numpy())
    print("kat", logits.shape, float(logits.abs().max()))
    return model, logits, B, frame, naux


def make_cfg1():
    frame, naux, L, B = 64, 2, 2, 1
    ei, nt, n = reference_graph(frame, naux, False, False, False, "grid", "grid")
    edge_index, node_type, batch_idx = collate(ei, nt, n, B)
    model = build_ref_model(frame, naux, L)
    fill_state_dict(model, seed=4321)
    model.eval()
    frames = synthetic_frames(B, 128, frame, seed=200)
    with torch.no_grad():
        logits, _, node_feats, layer_out = run_ref(model, frames, edge_index, node_type, batch_idx)
    rows = np.linspace(0, n - 1, 64).astype(np.int64)
    arg = O.landmark_argmax(logits, B, frame)
    np.savez_compressed(os.path.join(HERE, "cfg1_f64_a2.npz"),
                        frame=frame, naux=naux, layers=L, batch=B, weight_seed=4321, frame_seed=200,
                        num_nodes=n, sample_rows=rows, logits_rows=logits.numpy()[rows],
                        node_feats_rows=node_feats.numpy()[rows],
                        layer1_rows=layer_out[1].numpy()[rows],
                        logits_sum=np.float64(logits.double().sum().item()),
                        logits_abs_sum=np.float64(logits.double().abs().sum().item()),
                        argmax=arg.numpy())
    print("cfg1", logits.shape, arg.tolist())


def make_coord():
    frame, naux, L, B = 32, 4, 3, 2
    ei, nt, n = reference_graph(frame, naux, False, True, False, "grid", "grid")
    edge_index, node_type, batch_idx = collate(ei, nt, n, B)
    model = build_ref_model(frame, naux, L, coord=True)
    fill_state_dict(model, seed=777)
    frames = synthetic_frames(B, 128, frame, seed=201)
    coords0 = initial_coords(B, frame)
    # eval forward; capture coordinates after every layer through the coordinate MLP hook
    model.eval()
    deltas = []
    hooks = [m.register_forward_hook(lambda mod, i, o: deltas.append(o.detach().clone()))
             for m in model.node_coordinate_mlp]
    with torch.no_grad():
        logits, out_coords, node_feats, layer_out = run_ref(model, frames, edge_index, node_type, batch_idx, coords0)
    for h in hooks:
        h.remove()
    # train-mode (dropout p=0 so it is deterministic) forward+backward: gradient fixtures
    model_t = build_ref_model(frame, naux, L, coord=True)
    for mod in model_t.modules():
        if isinstance(mod, nn.Dropout):
            mod.p = 0.0
    fill_state_dict(model_t, seed=777)
    model_t.train()
    logits_t, coords_t, _, _ = run_ref(model_t, frames, edge_index, node_type, batch_idx, coords0)
    loss = (logits_t ** 2).mean() + (coords_t ** 2).mean() * 1e-3
    loss.backward()
    gn = {k: float(p.grad.double().norm()) for k, p in model_t.named_parameters() if p.grad is not None}
    keys = sorted(gn)
    np.savez_compressed(os.path.join(HERE, "coord_f32_a4.npz"),
                        frame=frame, naux=naux, layers=L, batch=B, weight_seed=777, frame_seed=201,
                        coords0=coords0.numpy(), deltas=np.stack([d.numpy() for d in deltas]),
                        out_coords=out_coords.numpy(), logits=logits.numpy(),
                        node_feats=node_feats.numpy(),
                        layer2=layer_out[2].numpy(),
                        train_logits=logits_t.detach().numpy(), train_coords=coords_t.detach().numpy(),
                        train_loss=np.float64(loss.item()),
                        grad_keys=np.array(keys), grad_norms=np.array([gn[k] for k in keys], dtype=np.float64))
    print("coord", logits.shape, out_coords.tolist())


def make_cfg4():
    """BASELINE configs[3] shape: 224x224, 7 aux levels, coordinate graph on, L=3.  Eval forward at B=2 and one train-mode
    (dropout p=0) forward+backward at B=1 through the reference's own forward (dense bilinear_interpolation included)."""
    frame, naux, L, B = 224, 7, 3, 2
    ei, nt, n = reference_graph(frame, naux, False, True, False, "grid", "grid")
    edge_index, node_type, batch_idx = collate(ei, nt, n, B)
    model = build_ref_model(frame, naux, L, coord=True)
    fill_state_dict(model, seed=2024)
    frames = synthetic_frames(B, 128, frame, seed=203)
    coords0 = initial_coords(B, frame)
    model.eval()
    deltas = []
    hooks = [m.register_forward_hook(lambda mod, i, o: deltas.append(o.detach().clone())) for m in model.node_coordinate_mlp]
    with torch.no_grad():
        logits, out_coords, node_feats, layer_out = run_ref(model, frames, edge_index, node_type, batch_idx, coords0)
    for h in hooks:
        h.remove()
    nv = logits.shape[0] // B
    rows = np.unique(np.concatenate([np.linspace(0, B * nv - 1, 384).astype(np.int64),
                                     np.arange(nv - 8, nv + 8), np.arange(0, 24)]))
    hrows = np.unique(np.concatenate([np.linspace(0, B * n - 1, 256).astype(np.int64), np.arange(n - 6, n + 6)]))
    arg = O.landmark_argmax(logits, B, frame)
    out = dict(frame=frame, naux=naux, layers=L, batch=B, weight_seed=2024, frame_seed=203, num_nodes=n,
               coords0=coords0.numpy(), deltas=np.stack([d.numpy() for d in deltas]), out_coords=out_coords.numpy(),
               sample_rows=rows, logits_rows=logits.numpy()[rows], hidden_rows=hrows,
               node_feats_rows=node_feats.numpy()[hrows], layer2_rows=layer_out[2].numpy()[hrows],
               logits_sum=np.float64(logits.double().sum().item()),
               logits_abs_sum=np.float64(logits.double().abs().sum().item()), argmax=arg.numpy())
    print("cfg4 eval", logits.shape, out_coords.tolist(), flush=True)
    # train mode, p = 0, B = 1
    Bt = 1
    edge_index, node_type, batch_idx = collate(ei, nt, n, Bt)
    model_t = build_ref_model(frame, naux, L, coord=True)
    for mod in model_t.modules():
        if isinstance(mod, nn.Dropout):
            mod.p = 0.0
    fill_state_dict(model_t, seed=2024)
    model_t.train()
    frames_t = synthetic_frames(Bt, 128, frame, seed=204)
    logits_t, coords_t, _, _ = run_ref(model_t, frames_t, edge_index, node_type, batch_idx, initial_coords(Bt, frame))
    loss = (logits_t ** 2).mean() + (coords_t ** 2).mean() * 1e-3
    loss.backward()
    gn = {k: float(p.grad.double().norm()) for k, p in model_t.named_parameters() if p.grad is not None}
    keys = sorted(gn)
    trows = np.linspace(0, logits_t.shape[0] - 1, 256).astype(np.int64)
    out.update(train_frame_seed=204, train_rows=trows, train_logits_rows=logits_t.detach().numpy()[trows],
               train_coords=coords_t.detach().numpy(), train_loss=np.float64(loss.item()), grad_keys=np.array(keys),
               grad_norms=np.array([gn[k] for k in keys], dtype=np.float64),
               grad_w0=model_t.gnn_layers[0].module_0.lin.weight.grad.numpy()[::8, ::8].copy())
    # The same train step in fp64 (same reference classes, parameters and inputs cast to double): what the fp32 numbers above
    # are an approximation OF.  The GPU test derives its tolerances from it: |HIP - fp64| <= 2 |reference fp32 - fp64| per
    # quantity, instead of asserting a chosen 3e-4 / 5e-3.
    model_64 = build_ref_model(frame, naux, L, coord=True)
    for mod in model_64.modules():
        if isinstance(mod, nn.Dropout):
            mod.p = 0.0
    fill_state_dict(model_64, seed=2024)
    model_64 = model_64.double()
    model_64.train()
    logits_64, coords_64, _, _ = run_ref(model_64, frames_t.double(), edge_index, node_type, batch_idx, initial_coords(Bt, frame).double())
    loss_64 = (logits_64 ** 2).mean() + (coords_64 ** 2).mean() * 1e-3
    loss_64.backward()
    g32 = {k: p.grad.detach() for k, p in model_t.named_parameters() if p.grad is not None}
    g64 = {k: p.grad.detach() for k, p in model_64.named_parameters() if p.grad is not None}
    assert sorted(g64) == keys
    offs, idxs, s32, s64 = [0], [], [], []
    for k in keys:
        m = g64[k].numel()
        idx = np.unique(np.linspace(0, m - 1, min(m, 256)).astype(np.int64))
        idxs.append(idx)
        s32.append(g32[k].reshape(-1).numpy()[idx])
        s64.append(g64[k].reshape(-1).numpy()[idx])
        offs.append(offs[-1] + len(idx))
    out.update(train64_logits_rows=logits_64.detach().numpy()[trows], train64_coords=coords_64.detach().numpy(),
               train64_loss=np.float64(loss_64.item()),
               train_logits_ref_err=np.float64((logits_t.detach().double() - logits_64.detach()).abs().max().item()),
               train_coords_ref_err=np.float64((coords_t.detach().double() - coords_64.detach()).abs().max().item()),
               grad64_norms=np.array([float(g64[k].norm()) for k in keys], dtype=np.float64),
               grad64_maxabs=np.array([float(g64[k].abs().max()) for k in keys], dtype=np.float64),
               grad_ref_err=np.array([float((g32[k].double() - g64[k]).abs().max()) for k in keys], dtype=np.float64),
               grad_sample_offsets=np.array(offs, dtype=np.int64), grad_sample_idx=np.concatenate(idxs),
               grad32_samples=np.concatenate(s32).astype(np.float32), grad64_samples=np.concatenate(s64).astype(np.float64))
    np.savez_compressed(os.path.join(HERE, "cfg4_f224_a7_coord.npz"), **out)
    print("cfg4 train", float(loss), "fp64", float(loss_64), "logits ref err", float(out["train_logits_ref_err"]), flush=True)


def make_unet_keys():
    """state_dict keys and shapes of the reference's UNet variant (models.py:639-756, the class configs/default.yml names): what
    echoglad_amd.examples.UNetNodeFeatureModel must register to load a reference checkpoint strict=True."""
    # (encoder_embedding_dims must be truthy: models.py:652-653 overwrites a supplied list with the default and crashes on None)
    m = RM.UNETHierarchicalPatchModel(encoder_embedding_dims=[1], frame_size=224, gnn_dropout_p=0.5, classifier_dropout_p=0.5, node_embedding_dim=128,
                                      node_hidden_dim=128, num_output_channels=4, num_gnn_layers=3, num_aux_graphs=7,
                                      gnn_jk_mode="last", classifier_hidden_dim=32, residual=True, use_coordinate_graph=True,
                                      output_activation="logit", use_connection_nodes=False, use_main_graph_only=False)
    keys = {k: list(v.shape) for k, v in m.state_dict().items()}
    json.dump({"n_parameters": int(sum(p.numel() for p in m.parameters())), "state_dict": keys},
              open(os.path.join(HERE, "unet_state_keys.json"), "w"), indent=0, sort_keys=True)
    print("unet keys", len(keys))


def make_mainonly():
    frame, naux, L, B = 16, 2, 3, 2
    ei, nt, n = reference_graph(frame, naux, True, False, False, "grid", "grid")
    edge_index, node_type, batch_idx = collate(ei, nt, n, B)
    model = build_ref_model(frame, naux, L, main_only=True)
    fill_state_dict(model, seed=99)
    model.eval()
    frames = synthetic_frames(B, 128, frame, seed=202)
    with torch.no_grad():
        logits, _, node_feats, layer_out = run_ref(model, frames, edge_index, node_type, batch_idx)
    np.savez_compressed(os.path.join(HERE, "mainonly_f16.npz"),
                        frame=frame, naux=naux, layers=L, batch=B, weight_seed=99, frame_seed=202,
                        node_feats=node_feats.numpy(), logits=logits.numpy())
    print("mainonly", logits.shape)


def make_losses(model, logits, B, frame, naux):
    """criterion.py:13-27 and :93-151 evaluated by the reference's own classes."""
    rs = np.random.RandomState(5)
    topo = HierTopology(TopologySpec(frame, naux))
    n = topo.num_nodes
    y = np.zeros((B, n, 4), dtype=np.float32)
    # one-hot per level per channel, like create_node_labels (datasets.py:1586-1612)
    levels = [(lv.base, lv.side) for lv in topo.aux_levels] + [(topo.main.base, topo.main.side)]
    for b in range(B):
        for ch in range(4):
            hh, ww = rs.randint(0, frame, size=2)
            for base, side in levels:
                r, c = hh * side // frame, ww * side // frame
                y[b, base + r * side + c, ch] = 1.0
    y_t = torch.from_numpy(y).view(B * n, 4)
    valid = torch.ones_like(y_t)
    bce = RC.WeightedBCEWithLogitsLoss(reduction="none", ones_weight=9000, loss_weight=1)
    elm = RC.ExpectedLandmarkMSE(loss_weight=10, batch_size=B, frame_size=frame, num_aux_graphs=naux,
                                 use_main_graph_only=False, num_output_channels=4)
    with torch.no_grad():
        l_bce = bce.compute(logits.view(B, n, 4), y_t.view(B, n, 4), valid)
        l_elm = elm.compute(logits, y_t, valid)
    np.savez_compressed(os.path.join(HERE, "losses_f16_a3.npz"), labels=y.reshape(B * n, 4),
                        bce=np.float64(l_bce.item()), elm=np.float64(l_elm.item()))
    print("losses", float(l_bce), float(l_elm))


def make_labels():
    """DummyDataset.create_node_labels (datasets.py:1586-1612) and one whole __getitem__ sample (:1381-1415) from the
    reference's own class: positions of the ones per landmark for hand-picked coordinates (corners, -1 wrap, bin edges)."""
    out = {}
    for frame, naux, main_only in ((16, 3, False), (30, 3, False), (64, 2, False), (224, 7, False), (16, 2, True)):
        ds = RD.DummyDataset(data_dir=None, data_info_file=None, mode="train", num_aux_graphs=naux, frame_size=frame,
                             main_graph_type="grid", aux_graph_type="grid", use_coordinate_graph=False,
                             use_connection_nodes=False, use_main_graph_only=main_only)
        coords = np.array([[0, 0], [frame - 1, frame - 1], [-1, 3], [frame // 2, frame // 2 - 1], [5, frame - 2], [frame // 4, frame // 8],
                           [7, -1], [frame - 1, 0]])
        ones = []
        for c in coords:
            y = ds.create_node_labels(c).numpy()[:, 0]
            assert set(np.unique(y)) <= {0.0, 1.0}
            ones.append(np.nonzero(y)[0])
        key = f"F{frame}_A{naux}_mo{int(main_only)}"
        out[key + "_coords"] = coords
        out[key + "_ones"] = np.stack(ones)              # one label per level: [n_coords, n_levels]
        out[key + "_len"] = np.int64(len(y))
    # one full sample under a fixed numpy seed (extract_coords draws from np.random, the frame from torch.randn)
    np.random.seed(77)
    torch.manual_seed(77)
    ds = RD.DummyDataset(data_dir=None, data_info_file=None, mode="train", num_aux_graphs=3, frame_size=16,
                         main_graph_type="grid", aux_graph_type="grid", use_coordinate_graph=True,
                         use_connection_nodes=False, use_main_graph_only=False,
                         transform=lambda t: torch.nn.functional.interpolate(t.unsqueeze(0), size=(16, 16)).squeeze(0))
    np.random.seed(78)
    g = ds[0]
    out["sample_y"] = g.y.numpy()
    out["sample_valid"] = g.valid_labels.numpy()
    out["sample_node_type"] = g.node_type.numpy()
    out["sample_node_coords"] = g.node_coords.numpy()
    out["sample_node_coord_y"] = g.node_coord_y.numpy()
    out["sample_pix2mm"] = np.array([float(g.pix2mm_x), float(g.pix2mm_y)])
    out["sample_x_shape"] = np.array(g.x.shape)
    np.random.seed(78)
    out["sample_draws"] = np.random.randint(low=0, high=16, size=12)     # the 3 x 4 integers extract_coords consumes
    np.savez_compressed(os.path.join(HERE, "labels.npz"), **out)
    print("labels", {k: v.shape for k, v in out.items() if k.endswith("_ones")}, out["sample_node_coord_y"].tolist())


def make_decode(frame, naux, B, seed, tag):
    """Losses + gradients (criterion.py:13-27,93-151) and landmark decode / width errors (evaluators.py:291-391,
    485-495) from the reference's own classes on seeded logits with near-ties and partly invalid labels."""
    from src.core import evaluators as RE
    rs = np.random.RandomState(seed)
    topo = HierTopology(TopologySpec(frame, naux))
    n = topo.num_nodes
    levels = [(lv.base, lv.side) for lv in topo.aux_levels] + [(topo.main.base, topo.main.side)]
    logits = (rs.standard_normal((B, n, 4)) * 2.0).astype(np.float32)
    y = np.zeros((B, n, 4), dtype=np.float32)
    for b in range(B):
        for ch in range(4):
            hh, ww = rs.randint(0, frame, size=2)
            ph, pw = rs.randint(0, frame, size=2)                       # where the prediction peaks
            for base, side in levels:
                y[b, base + (hh * side // frame) * side + (ww * side // frame), ch] = 1.0
                logits[b, base + (ph * side // frame) * side + (pw * side // frame), ch] += 6.0
    # a near-tie (two equal maxima) in frame 0, channel 1 of the main grid: first index must win
    mb, ms = levels[-1]
    logits[0, mb + 3 * ms + 5, 1] = logits[0, mb + 9 * ms + 2, 1] = 30.0
    valid = np.ones((B, n, 4), dtype=np.float32)
    valid[B - 1, :, 2] = 0.0                                             # one landmark unlabeled in the last frame
    if B > 1:
        valid[1, :, 0] = 0.0
    lg = torch.from_numpy(logits).view(B * n, 4).requires_grad_(True)
    y_t, v_t = torch.from_numpy(y).view(B * n, 4), torch.from_numpy(valid).view(B * n, 4)
    bce = RC.WeightedBCEWithLogitsLoss(reduction="none", ones_weight=9000, loss_weight=1)
    elm = RC.ExpectedLandmarkMSE(loss_weight=10, batch_size=B, frame_size=frame, num_aux_graphs=naux,
                                 use_main_graph_only=False, num_output_channels=4)
    l_bce = bce.compute(lg.view(B, n, 4), y_t.view(B, n, 4), v_t)
    g_bce, = torch.autograd.grad(l_bce, lg)
    l_elm = elm.compute(lg, y_t, v_t)
    g_elm, = torch.autograd.grad(l_elm, lg)
    ev = RE.LandmarkExpectedCoordiantesEvaluator(logger=None, batch_size=B, frame_size=frame, use_coord_graph=False)
    pix_x = torch.from_numpy(rs.uniform(0.2, 0.6, B).astype(np.float32))
    pix_y = torch.from_numpy(rs.uniform(0.2, 0.6, B).astype(np.float32))
    ev.update(lg.detach(), y_t, pix_x, pix_y, v_t)
    last = ev.get_last()
    det = ev.get_predictions()
    co = det["coordinates"]
    names = ["lvid_top", "lvid_bot", "lvpw", "ivs"]
    np.savez_compressed(
        os.path.join(HERE, f"decode_{tag}.npz"), frame=frame, naux=naux, batch=B,
        logits=logits.reshape(B * n, 4), labels=y.reshape(B * n, 4), valid=valid.reshape(B * n, 4),
        pix2mm_x=pix_x.numpy(), pix2mm_y=pix_y.numpy(),
        bce=np.float64(l_bce.item()), elm=np.float64(l_elm.item()), grad_bce=g_bce.numpy(), grad_elm=g_elm.numpy(),
        pred_coords=np.stack([co["pred_" + k].numpy() for k in names], axis=1),          # [B,4,2] (h,w)
        gt_coords=np.stack([co["gt_" + k].numpy() for k in names], axis=1).astype(np.int64),
        last_keys=np.array(sorted(last.keys())), last_vals=np.array([float(last[k]) for k in sorted(last.keys())]),
        width_keys=np.array(sorted(det["widths"].keys())),
        width_vals=np.stack([det["widths"][k].numpy() for k in sorted(det["widths"].keys())], axis=0),
        argmax_main=lg.detach().view(B, n, 4)[:, -frame * frame:, :].argmax(dim=1).numpy())
    print("decode", tag, float(l_bce), float(l_elm), {k: round(float(v), 4) for k, v in last.items()})


if __name__ == "__main__":
    torch.manual_seed(0)
    torch.set_num_threads(8)
    which = sys.argv[1:] or ["topology", "models"]
    if "topology" in which:
        make_topology()
    if "models" in which:
        model, logits, B, frame, naux = make_kat()
        make_losses(model, logits, B, frame, naux)
        make_cfg1()
        make_coord()
        make_mainonly()
    if "labels" in which or "models" in which:
        make_labels()
    if "cfg4" in which:
        make_cfg4()
    if "unet" in which or "models" in which:
        make_unet_keys()
    if "decode" in which or "models" in which:
        make_decode(16, 3, 3, 11, "f16_a3")
        make_decode(30, 3, 2, 12, "f30_a3")
    assert not os.path.exists(os.path.join(REF, "src", "__pycache__")), "bytecode leaked into the reference"
