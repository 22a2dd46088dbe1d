"""Debug: parameter-gradient error of one train step, HIP path and the fp32 CPU oracle, both against the oracle in fp64.
usage (GPU box, repo root): python3 tools/dbg_train_fp64.py frame naux coord main_only B L"""
import copy
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
import torch  # noqa: E402
from fixtures_util import initial_coords, synthetic_node_feats  # noqa: E402
from gpu_util import DEV, graph_tensors, model_pair  # noqa: E402

frame, naux, coord, main_only, B, L = [int(v) for v in sys.argv[1:7]]
hip, ref = model_pair(frame, naux, L, coord=bool(coord), main_only=bool(main_only), seed=frame + B)
for m in list(hip.modules()) + list(ref.modules()):
    if isinstance(m, torch.nn.Dropout):
        m.p = 0.0
topo, ei, nt, bi = graph_tensors(frame, naux, B, coord=bool(coord), main_only=bool(main_only))
feats = synthetic_node_feats(B * topo.num_nodes, 128, seed=5)
c0 = initial_coords(B, frame) if coord else None
ref64 = copy.deepcopy(ref).double()


def run(model, x, e, c):
    model.train()
    want, wc = model.forward_nodes(x, e, nt, B, c) if model is not hip else model.forward_nodes(x, e, B, c)
    ((want ** 2).mean() + (0 if wc is None else (wc ** 2).mean() * 1e-3)).backward()
    return want.detach().double().cpu(), {k: p.grad.double().cpu().clone() for k, p in model.named_parameters()}


o64, g64 = run(ref64, feats.double(), ei, None if c0 is None else c0.clone().double())
o32, g32 = run(ref, feats, ei, None if c0 is None else c0.clone())
oh, gh = run(hip, feats.to(DEV), ei.to(DEV), None if c0 is None else c0.clone().to(DEV))
print(f"logits: |ref32 - fp64| {float((o32 - o64).abs().max()):.2e}   |hip - fp64| {float((oh - o64).abs().max()):.2e}")
print(f"{'parameter':44s} {'ref32':>10s} {'hip':>10s}   (max abs error / max |grad fp64|)   channels > 1e-3: ref32 / hip")
for k in g64:
    sc = float(g64[k].abs().max())
    if sc < 1e-12:
        continue
    e32 = ((g32[k] - g64[k]).abs() / sc).reshape(g64[k].shape[0], -1).max(1).values
    eh = ((gh[k] - g64[k]).abs() / sc).reshape(g64[k].shape[0], -1).max(1).values
    if max(float(e32.max()), float(eh.max())) > 2e-4:
        print(f"{k:44s} {float(e32.max()):10.2e} {float(eh.max()):10.2e}   {int((e32 > 1e-3).sum())} / {int((eh > 1e-3).sum())} of {eh.numel()}")
for k in ("gnn_layers.%d.module_0.lin.weight" % (L - 1), "gnn_layers.0.module_0.lin.weight"):
    sc = float(g64[k].abs().max())
    for tag, g in (("ref32", g32), ("hip", gh)):
        e = ((g[k] - g64[k]).abs() / sc).reshape(g64[k].shape[0], -1).max(1).values
        srt = torch.sort(e, descending=True).values
        print(f"{k} [{tag}] per-channel max error: top {[f'{v:.1e}' for v in srt[:6].tolist()]} median {float(e.median()):.1e} min {float(e.min()):.1e}")
    # and per INPUT channel (columns)
    for tag, g in (("ref32", g32), ("hip", gh)):
        e = ((g[k] - g64[k]).abs() / sc).max(0).values
        srt = torch.sort(e, descending=True).values
        print(f"{k} [{tag}] per-input-channel max error: top {[f'{v:.1e}' for v in srt[:6].tolist()]} median {float(e.median()):.1e}")
