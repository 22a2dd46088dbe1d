"""Debug: parameter-gradient error of the train step on a connection-node graph -- HIP stencil handle vs HIP CSR handle vs the CPU
oracle in fp32, all against the oracle in fp64."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
import copy
import torch
from fixtures_util import initial_coords, synthetic_node_feats
from gpu_util import DEV, graph_tensors, model_pair
from echoglad_amd import topology

frame, naux = 64, 5
coord, conn, B = [int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else ("1", "1", "2"))]
print("== coord", coord, "conn", conn, "B", B)
hip, ref = model_pair(frame, naux, 3, coord=bool(coord), seed=frame + 2, use_connection_nodes=bool(conn))
for m in list(hip.modules()) + list(ref.modules()):
    if isinstance(m, torch.nn.Dropout):
        m.p = 0.0
topo, ei, nt, bi = graph_tensors(frame, naux, B, coord=bool(coord), conn=bool(conn))
feats = synthetic_node_feats(B * topo.num_nodes, 128, seed=9)
c0 = initial_coords(B, frame) if coord else None
ref64 = copy.deepcopy(ref).double()
state = {k: v.clone() for k, v in hip.state_dict().items()}


def run_ref(model, dt):
    model.train()
    want, wc = model.forward_nodes(feats.to(dt), ei, nt, B, None if c0 is None else c0.clone().to(dt))
    ((want ** 2).mean() + (0 if wc is None else (wc ** 2).mean() * 1e-3)).backward()
    return {k: p.grad.double().clone() for k, p in model.named_parameters()}


def run_hip(force_csr):
    orig = topology.HierTopology.is_structured
    if force_csr:
        topology.HierTopology.is_structured = lambda self: False
    try:
        hip.load_state_dict(state)
        hip._resolver = type(hip._resolver)(hip.topology_spec)
        for p in hip.parameters():
            p.grad = None
        hip.train()
        got, gc = hip.forward_nodes(feats.to(DEV), ei.to(DEV), B, None if c0 is None else c0.clone().to(DEV))
        g = hip._resolver.resolve(ei.to(DEV), feats.shape[0])[0]
        ((got ** 2).mean() + (0 if gc is None else (gc ** 2).mean() * 1e-3)).backward()
        return {k: p.grad.double().cpu().clone() for k, p in hip.named_parameters()}, g.structured
    finally:
        topology.HierTopology.is_structured = orig


g64 = run_ref(ref64, torch.float64)
g32 = run_ref(ref, torch.float32)
gs, s1 = run_hip(False)
gc_, s2 = run_hip(True)
print("structured:", s1, s2)
print(f"{'parameter':40s} {'ref32':>10s} {'stencil':>10s} {'csr':>10s}   (max abs error / max |grad fp64|)")
for k in g64:
    sc = float(g64[k].abs().max())
    if sc < 1e-12:
        continue
    e = [float((x[k] - g64[k]).abs().max()) / sc for x in (g32, gs, gc_)]
    if max(e) > 2e-4:
        print(f"{k:40s} {e[0]:10.2e} {e[1]:10.2e} {e[2]:10.2e}")
# error distribution of the worst parameter: per output channel (row of a weight / entry of a vector)
worst = max((k for k in g64 if float(g64[k].abs().max()) > 1e-12), key=lambda k: float((gs[k] - g64[k]).abs().max()) / float(g64[k].abs().max()))
sc = float(g64[worst].abs().max())
for tag, g in (("stencil", gs), ("csr", gc_), ("ref32", g32)):
    e = ((g[worst] - g64[worst]).abs() / sc).reshape(g64[worst].shape[0], -1).max(1).values
    srt = torch.sort(e, descending=True).values
    print(f"{worst} [{tag}] per-channel max error: top {[f'{v:.1e}' for v in srt[:5].tolist()]} median {float(e.median()):.1e}; channels > 1e-4: {int((e > 1e-4).sum())} of {e.numel()}")
