import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch
from gpu_util import DEV, graph_tensors, model_pair
from fixtures_util import initial_coords, synthetic_frames
frame, naux, coord, B, L = 32, 4, True, 2, 3
topo, ei, nt, bi = graph_tensors(frame, naux, B, coord=coord)
frames = synthetic_frames(B, 128, frame, 11).to(DEV)
coords0 = initial_coords(B, frame).to(DEV)


def run(knob, noise=0.0, feats_from=None):
    os.environ["EG_POOL_PYRAMID"] = knob
    hip, _ = model_pair(frame, naux, L, coord=coord, seed=31)
    for m in hip.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    hip.train()
    feats = hip.create_node_pixels(frames, B, coords0.reshape(B, 4, 2)) if feats_from is None else feats_from.clone()
    if noise:
        torch.manual_seed(1)
        feats = feats * (1 + noise * torch.randn_like(feats))
    got, gc = hip.forward_nodes(feats, ei.to(DEV), B, coords0.clone())
    ((got ** 2).mean() + (gc ** 2).mean() * 1e-3).backward()
    return feats.detach(), got.detach(), {k: p.grad.clone() for k, p in hip.named_parameters()}


f1, o1, g1 = run("1")
f0, o0, g0 = run("0")
_, o2, g2 = run("0", noise=1e-7)
_, o3, g3 = run("0", feats_from=f0)
print("logits: pyramid vs torch", float((o1 - o0).abs().max()), " torch+1e-7 noise vs torch", float((o2 - o0).abs().max()), " torch again", float((o3 - o0).abs().max()))
for k in ("gnn_layers.0.module_0.lin.weight", "gnn_layers.2.module_0.lin.weight", "node_classifiers.0.0.weight"):
    s = float(g0[k].abs().max())
    print(k, "pyramid vs torch %.2e   noise vs torch %.2e   torch again %.2e" % (float((g1[k] - g0[k]).abs().max()) / s, float((g2[k] - g0[k]).abs().max()) / s,
                                                                                  float((g3[k] - g0[k]).abs().max()) / s))
_, o4, g4 = run("0", feats_from=f1)          # the pyramid route's VALUES through the torch-pool route's code
_, o5, g5 = run("1", feats_from=f0)
k = "node_classifiers.0.0.weight"; s = float(g0[k].abs().max())
print("values of pyramid, route torch: %.2e ; values of torch, route pyramid: %.2e" % (float((g4[k] - g0[k]).abs().max()) / s, float((g5[k] - g0[k]).abs().max()) / s))
d = (f1 - f0).abs(); print("feature diff max", float(d.max()), "nonzero rows", int((d.max(dim=1).values > 0).sum()))
# one ulp on the same rows, other direction
f6 = f0.clone(); rows = (d.max(dim=1).values > 0)
f6[rows] = torch.nextafter(f0[rows], torch.full_like(f0[rows], 1e9))
_, o6, g6 = run("0", feats_from=f6)
print("one ulp up on those rows: %.2e" % (float((g6[k] - g0[k]).abs().max()) / s))
n = f0.shape[0] // B
for name, lo, hi in (("level 2x2", 0, 4), ("level 4x4", 4, 20), ("level 8x8", 20, 84)):
    f7 = f0.clone().view(B, n, 128)
    f7[:, lo:hi] = f1.view(B, n, 128)[:, lo:hi]
    _, o7, g7 = run("0", feats_from=f7.view(B * n, 128))
    print(name, "rows replaced: grad diff %.2e  logits diff %.2e" % (float((g7[k] - g0[k]).abs().max()) / s, float((o7 - o0).abs().max())))
for j in range(20, 84, 8):
    f7 = f0.clone().view(B, n, 128)
    f7[:, j:j + 8] = f1.view(B, n, 128)[:, j:j + 8]
    _, o7, g7 = run("0", feats_from=f7.view(B * n, 128))
    print("rows", j, j + 8, "grad diff %.2e" % (float((g7[k] - g0[k]).abs().max()) / s))
f8 = f0.clone().view(B, n, 128)
f8[:, 0:84] = f1.view(B, n, 128)[:, 0:84]
f8 = f8.view(B * n, 128)
print("f8 == f1 bitwise:", torch.equal(f8, f1), " int view equal:", torch.equal(f8.view(torch.int32), f1.view(torch.int32)), "strides", f1.stride(), f0.stride(), f1.is_contiguous())
_, o8, g8 = run("0", feats_from=f8)
print("all three levels replaced: grad diff %.2e" % (float((g8[k] - g0[k]).abs().max()) / s))
_, o9, g9 = run("0", feats_from=f1)
print("f1 again: grad diff %.2e" % (float((g9[k] - g0[k]).abs().max()) / s))
ne = (f8.view(torch.int32) != f1.view(torch.int32)).nonzero()
print("differing elements:", ne.shape[0], ne[:5].tolist(), [ (float(f8[i, j]), float(f1[i, j])) for i, j in ne[:5].tolist()])
print("---- per parameter: relative gradient difference, pyramid values vs torch values (same code route)")
for kk in g0:
    if g0[kk].dim() >= 1 and float(g0[kk].abs().max()) > 0:
        print(f"{kk:45s} {float((g9[kk] - g0[kk]).abs().max()) / float(g0[kk].abs().max()):.2e}")


def coords_of(feats):
    os.environ["EG_POOL_PYRAMID"] = "0"
    hip, _ = model_pair(frame, naux, L, coord=coord, seed=31)
    for m in hip.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    hip.train()
    seen = []
    got, gc = hip.forward_nodes(feats.clone(), ei.to(DEV), B, coords0.clone())
    return gc.detach()


c0, c1 = coords_of(f0), coords_of(f1)
print("final coords f0:", c0.flatten().tolist())
print("final coords f1:", c1.flatten().tolist())
