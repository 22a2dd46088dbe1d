"""Diagnostic: where the HOST spends a training step (bench.py's configs[3] step, no synchronisation inside): wall time of
each phase as the host sees it, then the step time with a synchronise at the end.  A phase that takes about as long as the GPU
needs for everything queued before it is a hidden host synchronisation.
usage (GPU box, repo root): python3 tools/tools_hosttime.py [batch]"""
import os
import sys
import time

R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, R)
import torch  # noqa: E402

import bench  # noqa: E402
from echoglad_amd import engine  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda:0")
step, topo = bench.train_workload(224, 7, 3, B, dev, 1, 0)
cl = {c.cell_contents.__class__.__name__: c.cell_contents for c in step.__closure__ if hasattr(c, "cell_contents")}
names = step.__code__.co_freevars
env = dict(zip(names, (c.cell_contents for c in step.__closure__)))
model, feats, edge_index, coords0, crit, y, coord_y, valid, opt = (env[k] for k in
    ("model", "feats", "edge_index", "coords0", "crit", "y", "coord_y", "valid", "opt"))
for _ in range(3):
    step()
torch.cuda.synchronize()
acc = {}
N = 10
t_all = time.perf_counter()
for _ in range(N):
    t = [time.perf_counter()]
    preds, cp = model.forward_nodes(feats, edge_index, B, coords0.clone()); t.append(time.perf_counter())
    ls = engine.compute_loss(crit, preds, y, cp, coord_y, valid, B); loss = sum(ls.values()); t.append(time.perf_counter())
    opt.zero_grad(set_to_none=True); loss.backward(); t.append(time.perf_counter())
    opt.step(); t.append(time.perf_counter())
    for k, a, b in zip(("forward", "losses", "backward", "optimizer"), t[:-1], t[1:]):
        acc[k] = acc.get(k, 0.0) + (b - a)
host = time.perf_counter() - t_all
torch.cuda.synchronize()
total = time.perf_counter() - t_all
print(f"host wall per step (no sync inside): {1e3 * host / N:.3f} ms; with the final synchronise: {1e3 * total / N:.3f} ms")
for k, v in acc.items():
    print(f"  {k:10s} {1e3 * v / N:7.3f} ms")
