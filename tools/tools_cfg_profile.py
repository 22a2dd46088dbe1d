"""Diagnostic: run one of bench.py's side configurations for a number of steps (to be wrapped in rocprofv3 --kernel-trace --stats).
usage (GPU box, repo root): python3 tools/tools_cfg_profile.py conn|diag|plain [steps]"""
import os
import sys

R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, R)
import torch  # noqa: E402

import bench  # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else "conn"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
dev = torch.device("cuda:0")
kw = dict(conn=True) if what == "conn" else dict(graph_type="grid-diagonal") if what == "diag" else {}
model, _, topo, feats, ei, step = bench.infer_workload(224, 7, 3, False, 8, dev, 0, **kw)
for _ in range(5):
    step()
torch.cuda.synchronize()
ms = bench.time_steps(step, iters=steps, warm=2)
print(f"{what}: {ms:.4f} ms per step")
