"""Diagnostic: run one of bench.py's side configurations for a number of steps (to be wrapped in rocprofv3 --kernel-trace --stats).
usage (GPU box, repo root): python3 tools/tools_cfg_profile.py conn|diag|plain|pyg|csr [steps]
  pyg: INTEGRATION route B -- the reference-shaped loop over Sequential / GCNConv + torch residual adds, row filter and heads
  csr: configs[1] through a CSR handle (3 x eg_gcn_layer_fwd + eg_classifier_fwd)"""
import os
import sys

R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, R)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else "conn"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
dev = torch.device("cuda:0")
if what in ("pyg", "csr"):
    from echoglad_amd import ops
    model, _, topo, feats, ei, _ = bench.infer_workload(224, 7, 3, False, 8, dev, 0, hip_graph=False)
    if what == "pyg":
        keep = torch.nonzero(torch.from_numpy(np.tile(topo.node_type(), 8)).to(dev) == 0).squeeze(1)

        def body():
            hidden = [feats]
            for i in range(3):
                h = model.gnn_layers[i](hidden[i], ei)            # models.py:431
                h = h + hidden[i]                                 # :434-435
                hidden.append(h)
            h = hidden[-1][keep]                                  # :485
            return torch.cat([clf(h) for clf in model.node_classifiers], dim=1)      # :488-490
    else:
        g = ops.Graph.csr(ei, feats.shape[0])
        folded, packed = model._folded_layers(), model._packed_classifier()

        def body():
            h = feats
            for i, (w, sc, sh) in enumerate(folded):
                h = ops.gcn_layer_fwd(g, 1, h, w, sc, sh, h, relu=i < 2)
            return ops.classifier_fwd(h, 8, topo.num_nodes, 0, topo.num_nodes, packed)

    def step():
        with torch.no_grad():
            return body()
else:
    kw = dict(conn=True) if what == "conn" else dict(graph_type="grid-diagonal") if what == "diag" else {}
    model, _, topo, feats, ei, step = bench.infer_workload(224, 7, 3, False, 8, dev, 0, **kw)
for _ in range(5):
    step()
torch.cuda.synchronize()
ms = bench.time_steps(step, iters=steps, warm=2)
print(f"{what}: {ms:.4f} ms per step")
