"""Diagnostic: time the producer/consumer layer kernel in its four chaining forms (cfg2, B=8)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
import torch
from echoglad_amd import ops
from echoglad_amd.synthetic import synthetic_node_feats
B = int(os.environ.get("EG_B", "8"))
g = ops.Graph.topo(224, 7)
x = synthetic_node_feats(B * g.num_nodes, 128, 1).cuda()
w = (synthetic_node_feats(128, 128, 2) * 0.1).cuda()
out = torch.empty_like(x)
ka, kb = ops.new_kidsum(g, B), ops.new_kidsum(g, B)
forms = {"plain (default kernel)": {}, "kout": dict(kidsum_out=ka), "kin+kout": dict(kidsum_in=ka, kidsum_out=kb), "kin": dict(kidsum_in=kb)}
for name, kw in forms.items():
    for _ in range(5):
        ops.gcn_layer_fwd(g, B, x, w, None, None, x, relu=True, out=out, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        ops.gcn_layer_fwd(g, B, x, w, None, None, x, relu=True, out=out, **kw)
    e1.record(); torch.cuda.synchronize()
    print(f"{name:24s} {e0.elapsed_time(e1) / 50:.4f} ms")
