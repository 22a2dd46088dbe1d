"""Cost of the queue-ring guard (an event recorded behind every launch): eager chained layer launches, B = 8, configs[1]."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from echoglad_amd import ops
DEV = "cuda:0"
B = 8
g = ops.Graph.topo(224, 7)
x = torch.randn(B * g.num_nodes, 128, device=DEV)
w = torch.randn(128, 128, device=DEV) * 0.08
ka, kb = ops.new_kidsum(g, B), ops.new_kidsum(g, B)
out = torch.empty_like(x)
def run():
    ops.gcn_layer_fwd(g, B, x, w, None, None, x, relu=True, out=out, kidsum_in=ka, kidsum_out=kb)
for _ in range(20): run()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
best = 1e9
for rep in range(5):
    torch.cuda.synchronize(); e0.record()
    for _ in range(100): run()
    e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) / 100)
print("EG_RING_GUARD=%s  chained layer launch, eager: %.4f ms" % (os.environ.get("EG_RING_GUARD", "1"), best))
