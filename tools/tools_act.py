"""Timing of the activation pass of the train forward: flat streaming kernel vs the tile-order kernel (with / without child sums),
and of the train-forward layer launch with / without kidsum_in.  224/7 + coordinate nodes, batch 32 (BASELINE configs[3])."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from echoglad_amd import ops

DEV = "cuda:0"
B = int(os.environ.get("B", "32"))
g = ops.Graph.topo(224, 7, False, True)
rows = B * g.num_nodes
torch.manual_seed(0)
z = torch.randn(rows, 128, device=DEV)
x = torch.randn(rows, 128, device=DEV)
sc = torch.rand(128, device=DEV) + 0.5
sh = torch.randn(128, device=DEV) * 0.1
ka, kb = ops.new_kidsum(g, B), ops.new_kidsum(g, B)


def t(fn, it=20, warm=3):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(it):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it


tag = " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("EG_") or k == "ECHOGLAD_LIB")
print("==", tag or "default", flush=True)
print("flat            %.4f ms" % t(lambda: ops.bn_act_fwd(z, sc, sh, x, True, 0.5, 7)))
print("tiles           %.4f ms" % t(lambda: ops.bn_act_fwd_tiles(g, B, z, sc, sh, x, True, 0.5, 7)))
print("tiles + kidsum  %.4f ms" % t(lambda: ops.bn_act_fwd_tiles(g, B, z, sc, sh, x, True, 0.5, 7, kidsum_out=ka)))
a = ops.bn_act_fwd(z, sc, sh, x, True, 0.5, 7)
b = ops.bn_act_fwd_tiles(g, B, z, sc, sh, x, True, 0.5, 7, kidsum_out=ka)
print("equal:", bool(torch.equal(a, b)))
W = torch.randn(128, 128, device=DEV) * 0.08
one, zero = torch.ones(128, device=DEV), torch.zeros(128, device=DEV)
args = (W, zero, one, zero, None, None, None, 1e-5, True, 0.5, 3, True)
print("train fwd (layer+act), child rows   %.4f ms" % t(lambda: ops.gcn_layer_train_fwd(g, B, a, *args), it=10))
print("train fwd, kidsum_in                %.4f ms" % t(lambda: ops.gcn_layer_train_fwd(g, B, a, *args, kidsum_in=ka), it=10))
print("train fwd, kidsum_in + kidsum_out   %.4f ms" % t(lambda: ops.gcn_layer_train_fwd(g, B, a, *args, kidsum_in=ka, kidsum_out=kb), it=10))
