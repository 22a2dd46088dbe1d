"""Diagnostic: per-phase cycle shares of the fused layer kernel (needs the -DEG_STAMP variant)."""
import ctypes as ct, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
import torch
from echoglad_amd import ops, _lib
from echoglad_amd.synthetic import synthetic_node_feats
B = int(os.environ.get("EG_STAMP_B", "8"))
g = ops.Graph.topo(224, 7)
x = synthetic_node_feats(B * g.num_nodes, 128, 1).cuda()
w = (synthetic_node_feats(128, 128, 2) * 0.1).cuda()
out = torch.empty_like(x)
buf = (ct.c_uint64 * 32)()
lib = _lib.load()
form = os.environ.get("EG_STAMP_FORM", "")          # "", "kin", "kout", "kin+kout", "kin+cls": chained forms run the producer/consumer kernel
kw = {}
if "kin" in form: kw["kidsum_in"] = ops.new_kidsum(g, B)
if "kout" in form: kw["kidsum_out"] = ops.new_kidsum(g, B)
packed = None
if "cls" in form:                                    # last layer + fused classifier heads (k_gcn_layer_ps<true>)
    f = lambda *shape: (synthetic_node_feats(int(torch.tensor(shape).prod()), 1, 7).reshape(shape) * 0.1).cuda().contiguous()
    packed = {"w1": f(128, 128), "s1": f(128) + 1, "t1": f(128), "w2": f(4, 16, 32), "s2": f(64) + 1, "t2": f(64), "w3": f(4, 16), "b3": f(4)}
import time
warm = float(os.environ.get("EG_STAMP_WARM", "0"))        # seconds of back-to-back launches first (DVFS steady state)
t_end = time.time() + warm
while time.time() < t_end:
    for _ in range(50):
        if packed is not None:
            ops.gcn_layer_cls_fwd(g, B, x, w, None, None, x, False, packed, kidsum_in=kw.get("kidsum_in"))
        else:
            ops.gcn_layer_fwd(g, B, x, w, None, None, x, relu=True, out=out, **kw)
    torch.cuda.synchronize()
lib.eg_debug_phase_cycles(g._h, buf, 1)
for it in range(3):
    if packed is not None:
        ops.gcn_layer_cls_fwd(g, B, x, w, None, None, x, False, packed, kidsum_in=kw.get("kidsum_in"))
    else:
        ops.gcn_layer_fwd(g, B, x, w, None, None, x, relu=True, out=out, **kw)
    lib.eg_debug_phase_cycles(g._h, buf, 1)
    v = list(buf)
names = ["c_mfma", "c_epilogue", "c_barrier", "c_loop", "p_issue", "p_main(wait+fma)", "p_kids+store", "p_claim+barrier"] if (os.environ.get("EG_LAYER_IMPL", "0") != "0" or form) else ["next_tile", "phase1", "barrier1", "mfma", "barrier2", "dwrite", "barrier3", "phase3"]
tot = sum(v[:8]); waves = max(v[8], 1)
tiles = 1128 * B
print(f"waves(counted)={waves} tiles={tiles}")
if v[10]:
    print(f"in-kernel clock = {v[9] / v[10] * 0.1:.3f} GHz  (shader cycles / 100-MHz ticks around the tile loop, stamp build)")
    n_wg = min(256, tiles)
    print(f"tile loop per workgroup: {v[10] / n_wg * 0.01:.1f} us mean ({v[9] / n_wg:.0f} cycles; {tiles / n_wg:.2f} tiles per workgroup, "
          f"{v[9] / max(tiles, 1):.0f} cycles per tile incl. pipeline fill / drain)")
wgs = 256 if (os.environ.get("EG_LAYER_IMPL", "0") != "0" or form) else waves / 8
waves = wgs * 4 if (os.environ.get("EG_LAYER_IMPL", "0") != "0" or form) else waves
for n, c in zip(names, v[:8]):
    print(f"{n:10s} {100*c/tot:6.2f}%  {c/waves/(tiles/wgs):9.0f} cyc/tile")

if len(v) >= 32 and form:
    print("per wave: cycles per tile waiting at the barrier / in all (waves 0-3 consumers, 4-7 producers)")
    for w in range(8):
        print(f"  wave {w}: {v[16 + w] / 256 / (tiles / 256):8.0f} / {v[24 + w] / 256 / (tiles / 256):8.0f}")
