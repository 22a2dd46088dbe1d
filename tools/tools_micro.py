"""Micro-timings on the GPU box: aggregation-only kernels vs a plain copy of the same bytes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
import torch
from echoglad_amd import ops
from echoglad_amd.topology import TopologySpec, get_topology
from echoglad_amd.synthetic import synthetic_node_feats

def timeit(fn, iters=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters

for B in (8, 2):
    g = ops.Graph.topo(224, 7)
    topo = get_topology(TopologySpec(224, 7))
    x = synthetic_node_feats(B * g.num_nodes, 128, 1).cuda()
    w = (synthetic_node_feats(128, 128, 2) * 0.1).cuda()
    out = torch.empty_like(x)
    mb = x.numel() * 4 / 1e6
    t = timeit(lambda: out.copy_(x)); print(f"B={B} copy {mb:.0f} MB: {t*1e3:.0f} us  ({2*mb/t/1e3:.2f} TB/s r+w)")
    t = timeit(lambda: ops.gcn_aggregate(g, B, x)); print(f"B={B} aggregate stencil: {t*1e3:.0f} us")
    ei = torch.from_numpy(topo.batched_edge_index(B)).cuda()
    gc = ops.Graph.csr(ei, B * g.num_nodes)
    t = timeit(lambda: ops.gcn_aggregate(gc, 1, x)); print(f"B={B} aggregate csr: {t*1e3:.0f} us")
    t = timeit(lambda: ops.linear128_fwd(x, w, None, None, x, relu=True)); print(f"B={B} linear128 (+residual): {t*1e3:.0f} us")
    t = timeit(lambda: ops.linear128_fwd(x, w, None, None, None, relu=True)); print(f"B={B} linear128: {t*1e3:.0f} us")
    t = timeit(lambda: ops.gcn_layer_fwd(g, B, x, w, None, None, x, relu=True, out=out)); print(f"B={B} fused layer stencil: {t*1e3:.0f} us")
    t = timeit(lambda: ops.gcn_layer_fwd(gc, 1, x, w, None, None, x, relu=True, out=out)); print(f"B={B} fused layer csr: {t*1e3:.0f} us")
