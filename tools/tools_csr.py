"""Diagnostic: launch time of the layer kernel on a CSR handle of configs[1]'s batch-8 graph, by EG_CSR_TILES mode
(0 row by row, 1 LDS stash over consecutive rows, 2 clustered tiles), next to the stencil handle's symmetric kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from echoglad_amd import ops
from echoglad_amd.synthetic import synthetic_node_feats
from echoglad_amd.topology import TopologySpec, get_topology
B = int(os.environ.get("EG_B", "8"))
topo = get_topology(TopologySpec(224, 7, False, False))
ei = torch.from_numpy(topo.batched_edge_index(B)).cuda()
n = B * topo.num_nodes
x = synthetic_node_feats(n, 128, 1).cuda()
w = (synthetic_node_feats(128, 128, 2) * 0.1).cuda()
out = torch.empty_like(x)


def timeit(g, batch, label):
    for _ in range(5):
        ops.gcn_layer_fwd(g, batch, x, w, None, None, x, relu=True, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30):
        ops.gcn_layer_fwd(g, batch, x, w, None, None, x, relu=True, out=out)
    e1.record(); torch.cuda.synchronize()
    print(f"{label:48s} {e0.elapsed_time(e1) / 30:.4f} ms")


for mode in os.environ.get("EG_MODES", "0,1,2").split(","):
    os.environ["EG_CSR_TILES"] = mode
    t0 = __import__("time").perf_counter()
    g = ops.Graph.csr(ei, n)
    torch.cuda.synchronize()
    print(f"  (eg_csr_create with EG_CSR_TILES={mode}: {__import__('time').perf_counter() - t0:.3f} s)")
    timeit(g, 1, f"CSR handle, EG_CSR_TILES={mode}")
timeit(ops.Graph.topo(224, 7), B, "stencil handle, plain call (symmetric kernel)")
