"""Diagnostic: per-step GPU time of hipGraph replays (where do one-off stalls land?)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
import torch
from echoglad_amd import nn as egnn
from echoglad_amd.topology import TopologySpec, get_topology
from echoglad_amd.synthetic import fill_state_dict, synthetic_node_feats
B = 8
kw = dict(frame_size=224, gnn_dropout_p=0.5, classifier_dropout_p=0.5, node_embedding_dim=128, node_hidden_dim=128,
          num_output_channels=4, num_gnn_layers=3, num_aux_graphs=7, classifier_hidden_dim=32, output_activation="logit")
m = egnn.HierarchicalPatchModel(**kw); fill_state_dict(m, 200); m = m.cuda().eval(); m.enable_hip_graph(True)
topo = get_topology(TopologySpec(224, 7)); N = topo.num_nodes
x = synthetic_node_feats(B * N, 128, 200).cuda()
ei = torch.from_numpy(topo.batched_edge_index(B)).cuda()
n = 300
with torch.no_grad():
    for _ in range(3): m.forward_nodes(x, ei, B)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    host = []
    t0 = time.perf_counter()
    ev[0].record()
    for i in range(n):
        h0 = time.perf_counter()
        m.forward_nodes(x, ei, B)
        ev[i + 1].record()
        host.append((time.perf_counter() - h0) * 1e3)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) * 1e3
gpu = [ev[i].elapsed_time(ev[i + 1]) for i in range(n)]
import statistics
print(f"wall {wall:.1f} ms for {n} steps = {wall/n:.3f} ms/step; gpu median {statistics.median(gpu):.3f} max {max(gpu):.3f} at step {gpu.index(max(gpu))}")
print("gpu steps > 1.5x median:", [(i, round(g, 2)) for i, g in enumerate(gpu) if g > 1.5 * statistics.median(gpu)][:20])
print(f"host enqueue median {statistics.median(host):.3f} ms max {max(host):.3f} at {host.index(max(host))}; host > 1 ms:", [(i, round(h, 2)) for i, h in enumerate(host) if h > 1.0][:20])
