"""Per-kernel timing of one eager step with HIP events (diagnostic)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
import torch, time
from echoglad_amd import ops, nn as egnn
from echoglad_amd.topology import TopologySpec, get_topology
from echoglad_amd.synthetic import fill_state_dict, synthetic_node_feats
B = 8
kw = dict(frame_size=224, gnn_dropout_p=0.5, classifier_dropout_p=0.5, node_embedding_dim=128, node_hidden_dim=128,
          num_output_channels=4, num_gnn_layers=3, num_aux_graphs=7, classifier_hidden_dim=32, output_activation="logit")
m = egnn.HierarchicalPatchModel(**kw); fill_state_dict(m, 200); m = m.cuda().eval()
topo = get_topology(TopologySpec(224, 7)); N = topo.num_nodes
x = synthetic_node_feats(B * N, 128, 200).cuda()
ei = torch.from_numpy(topo.batched_edge_index(B)).cuda()
g, gb = m._resolver.resolve(ei, x.shape[0])
folded = m._folded_layers(); packed = m._packed_classifier()
bufs = [torch.empty_like(x) for _ in range(3)]
def step(ev=None):
    h = x
    for i in range(3):
        w, sc, sh = folded[i]
        if ev: ev[i].record()
        h = ops.gcn_layer_fwd(g, gb, h, w, sc, sh, h, relu=(i < 2), out=bufs[i])
    if ev: ev[3].record()
    out = ops.classifier_fwd(h, B, N, 0, N, packed)
    if ev: ev[4].record()
    return out
for _ in range(5): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(30): step()
torch.cuda.synchronize()
print("direct ops loop: %.3f ms/step" % ((time.perf_counter() - t0) / 30 * 1e3))
ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
acc = [0.0] * 4
for _ in range(10):
    step(ev); torch.cuda.synchronize()
    for i in range(4): acc[i] += ev[i].elapsed_time(ev[i + 1])
print("per-kernel ms (layer0, layer1, layer2, classifier):", [round(a / 10, 3) for a in acc])
with torch.no_grad():
    for _ in range(3): m.forward_nodes(x, ei, B)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30): m.forward_nodes(x, ei, B)
    torch.cuda.synchronize(); print("model.forward_nodes eager: %.3f ms/step" % ((time.perf_counter() - t0) / 30 * 1e3))
    t0 = time.perf_counter()
    for _ in range(30): m.forward_nodes(x, ei, B)
    print("  host-side enqueue time per step: %.3f ms" % ((time.perf_counter() - t0) / 30 * 1e3)); torch.cuda.synchronize()
    m.enable_hip_graph(True)
    for _ in range(3): m.forward_nodes(x, ei, B)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30): m.forward_nodes(x, ei, B)
    torch.cuda.synchronize(); print("model.forward_nodes hipGraph: %.3f ms/step" % ((time.perf_counter() - t0) / 30 * 1e3))
