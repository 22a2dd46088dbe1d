"""Diagnostic: is one train-mode forward of a small model bit-reproducible (a) eagerly, (b) eager vs captured?"""
import os, sys, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
import numpy as np, torch
from gpu_util import DEV, model_pair
from echoglad_amd import data, engine, losses, ops

frame, naux, coord, p, B = int(os.environ.get("F", 16)), int(os.environ.get("NA", 3)), False, 0.5, 2
hip, _ = model_pair(frame, naux, 2, coord=coord, seed=11)
for m in hip.modules():
    if isinstance(m, torch.nn.Dropout):
        m.p = p
hip.train()
torch.manual_seed(11)
emb = torch.nn.Conv2d(1, 128, kernel_size=1).to(DEV)
np.random.seed(11)
ds = data.SyntheticEchoDataset(num_aux_graphs=naux, frame_size=frame, use_coordinate_graph=coord)
batch = data.to_device(data.collate([ds[i] for i in range(B)], ds.topology), DEV)
model = {"embedder": emb, "landmark": hip}
state = copy.deepcopy(hip.state_dict())


def fwd(with_emb=True):
    hip.load_state_dict(state)
    torch.manual_seed(5)
    if with_emb:
        return engine.forward_batch(model, batch, coord)[0].detach().clone()
    x = emb(batch.x).detach()
    return hip(x=x, node_coords=None, edge_index=batch.edge_index, batch_idx=batch.batch, node_type=batch.node_type)[0].detach().clone()


with torch.no_grad():
    xs = [emb(batch.x).clone() for _ in range(3)]
print("embedder eager repeat equal:", torch.equal(xs[0], xs[1]), torch.equal(xs[0], xs[2]))
outs = [fwd() for _ in range(4)]
print("eager forward repeat equal:", [torch.equal(outs[0], o) for o in outs[1:]], "max diff", max(float((outs[0] - o).abs().max()) for o in outs[1:]))
# captured
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2):
        fwd()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
hip.load_state_dict(state)
torch.manual_seed(5)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=s):
    with torch.no_grad():
        xg = emb(batch.x)
    pg = hip(x=xg, node_coords=None, edge_index=batch.edge_index, batch_idx=batch.batch, node_type=batch.node_type)[0]
hip.load_state_dict(state)      # (running stats moved by nothing yet: the capture executed nothing)
g.replay()
torch.cuda.synchronize()
print("embedder graph vs eager equal:", torch.equal(xg, xs[0]), float((xg - xs[0]).abs().max()))
print("forward graph vs eager equal:", torch.equal(pg.detach(), outs[0]), float((pg.detach() - outs[0]).abs().max()))
