"""INTEGRATION.md section E's loop taken apart: copy_batch_ alone, the graph replay alone, both (bench.documented_graphed_loop_ms).
    python tools/tools_docloop.py [B]        (under rocprofv3 --kernel-trace --stats for the per-kernel table of the replay)"""
import copy
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from echoglad_amd import data, engine, losses  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev = torch.device("cuda", 0)
frame, naux, C = 224, 7, 128
model = bench.build_model(bench.model_kwargs(frame, naux, 3, coord=True), dev, train=True)
emb = torch.nn.Conv2d(1, C, kernel_size=1).to(dev)
ds = data.SyntheticEchoDataset(num_aux_graphs=naux, frame_size=frame, use_coordinate_graph=True)
host = [data.collate([ds[i * B + j] for j in range(B)], ds.topology) for i in range(4)]
static = data.to_device(copy.copy(host[0]), dev)
crit = {"bce": losses.WeightedBCEWithLogitsLoss("none", 9000, 1), "elm": losses.ExpectedLandmarkMSE(10, B, frame, naux), "coordinate": engine.MSE(1)}
emb_grad = os.environ.get("EMB_GRAD", "1") != "0"
for q in emb.parameters():
    q.requires_grad_(emb_grad)
params = list(model.parameters()) + (list(emb.parameters()) if emb_grad else [])
opt = torch.optim.Adam(params, lr=1e-4, fused=True, capturable=True)
md = {"embedder": emb, "landmark": model}
coords0 = static.node_coords.clone()


def loss_fn():
    static.node_coords = coords0.clone()
    preds, cp = engine.forward_batch(md, static, True)
    return engine.total_loss(engine.compute_loss(crit, preds, static.y, cp, static.node_coord_y, static.valid_labels, B))


step = engine.GraphedTrainStep(loss_fn, opt, warmup=2)


def timed(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(n):
        fn(k)
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


print("emb requires grad:", emb_grad, " B =", B)
print("copy_batch_ alone  ms:", round(timed(lambda k=0: data.copy_batch_(static, host[k % 4])), 3))
print("graph replay alone ms:", round(timed(lambda k=0: step()), 3))
print("both               ms:", round(timed(lambda k=0: (data.copy_batch_(static, host[k % 4]), step())), 3))
for name in ("x", "y", "valid_labels", "node_coords", "node_coord_y", "pix2mm_x"):
    v = getattr(host[0], name)
    print(f"  {name}: {tuple(v.shape)} {v.dtype} {v.numel() * v.element_size() / 1e6:.3f} MB")
