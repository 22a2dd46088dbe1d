"""Diagnostic for tests/test_gpu_engine.py::test_graphed_train_step_...: where do the graph's and the eager run's states part?"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (R, os.path.join(R, "tests"), os.path.join(R, "tests", "golden")):
    sys.path.insert(0, p)
import numpy as np, torch
import test_gpu_engine as T
from gpu_util import DEV
from echoglad_amd import data, engine, losses, ops

frame, naux, coord, p, B, warm = 16, 3, False, 0.5, 2, int(os.environ.get("WARM", 2))


def build():
    hip, _, emb_hip, _, ds = T._setup(frame, naux, coord, 11)
    for m in hip.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = p
    hip.train()
    batch = data.to_device(data.collate([ds[i] for i in range(B)], ds.topology), DEV)
    crit = {"bce": losses.WeightedBCEWithLogitsLoss("none", 9000, 1), "elm": losses.ExpectedLandmarkMSE(10, B, frame, naux)}
    params = list(hip.parameters()) + list(emb_hip.parameters())
    opt = torch.optim.Adam(params, lr=1e-3, capturable=True)
    model = {"embedder": emb_hip, "landmark": hip}

    def loss_fn():
        preds, coord_preds = engine.forward_batch(model, batch, coord)
        ls = engine.compute_loss(crit, preds, batch.y, coord_preds, None, batch.valid_labels, B)
        return sum(ls.values()), preds
    return hip, emb_hip, opt, loss_fn, params


def run(side_stream):
    ops.dropout_epoch_set(0)
    torch.manual_seed(123)
    hip, emb, opt, loss_fn, params = build()
    torch.manual_seed(77)
    snaps = []

    def eager():
        out = loss_fn()
        opt.zero_grad(set_to_none=True)
        out[0].backward()
        opt.step()
        return out
    if side_stream:
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(warm):
                out = eager()
                snaps.append([q.detach().clone() for q in params] + [out[1].detach().clone()] + [q.grad.clone() for q in params])
        torch.cuda.current_stream().wait_stream(s)
    else:
        for _ in range(warm):
            out = eager()
            snaps.append([q.detach().clone() for q in params] + [out[1].detach().clone()] + [q.grad.clone() for q in params])
    torch.cuda.synchronize()
    return snaps, [n for n, _ in hip.named_parameters()] + ["emb.w", "emb.b"]


a, names = run(False)
b, _ = run(False)
c, _ = run(True)
np_ = len(names)
for label, x, y in (("default vs default", a, b), ("default vs side stream", a, c)):
    for k in range(warm):
        bad = [(names[i] if i < np_ else ("preds" if i == np_ else "grad " + names[i - np_ - 1]), float((u - v).abs().max())) for i, (u, v) in enumerate(zip(x[k], y[k])) if not torch.equal(u, v)]
        print(label, "step", k, "differing:", bad[:8], "(", len(bad), ")")
