"""-m gpu parity of the losses on the logits and the landmark decode (csrc/heatmap.hip) against the golden
fixtures produced by the reference's own classes and against the CPU oracle at other sizes."""
import os

import numpy as np
import pytest
import torch

from gpu_util import DEV
from oracle import gnn_oracle as O
from oracle import loss_oracle as LO
from echoglad_amd import evaluators as EV
from echoglad_amd import losses, ops

pytestmark = pytest.mark.gpu

FIXTURES = ["decode_f16_a3.npz", "decode_f30_a3.npz"]


def _load(golden_dir, name):
    d = np.load(os.path.join(golden_dir, name))
    return d, int(d["batch"]), int(d["frame"]), int(d["naux"])


@pytest.mark.parametrize("name", FIXTURES)
def test_losses_and_gradients_match_reference_fixture(golden_dir, name):
    d, B, F, naux = _load(golden_dir, name)
    lg = torch.from_numpy(d["logits"]).to(DEV).requires_grad_(True)
    y, v = torch.from_numpy(d["labels"]).to(DEV), torch.from_numpy(d["valid"]).to(DEV)
    n = lg.shape[0] // B
    bce = losses.WeightedBCEWithLogitsLoss(reduction="none", ones_weight=9000, loss_weight=1)
    l = bce.compute(lg.view(B, n, 4), y.view(B, n, 4), v)
    g, = torch.autograd.grad(l, lg)
    assert abs(float(l.detach()) - float(d["bce"])) <= 2e-6 * abs(float(d["bce"]))          # fp64 sums vs the reference's fp32 sums
    assert np.allclose(g.cpu().numpy(), d["grad_bce"], rtol=2e-5, atol=1e-9)
    elm = losses.ExpectedLandmarkMSE(loss_weight=10, batch_size=B, frame_size=F, num_aux_graphs=naux)
    l = elm.compute(lg, y, v)
    g, = torch.autograd.grad(l, lg)
    assert abs(float(l.detach()) - float(d["elm"])) <= 1e-5 * abs(float(d["elm"]))
    assert np.allclose(g.cpu().numpy(), d["grad_elm"], rtol=2e-4, atol=2e-8)


@pytest.mark.parametrize("name", FIXTURES)
def test_evaluator_matches_reference_fixture(golden_dir, name):
    d, B, F, _ = _load(golden_dir, name)
    ev = EV.LandmarkExpectedCoordiantesEvaluator(None, B, F, use_coord_graph=False)
    for _ in range(2):
        ev.update(torch.from_numpy(d["logits"]).to(DEV), torch.from_numpy(d["labels"]).to(DEV),
                  torch.from_numpy(d["pix2mm_x"]), torch.from_numpy(d["pix2mm_y"]), torch.from_numpy(d["valid"]).to(DEV))
    last = ev.get_last()
    for k, want in zip(d["last_keys"], d["last_vals"]):
        assert abs(float(last[str(k)]) - float(want)) <= 2e-5 * max(1.0, abs(float(want))), k
    co = ev.get_predictions()["coordinates"]
    names = ["lvid_top", "lvid_bot", "lvpw", "ivs"]
    assert np.array_equal(torch.stack([co["gt_" + k] for k in names], 1).numpy().astype(np.int64), d["gt_coords"])
    assert np.allclose(torch.stack([co["pred_" + k] for k in names], 1).numpy(), d["pred_coords"], rtol=1e-6, atol=2e-5)
    for k, want in zip(d["width_keys"], d["width_vals"]):
        assert np.allclose(ev.get_predictions()["widths"][str(k)].numpy(), want, rtol=2e-5, atol=2e-5), k
    mean = ev.compute()                                   # two identical iterations: the mean equals the last record
    for k in last:
        assert abs(float(mean[k]) - float(last[k])) <= 1e-6 * max(1.0, abs(float(last[k])))
    # hard arg max: bit-exact indices, first of two equal maxima
    got = EV.decode_landmarks(torch.from_numpy(d["logits"]).to(DEV), B, F)["argmax"].cpu().numpy()
    assert np.array_equal(got, d["argmax_main"])


@pytest.mark.parametrize("frame,naux,batch", [(224, 7, 2), (64, 6, 3), (17, 3, 2)])
def test_decode_and_losses_vs_oracle_at_full_size(frame, naux, batch):
    """BASELINE-size frames (50,176-node main grid = 25 chunks per level): same numbers as the CPU oracle."""
    rs = np.random.RandomState(frame + naux)
    levels = LO.level_grids(frame, naux)
    n = levels[-1][0] + frame * frame
    logits = (rs.standard_normal((batch, n, 4)) * 3).astype(np.float32)
    y = np.zeros((batch, n, 4), np.float32)
    for b in range(batch):
        for c in range(4):
            hh, ww = rs.randint(0, frame, 2)
            for st, s in levels:
                y[b, st + (hh * s // frame) * s + (ww * s // frame), c] = 1.0
                logits[b, st + (hh * s // frame) * s + min(s - 1, (ww * s // frame) + 1), c] += 9.0
    valid = np.ones_like(y)
    valid[0, :, 3] = 0.0
    lg_c = torch.from_numpy(logits).view(-1, 4).requires_grad_(True)
    y_c, v_c = torch.from_numpy(y).view(-1, 4), torch.from_numpy(valid).view(-1, 4)
    want_elm = LO.expected_landmark_mse(lg_c, y_c, v_c, batch, frame, naux, loss_weight=10)
    want_bce = LO.weighted_bce_with_logits(lg_c.view(batch, n, 4), y_c.view(batch, n, 4), v_c, 9000, 1)
    gw, = torch.autograd.grad(want_elm + want_bce, lg_c)
    lg = lg_c.detach().to(DEV).requires_grad_(True)
    yd, vd = y_c.to(DEV), v_c.to(DEV)
    elm = losses.ExpectedLandmarkMSE(loss_weight=10, batch_size=batch, frame_size=frame, num_aux_graphs=naux)
    bce = losses.WeightedBCEWithLogitsLoss("none", 9000, 1)
    got_elm, got_bce = elm.compute(lg, yd, vd), bce.compute(lg.view(batch, n, 4), yd.view(batch, n, 4), vd)
    gg, = torch.autograd.grad(got_elm + got_bce, lg)
    assert abs(float(got_elm.detach()) - float(want_elm.detach())) <= 2e-5 * abs(float(want_elm.detach())) + 1e-7
    assert abs(float(got_bce.detach()) - float(want_bce.detach())) <= 2e-5 * abs(float(want_bce.detach()))
    assert float((gg.cpu() - gw).abs().max()) <= 2e-4 * float(gw.abs().max()) + 1e-9
    dec = EV.decode_landmarks(lg.detach(), batch, frame, yd, vd)
    assert torch.equal(dec["argmax"].cpu(), O.landmark_argmax(lg_c.detach(), batch, frame))
    # the kernel sums in fp64; the reference's fp32 sums over 50k nodes drift by several 1e-3 px, so the tight
    # comparison is against the oracle evaluated in fp64 and the loose one against its fp32 (reference-like) form
    want_xy64 = O.landmark_expected_coords(lg_c.detach().double(), batch, frame)
    assert float((dec["expect"].cpu().double() - want_xy64).abs().max()) < 1e-4
    want_xy32 = O.landmark_expected_coords(lg_c.detach(), batch, frame)
    assert float((dec["expect"].cpu() - want_xy32).abs().max()) < 2e-2
    # size-independent property: a constant added to every logit of a frame changes nothing
    dec2 = EV.decode_landmarks(lg.detach() + 7.5, batch, frame)
    assert torch.equal(dec2["argmax"], dec["argmax"])
    assert float((dec2["expect"] - dec["expect"]).abs().max()) < 1e-4


def test_decode_rejects_cpu_tensors():
    with pytest.raises(RuntimeError):
        ops.heatmap_expect_fwd(torch.zeros(8, 4), 1, [(0, 2)])


def test_bce_accepts_contiguous_views_at_odd_offsets():
    """eg_bce_logits_fwd reads 16 bytes per lane and refuses unaligned pointers; the Python wrapper copies a
    contiguous view that starts at an odd element offset instead of raising."""
    rs = np.random.RandomState(3)
    n = 4 * 1000 + 1
    flat = torch.from_numpy(rs.standard_normal(n).astype(np.float32)).to(DEV)
    lab = torch.from_numpy((rs.uniform(size=n) < 0.01).astype(np.float32)).to(DEV)
    val = torch.ones(n, device=DEV)
    x, y, v = flat[1:].view(-1, 4), lab[1:].view(-1, 4), val[1:].view(-1, 4)
    assert x.data_ptr() % 16 != 0 and x.is_contiguous()
    got = ops.bce_logits(x.clone().requires_grad_(True), y, v, 9000.0)
    xv = x.detach().requires_grad_(True)
    got_view = ops.bce_logits(xv, y, v, 9000.0)
    got_view.backward()
    assert torch.equal(got.detach(), got_view.detach()) and xv.grad is not None and torch.isfinite(xv.grad).all()


@pytest.mark.parametrize("frame,naux,B,coord", [(32, 4, 2, True), (224, 7, 1, True), (16, 3, 3, False)])
def test_fused_criteria_node_equals_the_criteria_one_by_one(monkeypatch, frame, naux, B, coord):
    """engine.compute_loss (src/engine.py:582-600): WeightedBCEWithLogitsLoss + ExpectedLandmarkMSE (+ MSE on the landmark
    coordinates) as ONE autograd node over eg_criteria_fwd / eg_criteria_bwd against the three criteria computed one by one
    (EG_FUSED_CRITERIA=0: the route the CPU oracles pin in this file): every value, the total, and the gradients w.r.t. the logits
    and the coordinates -- through ``.total``, through ``sum(values)`` and through one component alone."""
    from echoglad_amd import engine, losses
    rs = np.random.RandomState(frame + B)
    lv = losses.level_grids(frame, naux)
    n = lv[-1][0] + lv[-1][1] ** 2
    logits = torch.from_numpy(rs.standard_normal((B * n, 4)).astype(np.float32) * 2).to(DEV)
    y = torch.zeros(B, n, 4)
    for b in range(B):
        for c in range(4):
            for st, s in lv:
                y[b, st + rs.randint(0, s * s), c] = 1.0
    y = y.reshape(B * n, 4).to(DEV)
    valid = torch.from_numpy((rs.uniform(size=(B * n, 4)) > 0.1).astype(np.float32)).to(DEV)
    cp = torch.from_numpy(rs.uniform(0, frame - 1, (B * 4, 2)).astype(np.float32)).to(DEV)
    cy = torch.from_numpy(rs.uniform(0, frame - 1, (B * 4, 2)).astype(np.float32)).to(DEV)
    crit = {"bce": losses.WeightedBCEWithLogitsLoss("none", 9000, 1), "elm": losses.ExpectedLandmarkMSE(10, B, frame, naux)}
    if coord:
        crit["coordinate"] = engine.MSE(0.5)

    def run(knob, how):
        monkeypatch.setenv("EG_FUSED_CRITERIA", knob)
        x = logits.clone().requires_grad_(True)
        c = cp.clone().requires_grad_(True)
        ls = engine.compute_loss(crit, x, y, c if coord else None, cy if coord else None, valid, B)
        assert isinstance(ls, losses.LossDict) == (knob == "1") and list(ls) == list(crit)
        if how == "total":
            loss = engine.total_loss(ls)
        elif how == "sum":
            loss = sum(ls.values())
        else:
            loss = ls["elm"] * 3.0
        loss.backward()
        return {k: float(v) for k, v in ls.items()}, float(loss), x.grad.clone(), (c.grad.clone() if (coord and c.grad is not None) else None)

    for how in ("total", "sum", "elm"):
        va, la, ga, ca = run("1", how)
        vb, lb, gb, cb = run("0", how)
        for k in va:
            assert abs(va[k] - vb[k]) <= 2e-6 * abs(vb[k]) + 1e-7, (how, k, va[k], vb[k])
        assert abs(la - lb) <= 2e-6 * abs(lb) + 1e-7
        assert float((ga - gb).abs().max()) <= 2e-6 * float(gb.abs().max()) + 1e-12, how
        if coord and how != "elm":
            assert float((ca - cb).abs().max()) <= 2e-6 * float(cb.abs().max())
        if how == "elm":
            assert ca is None or float(ca.abs().max()) == 0.0
