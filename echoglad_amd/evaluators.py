"""Landmark decode and width errors — host-side mirror of ``LandmarkExpectedCoordiantesEvaluator``
(src/core/evaluators.py:237-617; the class name keeps the reference's spelling so a builder can swap it in) over
the HIP decode kernel in csrc/heatmap.hip.

The reference moves the full logits to the host every step (engine.py:466-492) and evaluates the softmax heat map
there.  Here the logits stay on the device: one kernel pass yields, per frame and landmark, the softmax-expected
(h, w) over the last F*F rows, the label's (h, w) and the mean of ``valid``; only those [B,4,*] numbers are read back.
"""
from __future__ import annotations

import numpy as np
import torch

from . import ops

NAMES = ("lvid_top", "lvid_bot", "lvpw", "ivs")


def decode_landmarks(logits: torch.Tensor, batch_size: int, frame_size: int, labels=None, valid=None):
    """Device-side decode of the main-grid heat maps (the last F*F rows of every frame).

    Returns dict: expect [B,4,2] softmax-expected (h, w); argmax [B,4] hard arg max row index (h * F + w);
    gt [B,4,2] / vmean [B,4] when labels / valid are given."""
    lg = logits.reshape(-1, logits.shape[-1]).to(torch.float32).contiguous()
    n_rows = lg.shape[0] // batch_size
    level = [(n_rows - frame_size * frame_size, frame_size)]
    prep = (lambda t: None if t is None else t.reshape(-1, t.shape[-1]).to(torch.float32).contiguous())
    r = ops.heatmap_expect_fwd(lg, batch_size, level, prep(labels), prep(valid), want_argmax=True)
    return {"expect": r["expect"][:, 0], "argmax": r["argmax"][:, 0],
            "gt": None if r["gt"] is None else r["gt"][:, 0], "vmean": None if r["vmean"] is None else r["vmean"][:, 0]}


def pixel_length(x0, y0, x1, y1, pix2mm_x, pix2mm_y):
    """evaluators.py:619-620."""
    return torch.sqrt(((x0 - x1) * pix2mm_x) ** 2 + ((y0 - y1) * pix2mm_y) ** 2)


class LandmarkExpectedCoordiantesEvaluator(object):
    """Same constructor, methods and recorded numbers as the reference class (evaluators.py:237-617)."""

    def __init__(self, logger, batch_size, frame_size, use_coord_graph):
        self.batch_size = batch_size
        self.frame_size = frame_size
        self.use_coord_graph = use_coord_graph
        self.detailed_performance = {}
        self.reset()

    def reset(self):
        self.coordinate_errors = {k: [] for k in ("ivs", "lvid_top", "lvid_bot", "lvpw")}
        self.valid_errors = {k: [] for k in ("ivs", "lvid_top", "lvid_bot", "lvpw")}
        self.width_MAE = {k: [] for k in ("lvid", "ivs", "lvpw")}
        self.width_MPE = {k: [] for k in ("lvid", "ivs", "lvpw")}
        self.detailed_performance.clear()

    def update(self, y_pred, y_true, pix2mm_x, pix2mm_y, valid):
        """evaluators.py:291-391.  y_pred / y_true / valid: [B * nodes, 4] (device tensors are decoded on the device)."""
        self.detailed_performance.clear()
        B, F = self.batch_size, self.frame_size
        if self.use_coord_graph:
            preds = y_pred.detach().reshape(-1, 4, 2).float().cpu()
            gt = y_true.detach().reshape(-1, 4, 2).float().cpu()
            # the reference uses valid_subset / num_valid_samples of the heat-map branch here and fails when they are
            # undefined (evaluators.py:352-354); every landmark of every frame counts as labelled in this branch
            vs = torch.ones(preds.shape[0], 4)
        else:
            d = decode_landmarks(y_pred.detach(), B, F, y_true.detach(), valid)
            preds, gt, vs = d["expect"].cpu(), d["gt"].cpu(), d["vmean"].cpu()
        pix2mm_x, pix2mm_y = pix2mm_x.detach().cpu().float(), pix2mm_y.detach().cpu().float()
        nv = vs.sum(dim=0, keepdim=True)
        for i, name in enumerate(NAMES):
            self.valid_errors[name].append(bool(nv[0, i] > 0))
        nv = torch.where(nv == 0, torch.ones_like(nv), nv)
        gt_h, gt_w, pr_h, pr_w = gt[:, :, 0], gt[:, :, 1], preds[:, :, 0], preds[:, :, 1]
        err = pixel_length(gt_w, gt_h, pr_w, pr_h, pix2mm_x.unsqueeze(1), pix2mm_y.unsqueeze(1)).numpy()
        err = np.squeeze(np.sum(err * vs.numpy(), axis=0) / nv.numpy())
        for i, name in enumerate(NAMES):
            self.coordinate_errors[name].append(err[i])
        widths = self.calculate_widths(preds, gt, pix2mm_x, pix2mm_y)
        w_lvid = vs[:, 0] * vs[:, 1] / torch.min(nv[0, 0], nv[0, 1])
        w_ivs = vs[:, 3] / nv[0, 3]
        w_lvpw = vs[:, 2] / nv[0, 2]
        ivs_e, lvid_e, lvpw_e = self.calculate_width_MAE(widths)
        self.width_MAE["ivs"].append((ivs_e * w_ivs).sum().item())
        self.width_MAE["lvid"].append((lvid_e * w_lvid).sum().item())
        self.width_MAE["lvpw"].append((lvpw_e * w_lvpw).sum().item())
        ivs_e, lvid_e, lvpw_e = self.calculate_width_MPE(widths)
        self.width_MPE["ivs"].append((ivs_e * w_ivs).sum().item())
        self.width_MPE["lvid"].append((lvid_e * w_lvid).sum().item())
        self.width_MPE["lvpw"].append((lvpw_e * w_lvpw).sum().item())
        coordinates = {"pred_ivs": preds[:, 3], "pred_lvid_top": preds[:, 0], "pred_lvid_bot": preds[:, 1],
                       "pred_lvpw": preds[:, 2], "gt_ivs": gt[:, 3], "gt_lvid_top": gt[:, 0], "gt_lvid_bot": gt[:, 1],
                       "gt_lvpw": gt[:, 2]}
        self.detailed_performance = {"widths": widths, "coordinates": coordinates}

    def calculate_widths(self, preds, gt, pix2mm_x, pix2mm_y):
        """evaluators.py:393-407: [N,4,2] (h, w) -> landmark-pair distances in mm."""
        def w3(c, tag):
            return {tag + "_ivs_mm": pixel_length(c[:, 3, 1], c[:, 3, 0], c[:, 0, 1], c[:, 0, 0], pix2mm_x, pix2mm_y),
                    tag + "_lvid_mm": pixel_length(c[:, 0, 1], c[:, 0, 0], c[:, 1, 1], c[:, 1, 0], pix2mm_x, pix2mm_y),
                    tag + "_lvpw_mm": pixel_length(c[:, 1, 1], c[:, 1, 0], c[:, 2, 1], c[:, 2, 0], pix2mm_x, pix2mm_y)}
        return {**w3(preds, "pred"), **w3(gt, "gt")}

    def calculate_width_MAE(self, widths):
        return (torch.abs(widths["pred_ivs_mm"] - widths["gt_ivs_mm"]), torch.abs(widths["pred_lvid_mm"] - widths["gt_lvid_mm"]),
                torch.abs(widths["pred_lvpw_mm"] - widths["gt_lvpw_mm"]))

    def calculate_width_MPE(self, widths):
        return tuple(100 * torch.abs(widths["pred_" + k] - widths["gt_" + k]) / widths["gt_" + k]
                     for k in ("ivs_mm", "lvid_mm", "lvpw_mm"))

    def compute(self):
        """evaluators.py:428-447: means over the recorded iterations, counting only iterations with a labelled landmark."""
        def cnt(*keys):
            m = np.asarray(self.valid_errors[keys[0]])
            for k in keys[1:]:
                m = np.logical_and(m, np.asarray(self.valid_errors[k]))
            return np.count_nonzero(m)
        t = {k: np.asarray(self.coordinate_errors[k]).sum() / cnt(k) for k in NAMES}
        t["ivs_w"] = np.asarray(self.width_MAE["ivs"]).sum() / cnt("ivs")
        t["lvid_w"] = np.asarray(self.width_MAE["lvid"]).sum() / cnt("lvid_top", "lvid_bot")
        t["lvpw_w"] = np.asarray(self.width_MAE["lvpw"]).sum() / cnt("lvpw")
        t["ivs_mpe"] = np.asarray(self.width_MPE["ivs"]).sum() / cnt("ivs")
        t["lvid_mpe"] = np.asarray(self.width_MPE["lvid"]).sum() / cnt("lvid_top", "lvid_bot")
        t["lvpw_mpe"] = np.asarray(self.width_MPE["lvpw"]).sum() / cnt("lvpw")
        return t

    def get_sum_of_width_MAE(self):
        t = self.compute()
        return sum(v for k, v in t.items() if k in ("ivs_w", "lvid_w", "lvpw_w"))

    def get_sum_of_width_MPE(self):
        t = self.compute()
        return sum(v for k, v in t.items() if k in ("ivs_mpe", "lvid_mpe", "lvpw_mpe"))

    def get_last(self):
        t = {k: self.coordinate_errors[k][-1] for k in NAMES}
        for k in ("ivs", "lvid", "lvpw"):
            t[k + "_w"] = self.width_MAE[k][-1]
            t[k + "_mpe"] = self.width_MPE[k][-1]
        return t

    def get_predictions(self):
        return self.detailed_performance


LandmarkExpectedCoordinatesEvaluator = LandmarkExpectedCoordiantesEvaluator
