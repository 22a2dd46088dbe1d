"""TEST INFRASTRUCTURE ONLY — CPU restatement of the reference's losses on the logits and of its landmark
decode / width-error evaluator.  Nothing under echoglad_amd/ may import this file;
only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do.

Follows, function by function:
  weighted_bce_with_logits   src/core/criterion.py:7-35   (WeightedBCE.compute with nn.BCEWithLogitsLoss)
  level_grids                src/core/criterion.py:78-87  (grid sizes / end indices of the flattened levels)
  gt_coords / softmax_expectation / expected_landmark_mse   src/core/criterion.py:93-151
  evaluate_landmarks         src/core/evaluators.py:291-391 (update), :393-432 (widths, MAE, MPE), :485-495 (softmax heat map)

Pinned: tests/test_oracle.py checks every function against tests/golden/decode_f16_a3.npz and decode_f30_a3.npz,
which tests/golden/make_golden.py produced by running the reference's own classes (losses, autograd gradients,
evaluator records) on seeded logits with near-ties and partly invalid labels.
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import numpy as np
import torch


def level_grids(frame_size: int, num_aux_graphs: int, use_main_graph_only: bool = False) -> List[Tuple[int, int]]:
    """[(first row, side)] of every level inside a frame's valid rows (criterion.py:78-87)."""
    sizes = [frame_size] if use_main_graph_only else [2 ** g for g in range(1, num_aux_graphs + 1)] + [frame_size]
    out, start = [], 0
    for s in sizes:
        out.append((start, s))
        start += s * s
    return out


def weighted_bce_with_logits(pred: torch.Tensor, y: torch.Tensor, valid: torch.Tensor, ones_weight: float,
                             loss_weight: float) -> torch.Tensor:
    """criterion.py:13-27: elementwise BCE-with-logits, x ones_weight where y == 1, sum(loss*valid)/sum(valid)."""
    loss = torch.nn.functional.binary_cross_entropy_with_logits(pred, y, reduction="none")
    valid = valid.view(pred.shape)
    if ones_weight > 1:
        w = torch.where(y == 1, torch.full_like(loss, float(ones_weight)), torch.ones_like(loss))
        loss = w * loss
    return loss_weight * (torch.sum(loss * valid) / torch.sum(valid))


def gt_coords(y_level: torch.Tensor) -> torch.Tensor:
    """[B,S,S,4] heat map -> [B,4,2] (h, w): argmax over rows of the row maxima / over columns of the column maxima
    (criterion.py:118-122, evaluators.py:329-333)."""
    max_along_w, _ = torch.max(y_level, dim=-2)
    max_along_h, _ = torch.max(y_level, dim=-3)
    _, gt_h = torch.max(max_along_w, dim=-2)
    _, gt_w = torch.max(max_along_h, dim=-2)
    return torch.stack((gt_h, gt_w), dim=2)


def softmax_expectation(pred_level: torch.Tensor, side: int) -> torch.Tensor:
    """[B,S*S,4] logits -> [B,4,2] expected (h, w) under a softmax over the level's nodes (criterion.py:124-135)."""
    B = pred_level.shape[0]
    p = torch.softmax(pred_level, dim=1).view(B, side, side, -1)
    hh = torch.linspace(0, side - 1, side).view(1, -1, 1, 1)
    ww = torch.linspace(0, side - 1, side).view(1, 1, -1, 1)
    return torch.stack(((p * hh).sum(dim=(1, 2)), (p * ww).sum(dim=(1, 2))), dim=2)


def expected_landmark_mse(pred: torch.Tensor, y: torch.Tensor, valid: torch.Tensor, batch: int, frame_size: int,
                          num_aux_graphs: int, use_main_graph_only: bool = False, loss_weight: float = 1.0,
                          num_output_channels: int = 4) -> torch.Tensor:
    """criterion.py:93-151."""
    pred = pred.view(batch, -1, num_output_channels)
    y = y.view(batch, -1, num_output_channels)
    valid = valid.view(batch, -1, num_output_channels)
    loss = 0
    for start, side in level_grids(frame_size, num_aux_graphs, use_main_graph_only):
        end = start + side * side
        gt = gt_coords(y[:, start:end, :].reshape(batch, side, side, num_output_channels)).to(pred.dtype)
        vs = valid[:, start:end, :].permute(0, 2, 1).mean(dim=-1).unsqueeze(-1)               # [B,4,1]
        nv = vs.sum(dim=0, keepdim=True)
        nv = torch.where(nv == 0, torch.ones_like(nv), nv)
        ex = softmax_expectation(pred[:, start:end, :], side)
        d = ((ex / side - gt / side) ** 2) * vs
        loss = loss + (d.sum(dim=0, keepdim=True) / nv).sum()
    return loss * loss_weight


def pixel_length(x0, y0, x1, y1, pix2mm_x, pix2mm_y):
    """evaluators.py:619-620."""
    return torch.sqrt(((x0 - x1) * pix2mm_x) ** 2 + ((y0 - y1) * pix2mm_y) ** 2)


def evaluate_landmarks(pred: torch.Tensor, y: torch.Tensor, pix2mm_x: torch.Tensor, pix2mm_y: torch.Tensor,
                       valid: torch.Tensor, batch: int, frame_size: int) -> Dict[str, object]:
    """One `LandmarkExpectedCoordiantesEvaluator.update` (evaluators.py:291-391, non-coordinate-graph branch).
    Returns the numbers it records: per-landmark mm errors, width MAE / MPE sums, the coordinates and widths."""
    F = frame_size
    pred = pred.view(batch, -1, pred.shape[-1]).detach()
    y = y.view(batch, -1, y.shape[-1]).detach()
    valid = valid.view(batch, -1, valid.shape[-1])
    vs = valid[:, -F * F:, :].permute(0, 2, 1).mean(dim=-1)                                    # [B,4]
    nv = vs.sum(dim=0, keepdim=True)
    present = [bool(nv[0, i] > 0) for i in range(4)]
    nv = torch.where(nv == 0, torch.ones_like(nv), nv)
    gt = gt_coords(y[:, -F * F:, :].reshape(batch, F, F, -1))
    preds = softmax_expectation(pred[:, -F * F:, :], F)
    gt_h, gt_w, pr_h, pr_w = gt[:, :, 0], gt[:, :, 1], preds[:, :, 0], preds[:, :, 1]
    err = pixel_length(gt_w, gt_h, pr_w, pr_h, pix2mm_x.unsqueeze(1), pix2mm_y.unsqueeze(1)).numpy()
    err = err * vs.numpy()
    err = np.squeeze(np.sum(err, axis=0) / nv.numpy())
    names = ["lvid_top", "lvid_bot", "lvpw", "ivs"]
    out: Dict[str, object] = {n: float(err[i]) for i, n in enumerate(names)}

    def widths_of(c):
        return {"ivs": pixel_length(c[:, 3, 1], c[:, 3, 0], c[:, 0, 1], c[:, 0, 0], pix2mm_x, pix2mm_y),
                "lvid": pixel_length(c[:, 0, 1], c[:, 0, 0], c[:, 1, 1], c[:, 1, 0], pix2mm_x, pix2mm_y),
                "lvpw": pixel_length(c[:, 1, 1], c[:, 1, 0], c[:, 2, 1], c[:, 2, 0], pix2mm_x, pix2mm_y)}

    wp, wg = widths_of(preds), widths_of(gt)
    w_lvid = vs[:, 0] * vs[:, 1] / torch.min(nv[0, 0], nv[0, 1])
    w_ivs = vs[:, 3] / nv[0, 3]
    w_lvpw = vs[:, 2] / nv[0, 2]
    for key, wt in (("ivs", w_ivs), ("lvid", w_lvid), ("lvpw", w_lvpw)):
        mae = torch.abs(wp[key] - wg[key])
        out[key + "_w"] = float((mae * wt).sum())
        out[key + "_mpe"] = float((100 * mae / wg[key] * wt).sum())
    out["present"] = present
    out["pred_coords"] = preds
    out["gt_coords"] = gt
    out["widths"] = {**{"pred_" + k + "_mm": v for k, v in wp.items()}, **{"gt_" + k + "_mm": v for k, v in wg.items()}}
    return out
