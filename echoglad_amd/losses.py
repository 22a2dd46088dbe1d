"""Losses on the logits — host-side mirror of the reference's criterion classes over the HIP kernels in
csrc/heatmap.hip.  Same class names, constructor arguments and ``compute`` signatures as
``src/core/criterion.py`` so that a criterion builder can swap them in; everything stays on the device (the
reference's WeightedBCE round-trips the labels through numpy to build its weight tensor, criterion.py:18-21).
"""
from __future__ import annotations

from typing import List, Tuple

import torch

from . import ops


def level_grids(frame_size: int, num_aux_graphs: int, use_main_graph_only: bool = False) -> List[Tuple[int, int]]:
    """[(first row, side)] of every level inside a frame's rows (criterion.py:78-87)."""
    sizes = [frame_size] if use_main_graph_only else [2 ** g for g in range(1, num_aux_graphs + 1)] + [frame_size]
    out, start = [], 0
    for s in sizes:
        out.append((start, s))
        start += s * s
    return out


def _rows4(t: torch.Tensor) -> torch.Tensor:
    return t.reshape(-1, t.shape[-1]).to(torch.float32).contiguous()


class WeightedBCEWithLogitsLoss:
    """criterion.py:29-33 on top of WeightedBCE (:7-27): elementwise BCE-with-logits, ``ones_weight`` on the
    positive labels, ``sum(loss * valid) / sum(valid)`` times ``loss_weight``."""

    def __init__(self, reduction, ones_weight, loss_weight):
        if reduction != "none":
            raise NotImplementedError("the reference builds this loss with reduction='none' (configs/default.yml)")
        self.ones_weight = ones_weight
        self.loss_weight = loss_weight

    def compute(self, pred_y, y, valid=None):
        x = _rows4(pred_y)
        return self.loss_weight * ops.bce_logits(x, _rows4(y), None if valid is None else _rows4(valid), self.ones_weight)


class ExpectedLandmarkMSE:
    """criterion.py:63-151: per level, softmax over the level's nodes -> expected (h, w), squared distance to the
    label's (h, w) in units of the level's side, weighted by the per-(frame, channel) mean of ``valid``."""

    def __init__(self, loss_weight=1, batch_size=2, frame_size=128, num_aux_graphs=6, use_main_graph_only=False,
                 num_output_channels=4):
        if num_output_channels != 4:
            raise NotImplementedError("the HIP kernels are built for 4 landmark channels")
        self.loss_weight = loss_weight
        self.batch_size = batch_size
        self.frame_size = frame_size
        self.num_aux_graphs = num_aux_graphs
        self.num_output_channels = num_output_channels
        self.use_main_graph_only = use_main_graph_only
        self.levels = level_grids(frame_size, num_aux_graphs, use_main_graph_only)
        self.grid_sizes = [s for _, s in self.levels]
        self.end_indices = [st + s * s for st, s in self.levels]
        self._side = {}                    # device -> [L] 1 / level side (built once: a host list -> device tensor is a blocking copy)

    def compute(self, pred_y, y, valid):
        expect, gt, vmean = ops.heatmap_expect(_rows4(pred_y), self.batch_size, self.levels, _rows4(y), _rows4(valid))
        inv_side = self._side.get(expect.device)
        if inv_side is None:
            inv_side = self._side[expect.device] = (1.0 / torch.tensor(self.grid_sizes, dtype=torch.float32,
                                                                       device=expect.device)).contiguous()
        return _ElmReduceFn.apply(expect, gt, vmean, inv_side, float(self.loss_weight))


class _ElmReduceFn(torch.autograd.Function):
    """criterion.py:133-151 on the per-(frame, level, channel) expectations:
        loss = w * sum_{l,c,xy} [ sum_b ((e - gt) / side)^2 * vmean ] / nv,    nv = sum_b vmean (1 where that is 0)
    value AND gradient from one single-workgroup launch (eg_elm_reduce); the backward is one multiply.  (Autograd's graph of the
    same expression took about 35 launches, round 4's hand-written torch form 13: each costs ~5 us of GPU time and ~10 us of host
    time whatever it computes, and at batch 1 the training step is bound by the host.)"""

    @staticmethod
    def forward(ctx, expect, gt, vmean, inv_side, weight):
        loss, d = ops.elm_reduce(expect.contiguous(), gt.contiguous(), vmean.contiguous(), inv_side, weight)
        ctx.save_for_backward(d)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        (d,) = ctx.saved_tensors
        return d * g, None, None, None, None


class LossDict(dict):
    """``engine.compute_loss``'s result when the criteria ran as one node: the reference's ``{name: loss}`` dictionary
    (engine.py:582-600) plus ``total`` = their sum as it comes out of the same node (``sum(d.values())`` gives the same value and
    gradients through three more autograd nodes and a handful of tiny launches)."""
    total: torch.Tensor


def fused_criteria(criterion: dict, preds, y, valid, coord_preds, coord_y, batch_size: int):
    """The criteria of engine.py:582-600 as ONE autograd node (ops.landmark_criteria) when they are exactly the configured set --
    a WeightedBCEWithLogitsLoss and an ExpectedLandmarkMSE on the logits and optionally ``engine.MSE`` under the name
    'coordinate' -- else None (the caller computes them one by one).  EG_FUSED_CRITERIA=0: always one by one."""
    import os
    if os.environ.get("EG_FUSED_CRITERIA", "1") == "0":
        return None
    bce = [(k, c) for k, c in criterion.items() if isinstance(c, WeightedBCEWithLogitsLoss)]
    elm = [(k, c) for k, c in criterion.items() if isinstance(c, ExpectedLandmarkMSE)]
    rest = [k for k, c in criterion.items() if not isinstance(c, (WeightedBCEWithLogitsLoss, ExpectedLandmarkMSE))]
    if len(bce) != 1 or len(elm) != 1 or any(k != "coordinate" for k in rest):
        return None
    coord = criterion.get("coordinate")
    if coord is not None and (type(coord).__name__ != "MSE" or coord_preds is None or coord_y is None):
        return None
    if valid is None or not preds.is_cuda or elm[0][1].batch_size != batch_size:
        return None
    (kb, cb), (ke, ce) = bce[0], elm[0]
    x, yy, vv = _rows4(preds), _rows4(y), _rows4(valid)
    inv_side = ce._side.get(x.device)
    if inv_side is None:
        inv_side = ce._side[x.device] = (1.0 / torch.tensor(ce.grid_sizes, dtype=torch.float32, device=x.device)).contiguous()
    total, vb, ve, vc = ops.landmark_criteria(x, yy, vv, batch_size, ce.levels, inv_side, cb.ones_weight, cb.loss_weight, ce.loss_weight,
                                              coord_preds if coord is not None else None, coord_y if coord is not None else None,
                                              coord.loss_weight if coord is not None else 1.0)
    out = LossDict()
    for k in criterion:                      # the caller's order
        out[k] = vb if k == kb else (ve if k == ke else vc)
    out.total = total
    return out
