"""One training / evaluation step over the HIP path — host-side counterpart of the inner loops of the reference's
``Engine`` (src/engine.py:239-273 train step, :394-398 eval forward, :582-600 ``compute_loss``): embed the frames,
run the landmark model, sum the criteria, backward, (data-parallel gradient all-reduce,) optimizer step, update the
evaluators.  Logging, checkpointing and wandb of the reference are out of scope.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

from .parallel import GradientAllReducer


class MSE:
    """criterion.py:36-48 (the 'coordinate' criterion of the coordinate-graph configs)."""

    def __init__(self, loss_weight=1):
        self.loss_weight = loss_weight

    def compute(self, pred_y, y):
        return self.loss_weight * torch.nn.functional.mse_loss(pred_y, y)


class MAE:
    """criterion.py:51-63."""

    def __init__(self, loss_weight=1):
        self.loss_weight = loss_weight

    def compute(self, pred_y, y):
        return self.loss_weight * torch.nn.functional.l1_loss(pred_y, y)


def compute_loss(criterion: Dict[str, object], node_landmark_preds, node_landmark_y, node_coord_preds, node_coord_y,
                 valid_labels, batch_size: int, num_output_channels: int = 4) -> Dict[str, torch.Tensor]:
    """engine.py:582-600: the 'coordinate' criterion sees the coordinate predictions, every other one the logits
    reshaped to [B, nodes, channels]."""
    losses = {}
    preds = node_landmark_preds.view(batch_size, -1, num_output_channels)
    y = node_landmark_y.view(batch_size, -1, num_output_channels)
    for name, crit in criterion.items():
        if name == "coordinate":
            losses[name] = crit.compute(node_coord_preds, node_coord_y)
        else:
            losses[name] = crit.compute(preds, y, valid_labels)
    return losses


def forward_batch(model: Dict[str, torch.nn.Module], batch, use_coordinate_graph: bool):
    """engine.py:239-255: frame embeddings, then the landmark model on the collated batch."""
    x = model["embedder"](batch.x)
    node_coords = batch.node_coords if use_coordinate_graph else None
    return model["landmark"](x=x, node_coords=node_coords, edge_index=batch.edge_index, batch_idx=batch.batch,
                             node_type=batch.node_type)


def train_step(model: Dict[str, torch.nn.Module], batch, criterion: Dict[str, object], optimizer, batch_size: int,
               use_coordinate_graph: bool = False, reducer: Optional[GradientAllReducer] = None, evaluators=None):
    """engine.py:239-291 for one batch.  ``reducer`` (parallel.GradientAllReducer) averages the gradients over the
    data-parallel ranks between backward and the optimizer step; it replaces torch_geometric's DataParallel
    (engine.py:105-110).  Returns (loss, losses dict, logits, coordinate predictions)."""
    preds, coord_preds = forward_batch(model, batch, use_coordinate_graph)
    coord_y = batch.node_coord_y if use_coordinate_graph else None
    losses = compute_loss(criterion, preds, batch.y, coord_preds, coord_y, batch.valid_labels, batch_size)
    loss = sum(losses.values())
    optimizer.zero_grad()
    loss.backward()                 # with reducer.attach_hooks(): the buckets' all-reduces are issued from inside backward
    if reducer is not None:
        reducer.finish()
    optimizer.step()
    if evaluators:
        with torch.no_grad():
            update_evaluators(evaluators, preds, batch.y, coord_preds, coord_y, batch.pix2mm_x, batch.pix2mm_y,
                              batch.valid_labels, use_coordinate_graph)
    return loss.detach(), {k: v.detach() for k, v in losses.items()}, preds.detach(), coord_preds


@torch.no_grad()
def eval_step(model: Dict[str, torch.nn.Module], batch, criterion: Optional[Dict[str, object]], batch_size: int,
              use_coordinate_graph: bool = False, evaluators=None):
    """engine.py:340-460 for one batch (models in eval mode are the caller's business, like in the reference)."""
    preds, coord_preds = forward_batch(model, batch, use_coordinate_graph)
    coord_y = batch.node_coord_y if use_coordinate_graph else None
    losses = {}
    if criterion:
        losses = compute_loss(criterion, preds, batch.y, coord_preds, coord_y, batch.valid_labels, batch_size)
    if evaluators:
        update_evaluators(evaluators, preds, batch.y, coord_preds, coord_y, batch.pix2mm_x, batch.pix2mm_y,
                          batch.valid_labels, use_coordinate_graph)
    return preds, coord_preds, losses


def update_evaluators(evaluators, preds, y, coord_preds, coord_y, pix2mm_x, pix2mm_y, valid, use_coordinate_graph):
    """engine.py:466-492 without the `.detach().cpu()` of every tensor: the evaluators decode on the device."""
    for ev in evaluators.values():
        if use_coordinate_graph:
            ev.update(coord_preds, coord_y, pix2mm_x, pix2mm_y, valid)
        else:
            ev.update(preds, y, pix2mm_x, pix2mm_y, valid)
