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
    if num_output_channels == 4:
        from .losses import fused_criteria
        fused = fused_criteria(criterion, node_landmark_preds, node_landmark_y, valid_labels, node_coord_preds, node_coord_y, batch_size)
        if fused is not None:                # one autograd node, 5 launches; ``.total`` is the sum out of the same node
            return fused
    losses = {}
    preds = node_landmark_preds.view(batch_size, -1, num_output_channels)
    y = node_landmark_y.view(batch_size, -1, num_output_channels)
    for name, crit in criterion.items():
        if name == "coordinate":
            losses[name] = crit.compute(node_coord_preds, node_coord_y)
        else:
            losses[name] = crit.compute(preds, y, valid_labels)
    return losses


def total_loss(losses) -> torch.Tensor:
    """engine.py:271: the sum of the criteria -- straight out of the fused node when compute_loss used it."""
    total = getattr(losses, "total", None)
    return total if total is not None else sum(losses.values())


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
    loss = total_loss(losses)
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


class GraphedTrainStep:
    """One WHOLE training step -- forward, criteria, backward, optimizer update (engine.py:239-273 of the reference) -- captured
    once into a HIP graph and replayed: ~130 kernel launches per step leave the host as one.  At the reference's own batch size
    (configs/default.yml:27, ``batch_size: 1``) an eager step is bound by the host's launch rate (2.4 - 3.4 ms for 1.0 ms of GPU
    work); at batch 32 the GPU is the bound and a replay changes nothing.

    ``loss_fn()`` -> loss, or (loss, *tensors to keep): it must read its inputs from tensors that stay where they are (write a new
    batch INTO them before calling the step) and must not synchronise with the host.  Dropout: the seeds a train-mode forward
    draws on the host are frozen into the graph's kernel arguments, so the graph's first node bumps the device's dropout EPOCH
    (include/echoglad_hip.h: the kernels hash with seed + epoch) -- every replay draws fresh masks, its forward and backward see
    the same ones.  BatchNorm running statistics, ``num_batches_tracked`` and the optimizer's step count live on the device and
    advance with every replay.

    Single rank (``reducer`` None): the optimizer update is part of the graph, so ``optimizer`` has to be capturable
    (``torch.optim.Adam(..., capturable=True)``; ``fused=True`` as well for one launch).
    Data parallel (``reducer`` = a ``parallel.GradientAllReducer`` WITHOUT attached hooks): the graph holds forward + backward
    only; every call replays it, averages the gradients over the ranks with the reducer's hook-less ``allreduce()`` (a handful of
    RCCL collectives on ~280 KB: they are not captured) and runs ``optimizer.step()`` eagerly -- any optimizer will do.  The
    overlap of the collectives with backward that ``train_step`` + ``attach_hooks()`` gives is traded for the launch-free step:
    the trade pays where the step is host-bound (small per-rank batches), not at batch 32.

    ``warmup`` eager steps run first (on the capture stream, as torch.cuda.graph asks): they allocate every workspace, topology
    handle and the epoch word outside the capture -- and they ARE training steps."""

    def __init__(self, loss_fn, optimizer, warmup: int = 3, reducer: Optional[GradientAllReducer] = None):
        from . import ops
        if warmup < 1:
            raise ValueError("at least one eager warm-up step is needed (workspaces and handles are created by it)")
        if reducer is None:
            for group in optimizer.param_groups:
                if not group.get("capturable", False):
                    raise ValueError("GraphedTrainStep needs a capturable optimizer (e.g. torch.optim.Adam(params, capturable=True))")
        elif getattr(reducer, "_hooks", None):
            raise ValueError("GraphedTrainStep drives the reducer itself: pass a GradientAllReducer without attach_hooks() "
                             "(collectives fired from inside backward cannot be part of the captured graph)")
        self.loss_fn, self.optimizer, self.reducer = loss_fn, optimizer, reducer
        self.replays = 0
        self._written = [p for g in optimizer.param_groups for p in g["params"]]
        stream = torch.cuda.Stream()
        stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(stream):
            for _ in range(warmup):
                self.optimizer.zero_grad(set_to_none=True)
                self._forward_backward(ops, bump=False)
                self._update()
        torch.cuda.current_stream().wait_stream(stream)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        optimizer.zero_grad(set_to_none=True)
        with torch.cuda.graph(self.graph, stream=stream):
            self.outputs = self._forward_backward(ops, bump=True)
            if reducer is None:
                self.optimizer.step()

    def _forward_backward(self, ops, bump: bool):
        if bump:
            ops.dropout_epoch_add(1)
        out = self.loss_fn()
        loss = out[0] if isinstance(out, (tuple, list)) else out
        # (d loss = 1 from a tensor that exists already: autograd's own ones_like is a fill kernel -- one more node of every replay)
        one = getattr(self, "_one", None)
        if one is None or one.device != loss.device or one.dtype != loss.dtype or one.shape != loss.shape:
            one = self._one = torch.ones_like(loss)
        loss.backward(gradient=one)
        if isinstance(out, (tuple, list)):
            return tuple(t.detach() for t in out)
        return (loss.detach(),)

    def _update(self):
        if self.reducer is not None:
            self.reducer.allreduce()
        self.optimizer.step()

    def __call__(self):
        """Replay the step; returns the graph's static output tensors (loss first), valid until the next replay."""
        self.graph.replay()
        if self.reducer is not None:
            self._update()
        else:
            # a replay runs no host code: the in-place version counters of what the captured update wrote stand still -- and whatever
            # is cached on them (the model's folded inference parameters, its inference graphs) would not notice the step
            torch.autograd.graph.increment_version(self._written)
        self.replays += 1
        return self.outputs


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
