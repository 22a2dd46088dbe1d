"""Data parallelism for the GNN hot path: one process per GPU, frames sharded over the batch dimension.

The reference's only multi-GPU mechanism is single-process ``torch_geometric.nn.DataParallel`` +
``DataListLoader`` (src/engine.py:84, :105-110; src/builders/dataloader_builder.py:17-22): per-step
parameter broadcast from GPU 0, scatter of the sample list, output gather and gradient reduce to GPU 0,
BatchNorm statistics per replica.  Here every rank owns a contiguous shard of frames (frames are
independent graph components, so the forward needs no exchange at all) and training adds exactly one
collective per step: a sum all-reduce of one flat fp32 gradient buffer over RCCL/xGMI (backend "nccl" on
ROCm; "gloo" in the CPU tests), divided by the world size.  BatchNorm statistics stay per rank, matching
the reference's DataParallel semantics."""
from __future__ import annotations

from typing import Iterable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced [lo, hi) of ``n_items`` frames for ``rank`` (first ``n % world`` ranks get one more)."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    q, r = divmod(n_items, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def shard_frames(rank: int, world: int, nodes_per_frame: int, *, frames: Optional[torch.Tensor] = None,
                 node_feats: Optional[torch.Tensor] = None, node_coords: Optional[torch.Tensor] = None,
                 labels: Optional[torch.Tensor] = None):
    """Slice a global batch into this rank's shard.  frames [B,...]; node_feats / labels [B*N, ...];
    node_coords [4B, 2].  Returns a dict with the same keys (views, no copies)."""
    if frames is not None:
        B = frames.shape[0]
    elif node_feats is not None:
        B = node_feats.shape[0] // nodes_per_frame
    else:
        raise ValueError("need frames or node_feats to infer the batch size")
    lo, hi = shard_range(B, rank, world)
    out = {"frame_range": (lo, hi)}
    if frames is not None:
        out["frames"] = frames[lo:hi]
    if node_feats is not None:
        out["node_feats"] = node_feats[lo * nodes_per_frame:hi * nodes_per_frame]
    if labels is not None:
        per = labels.shape[0] // B
        out["labels"] = labels[lo * per:hi * per]
    if node_coords is not None:
        out["node_coords"] = node_coords[4 * lo:4 * hi]
    return out


class GradientAllReducer:
    """One flat fp32 buffer for all gradients -> ONE all-reduce per step (277 KB for the GNN + classifier
    parameters, 32 MB with the UNet front-end: far below where a ring's per-link bandwidth matters, so a
    single collective keeps the launch/latency cost to one)."""

    def __init__(self, params: Iterable[torch.nn.Parameter], group=None, average: bool = True):
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        self.group = group
        self.average = average
        self._flat: Optional[torch.Tensor] = None

    def _buffer(self) -> torch.Tensor:
        n = sum(p.numel() for p in self.params)
        p0 = self.params[0]
        if self._flat is None or self._flat.numel() != n or self._flat.device != p0.device:
            self._flat = torch.zeros(n, dtype=torch.float32, device=p0.device)
        return self._flat

    def allreduce(self, async_op: bool = False):
        """Pack -> all_reduce(SUM) -> (divide) -> unpack into ``p.grad``.  Parameters without a gradient on this
        rank contribute zeros (every rank must call this with the same parameter list)."""
        if not self.params:
            return None
        flat = self._buffer()
        off = 0
        for p in self.params:
            n = p.numel()
            if p.grad is None:
                flat[off:off + n].zero_()
            else:
                flat[off:off + n].copy_(p.grad.reshape(-1))
            off += n
        world = dist.get_world_size(self.group) if dist.is_initialized() else 1
        work = None
        if world > 1:
            work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)
        if async_op and work is not None:
            return _Pending(self, work, world)
        self._finish(world)
        return None

    def _finish(self, world: int) -> None:
        flat = self._flat
        if self.average and world > 1:
            flat.div_(world)
        off = 0
        for p in self.params:
            n = p.numel()
            g = flat[off:off + n].view_as(p)
            if p.grad is None:
                p.grad = g.clone()
            else:
                p.grad.copy_(g)
            off += n


class _Pending:
    def __init__(self, owner: GradientAllReducer, work, world: int):
        self.owner, self.work, self.world = owner, work, world

    def wait(self) -> None:
        self.work.wait()
        self.owner._finish(self.world)


def broadcast_parameters(module: torch.nn.Module, src: int = 0, group=None) -> None:
    """Make every rank start from rank ``src``'s parameters and buffers (replaces DataParallel's per-step replicate)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src=src, group=group)
