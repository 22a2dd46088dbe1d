"""Data parallelism for the GNN hot path: one process per GPU, frames sharded over the batch dimension.

The reference's only multi-GPU mechanism is single-process ``torch_geometric.nn.DataParallel`` +
``DataListLoader`` (src/engine.py:84, :105-110; src/builders/dataloader_builder.py:17-22): per-step
parameter broadcast from GPU 0, scatter of the sample list, output gather and gradient reduce to GPU 0,
BatchNorm statistics per replica.  Here every rank owns a contiguous shard of frames (frames are
independent graph components, so the forward needs no exchange at all) and training adds exactly one
kind of collective per step: sum all-reduces over the buckets of one flat fp32 gradient buffer over RCCL/xGMI (backend
"nccl" on ROCm; "gloo" in the CPU tests), issued from inside backward on a side stream and divided by the world size.  BatchNorm statistics stay per rank, matching
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


class _Bucket:
    __slots__ = ("params", "lo", "hi", "ready", "launched")

    def __init__(self, params, lo, hi):
        self.params, self.lo, self.hi = params, lo, hi
        self.ready, self.launched = 0, False


class GradientAllReducer:
    """Bucketed gradient all-reduce, overlapped with backward.

    The parameters are laid out in ONE flat fp32 buffer in REVERSE registration order (the order backward finishes
    them: classifier heads first, the first GNN layer last) and cut into a few contiguous buckets.  With
    ``attach_hooks()`` every parameter's post-accumulate-grad hook counts its bucket down; the moment a bucket is
    complete AND every bucket before it has gone out, it is packed (one ``torch.cat`` into its slice) on the producing
    stream and its sum all-reduce is issued on a SIDE stream behind an event, so RCCL moves the finished layers'
    gradients over xGMI while the kernels of the remaining backward keep running.  Buckets leave strictly in index
    order on every rank: RCCL and gloo pair collectives by ISSUE ORDER, not by tensor, so a bucket that completes early on
    one rank (a parameter without a gradient there) must not overtake.  ``finish()`` (after ``loss.backward()``) issues
    whatever never fired, in order (parameters without a gradient contribute zeros), waits for the collectives, divides
    by the world size and hands the reduced values back as ``p.grad``.  If backward raised, call ``reset()`` before the
    next step.  Replaces torch_geometric's DataParallel reduce-to-GPU-0 (src/engine.py:105-110).

    ``force_collective=True`` issues the collectives (side stream, event, async all-reduce) at world size 1 as well: the
    single-GPU way to run the exact code path the multi-GPU step takes.

    Sizing for xGMI (point-to-point links, ~20 us per collective): the 277 KB of GNN + classifier gradients go out as
    at most ~5 collectives; with a 32 MB front-end the buckets grow to total / 8 (4 MB), capped at 8 MB."""

    def __init__(self, params: Iterable[torch.nn.Parameter], group=None, average: bool = True,
                 bucket_bytes: Optional[int] = None, force_collective: bool = False):
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        self.group = group
        self.average = average
        self.force_collective = bool(force_collective)
        self._next = 0                      # index of the next bucket to go out (strict order)
        self.collectives_issued = 0         # all-reduces issued so far (diagnostics / tests)
        order = list(reversed(self.params))
        total = sum(p.numel() for p in order)
        if bucket_bytes is None:
            bucket_bytes = min(max(4 * total // 8, 64 << 10), 8 << 20)
        self.bucket_bytes = int(bucket_bytes)
        self._flat: Optional[torch.Tensor] = None
        self._total = total
        self._buckets: List[_Bucket] = []
        self._where = {}
        lo = off = 0
        cur: List[torch.nn.Parameter] = []
        for p in order:
            cur.append(p)
            off += p.numel()
            if 4 * (off - lo) >= self.bucket_bytes:
                self._buckets.append(_Bucket(cur, lo, off))
                cur, lo = [], off
        if cur:
            self._buckets.append(_Bucket(cur, lo, off))
        for bi, b in enumerate(self._buckets):
            for p in b.params:
                self._where[id(p)] = bi
        self._hooks = []
        self._works = []
        self._side = None
        # diagnostics (bench.py --mode train at world > 1): how long the compute stream sat in finish() behind the collectives
        self.profile = False
        self._wait_events: List[tuple] = []
        self.finish_calls = 0
        # optional trace of one or more steps (set to a list): one entry per bucket launch --
        # (bucket index, fired from a gradient hook inside backward?, issued on the side stream?)
        self.trace: Optional[list] = None
        self._in_hook = False

    # ---- plumbing -----------------------------------------------------------------------------------------------
    def _buffer(self) -> torch.Tensor:
        p0 = self.params[0]
        if self._flat is None or self._flat.device != p0.device:
            self._flat = torch.zeros(self._total, dtype=torch.float32, device=p0.device)
        return self._flat

    def _world(self) -> int:
        return dist.get_world_size(self.group) if dist.is_initialized() else 1

    def attach_hooks(self) -> "GradientAllReducer":
        """Fire each bucket's all-reduce from inside backward, as soon as its last gradient has been accumulated."""
        if not self._hooks:
            for p in self.params:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))
        return self

    def detach_hooks(self) -> None:
        for h in self._hooks:
            h.remove()
        self._hooks = []

    def _on_grad(self, p: torch.nn.Parameter) -> None:
        self._buckets[self._where[id(p)]].ready += 1
        # every complete bucket at the head of the line goes out; a complete bucket behind an incomplete one waits
        self._in_hook = True
        try:
            while self._next < len(self._buckets) and self._buckets[self._next].ready >= len(self._buckets[self._next].params):
                self._launch(self._buckets[self._next])
        finally:
            self._in_hook = False

    def _flush(self) -> None:
        while self._next < len(self._buckets):
            self._launch(self._buckets[self._next])

    def reset(self) -> None:
        """Drop the state of a step that did not reach ``finish()`` (backward raised): waits for what is in flight."""
        for w in self._works:
            w.wait()
        if self._flat is not None and self._flat.is_cuda and self._side is not None:
            torch.cuda.current_stream(self._flat.device).wait_stream(self._side)
        for b in self._buckets:
            b.ready, b.launched = 0, False
        self._works, self._next = [], 0

    def _launch(self, b: _Bucket) -> None:
        flat = self._buffer()
        piece = flat[b.lo:b.hi]
        parts = [(p.grad.reshape(-1) if p.grad is not None else torch.zeros(p.numel(), dtype=flat.dtype, device=flat.device))
                 for p in b.params]
        torch.cat(parts, out=piece)                          # pack: one kernel per bucket, on the producing stream
        assert self._buckets[self._next] is b, "buckets leave in index order"
        b.launched = True
        self._next += 1
        collective = self._world() > 1 or (self.force_collective and dist.is_initialized())
        if self.trace is not None:
            self.trace.append((self._next - 1, self._in_hook, bool(collective and flat.is_cuda)))
        if collective:
            self.collectives_issued += 1
            if flat.is_cuda:
                if self._side is None:
                    self._side = torch.cuda.Stream(device=flat.device)
                ev = torch.cuda.Event()
                ev.record()
                with torch.cuda.stream(self._side):
                    self._side.wait_event(ev)
                    self._works.append(dist.all_reduce(piece, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            else:
                self._works.append(dist.all_reduce(piece, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def finish(self) -> None:
        """After backward: flush, wait, average, unpack into ``p.grad``.  Every rank must call it every step."""
        if not self.params:
            return
        self._flush()
        flat = self._buffer()
        self.finish_calls += 1
        timed = self.profile and flat.is_cuda
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        for w in self._works:
            w.wait()
        if flat.is_cuda and self._side is not None:
            torch.cuda.current_stream(flat.device).wait_stream(self._side)
        if timed:
            e1.record()
            self._wait_events.append((e0, e1))
        world = self._world()
        if self.average and world > 1:
            flat.div_(world)
        have, views = [], []
        for b in self._buckets:
            off = b.lo
            for p in b.params:
                v = flat[off:off + p.numel()].view_as(p)
                off += p.numel()
                if p.grad is None:
                    p.grad = v.clone()
                else:
                    have.append(p.grad)
                    views.append(v)
            b.ready, b.launched = 0, False
        if have:
            torch._foreach_copy_(have, views)
        self._works, self._next = [], 0

    def describe(self) -> dict:
        """What one step sends: number of collectives (= buckets), bytes per collective, total bytes."""
        return {"collectives_per_step": len(self._buckets), "bucket_bytes_target": self.bucket_bytes,
                "bytes_per_collective": [4 * (b.hi - b.lo) for b in self._buckets], "bytes_per_step": 4 * self._total}

    def collective_wait_ms(self, reset: bool = True) -> Optional[float]:
        """Mean time per finish() the compute stream spent behind the collectives (``profile = True``; synchronises)."""
        if not self._wait_events:
            return None
        torch.cuda.synchronize()
        ms = sum(a.elapsed_time(b) for a, b in self._wait_events) / len(self._wait_events)
        if reset:
            self._wait_events = []
        return ms

    # ---- hook-less form (kept for callers that reduce after backward has returned) ---------------------------------
    def allreduce(self, async_op: bool = False):
        """Pack -> all_reduce(SUM) -> (divide) -> unpack into ``p.grad`` for every bucket now.  Parameters without a
        gradient on this rank contribute zeros (every rank must call this with the same parameter list)."""
        if not self.params:
            return None
        self._flush()
        if async_op:
            return _Pending(self)
        self.finish()
        return None


class _Pending:
    def __init__(self, owner: GradientAllReducer):
        self.owner = owner

    def wait(self) -> None:
        self.owner.finish()


def broadcast_parameters(module: torch.nn.Module, src: int = 0, group=None) -> None:
    """Make every rank start from rank ``src``'s parameters and buffers (replaces DataParallel's per-step replicate)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src=src, group=group)
