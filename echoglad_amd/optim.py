"""Adam as ONE launch per step (``eg_adam_step``, csrc/adam.hip) -- the optimizer update of the training step.

The reference trains with ``torch.optim.Adam`` (src/engine.py's optimizer builder).  torch's fused form is a ``_foreach_add_`` on
the step counts plus one multi-tensor launch per ~30 tensors: 4 launches (34 us of a 0.9-ms captured batch-1 step) for the 73
parameter tensors of the GNN stack and heads.  ``Adam`` below runs the same arithmetic (torch's fused Adam, amsgrad off) in one launch
per 96 tensors; the step counts are device floats (one per parameter, as in torch), so the update can be captured into a HIP graph
(``engine.GraphedTrainStep`` asks for ``capturable``: this optimizer always is).  State layout as torch's (``state[p]`` = ``step``,
``exp_avg``, ``exp_avg_sq``), so ``state_dict()`` / ``load_state_dict()`` round-trip and a ``torch.optim.Adam`` state loads.

``lr`` may be a CUDA float32 scalar tensor (torch's convention for captured steps: the kernel reads it, so a scheduler's ``fill_`` reaches the
replays); a float lr is frozen into a captured launch.  CUDA float32 parameters with dense gradients only; anything else raises (there
is no CPU path)."""
from __future__ import annotations

import ctypes as ct
from typing import Iterable

import torch

from . import _lib

_MAX_TENSORS = 96


class _AdamTensor(ct.Structure):
    """eg_adam_tensor (include/echoglad_hip.h)"""
    _fields_ = [("param", ct.c_void_p), ("grad", ct.c_void_p), ("exp_avg", ct.c_void_p), ("exp_avg_sq", ct.c_void_p), ("numel", ct.c_int64)]


class Adam(torch.optim.Optimizer):
    def __init__(self, params: Iterable, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0,
                 maximize: bool = False):
        if not torch.is_tensor(lr) and not 0.0 <= lr:
            raise ValueError(f"Invalid learning rate: {lr}")
        if not 0.0 <= eps:
            raise ValueError(f"Invalid epsilon value: {eps}")
        if not (0.0 <= betas[0] < 1.0 and 0.0 <= betas[1] < 1.0):
            raise ValueError(f"Invalid betas: {betas}")
        if not 0.0 <= weight_decay:
            raise ValueError(f"Invalid weight_decay value: {weight_decay}")
        # (capturable / fused: what engine.GraphedTrainStep and code written for torch.optim.Adam look for)
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, maximize=maximize, capturable=True, fused=True)
        super().__init__(params, defaults)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib, stream = None, None
        for gi, group in enumerate(self.param_groups):
            ps = [p for p in group["params"] if p.grad is not None]
            if not ps:
                continue
            if not ps[0].is_cuda:
                raise RuntimeError("echoglad_amd.optim.Adam: CUDA float32 parameters with dense float32 gradients only")
            if lib is None:
                lib = _lib.load()
                stream = ct.c_void_p(torch.cuda.current_stream().cuda_stream)
            # the prepared launches of this group (pointer tables, step-count arrays) are kept while nothing moved: the same parameters with
            # the same addresses and gradients at the same addresses (the caching allocator hands a step's gradients the same blocks
            # again) -- checks and 82 ctypes structs per step were 0.2 ms of a batch-1 step the host is the bound of
            key = (tuple(map(id, ps)), tuple(p.data_ptr() for p in ps), tuple(p.grad.data_ptr() for p in ps))
            cache = self.__dict__.setdefault("_prepared", {}).setdefault(gi, {})
            hit = cache.get(key)
            if hit is None:                             # (the allocator cycles through a few sets of blocks for a step's gradients)
                if len(cache) >= 4:
                    cache.clear()
                hit = cache[key] = (key, self._prepare(ps))
            b1, b2 = group["betas"]
            lr = group["lr"]
            lr_dev = None
            if torch.is_tensor(lr):                     # (torch's convention for a captured step: schedulers fill_() a tensor lr)
                if not lr.is_cuda or lr.dtype != torch.float32 or lr.numel() != 1:
                    raise RuntimeError("echoglad_amd.optim.Adam: a tensor lr must be a CUDA float32 scalar")
                lr_dev, lr = ct.c_void_p(lr.data_ptr()), 0.0
            for part, table, counts, _keep in hit[1]:
                _lib.check(lib.eg_adam_step(table, len(part), ct.c_void_p(counts.data_ptr()), float(lr), lr_dev, float(b1), float(b2),
                                            float(group["eps"]), float(group["weight_decay"]), int(bool(group["maximize"])), stream),
                           "eg_adam_step")
                # the kernel wrote the parameters behind autograd's back: tell it (in-place version counters: what saved-tensor checks
                # and the model's caches of folded inference parameters are keyed on)
                torch.autograd.graph.increment_version(part)
        return loss

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self.__dict__.pop("_prepared", None)            # (the moments and counts are new tensors)
        self.__dict__.pop("_count_arrays", None)

    def _prepare(self, ps):
        """[(parameters, eg_adam_tensor table, step counts, tensors kept alive)] -- one entry per launch of up to 96 parameters."""
        for p in ps:
            if not p.is_cuda or p.dtype != torch.float32 or p.grad.is_sparse or p.grad.dtype != torch.float32:
                raise RuntimeError("echoglad_amd.optim.Adam: CUDA float32 parameters with dense float32 gradients only")
            if not p.is_contiguous():
                raise RuntimeError("echoglad_amd.optim.Adam: parameters must be contiguous")
            if not p.grad.is_contiguous():
                p.grad = p.grad.contiguous()            # (its address is not the key's: prepared again next step -- correct, just not cached)
            st = self.state[p]
            if "exp_avg" not in st:
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
        out = []
        for lo in range(0, len(ps), _MAX_TENSORS):
            part = ps[lo:lo + _MAX_TENSORS]
            counts = self._counts_of(part)
            table = (_AdamTensor * len(part))()
            keep = []
            for k, p in enumerate(part):
                st = self.state[p]
                table[k] = _AdamTensor(p.data_ptr(), p.grad.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), p.numel())
                keep += [st["exp_avg"], st["exp_avg_sq"]]
            out.append((part, table, counts, keep))
        return out

    def _counts_of(self, part):
        """The step counts of the parameters of one launch as ONE device array (the kernel takes steps[k]); ``state[p]["step"]`` are 0-d
        views of it.  Kept while the same parameters come in the same order; otherwise (first step, a parameter that had no gradient
        last time, a loaded state) re-assembled from the per-parameter counts -- torch's state layout stays the truth."""
        key = tuple(id(p) for p in part)
        hit = self.__dict__.setdefault("_count_arrays", {}).get(key)
        if hit is not None and all(self.state[p].get("step") is v for p, v in zip(part, hit[1])):
            return hit[0]
        dev = part[0].device
        vals = []
        for p in part:
            t = self.state[p].get("step")
            vals.append(torch.zeros((), dtype=torch.float32, device=dev) if t is None else
                        (t.detach().to(device=dev, dtype=torch.float32).reshape(()) if torch.is_tensor(t) else torch.tensor(float(t), dtype=torch.float32, device=dev)))
        counts = torch.stack(vals)
        views = [counts[k] for k in range(len(part))]
        for p, v in zip(part, views):
            self.state[p]["step"] = v
        if len(self._count_arrays) > 8:
            self._count_arrays.clear()
        self._count_arrays[key] = (counts, views)
        return counts
