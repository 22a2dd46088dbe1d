"""Deterministic synthetic inputs and weights for benchmarks, smoke runs and tests (the DummyDataset side of the reference,
src/core/datasets.py:1340-1439): everything is drawn from ``np.random.RandomState`` (a frozen stream), never from torch's
RNG, so a workload is fully described by its seeds."""
from __future__ import annotations

import numpy as np
import torch


def fill_state_dict(module: torch.nn.Module, seed: int, trained_like: bool = True) -> None:
    """Overwrite every parameter/buffer in state_dict order from RandomState(seed).

    weights: glorot-like uniform; biases: small normal; BN running stats /
    affine: non-trivial ('trained-like') unless trained_like=False."""
    rs = np.random.RandomState(seed)
    sd = module.state_dict()
    new = {}
    for key, t in sd.items():
        shape = tuple(t.shape)
        if key.endswith("num_batches_tracked"):
            new[key] = torch.zeros_like(t)
            continue
        if key.endswith("running_mean"):
            v = rs.standard_normal(shape) * 0.3 if trained_like else np.zeros(shape)
        elif key.endswith("running_var"):
            v = rs.uniform(0.25, 1.75, shape) if trained_like else np.ones(shape)
        elif t.dim() >= 2:
            fan_out, fan_in = shape[0], int(np.prod(shape[1:]))
            a = np.sqrt(6.0 / (fan_in + fan_out))
            v = rs.uniform(-a, a, shape)
        elif key.endswith("weight"):          # BN gamma
            v = 1.0 + 0.3 * rs.standard_normal(shape) if trained_like else np.ones(shape)
        else:                                  # biases, BN beta
            v = 0.1 * rs.standard_normal(shape) if trained_like else np.zeros(shape)
        new[key] = torch.from_numpy(np.asarray(v, dtype=np.float32)).reshape(shape)
    module.load_state_dict(new, strict=True)


def synthetic_frames(batch: int, channels: int, size: int, seed: int) -> torch.Tensor:
    """N(0,1) frames like DummyDataset (reference src/core/datasets.py:1385), from RandomState."""
    rs = np.random.RandomState(seed)
    return torch.from_numpy(rs.standard_normal((batch, channels, size, size)).astype(np.float32))


def synthetic_node_feats(rows: int, channels: int, seed: int) -> torch.Tensor:
    rs = np.random.RandomState(seed)
    return torch.from_numpy(rs.standard_normal((rows, channels)).astype(np.float32))


AVERAGE_COORDS = [[99.99, 112.57], [142.71, 90.67], [151.18, 86.25], [91.81, 117.91]]  # datasets.py:1361


def initial_coords(batch: int, frame_size: int) -> torch.Tensor:
    """The fixed average coords (224-px units regardless of F in the reference); for small
    parity frames they are rescaled so they land inside the frame."""
    c = torch.tensor(AVERAGE_COORDS, dtype=torch.float32) * (frame_size / 224.0)
    return c.repeat(batch, 1)
