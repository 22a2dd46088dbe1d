"""Deterministic inputs / weights shared by the golden generator and the tests.

Everything is drawn from ``np.random.RandomState`` (frozen stream), never from
torch's RNG, so fixtures can store seeds instead of tensors."""
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


def gcn_known_answer():
    """A literal GCNConv known-answer case (PyG 2.0.2 gcn_norm semantics as cited at reference models.py:330-332), worked
    out BY HAND — the numbers below are written out, not computed by any implementation under test.

    6 nodes; directed edges (source -> target):  0->1, 1->0, 1->2, 2->1, 2->2 (an existing self loop: dropped, then every
    node gets exactly one), 0->1 AGAIN (a duplicate: counted twice), 3->4 (one direction only); node 5 is isolated.

    in-degree by target + 1 self loop:  [2, 4, 2, 1, 2, 1]   ->  d^-1/2 = [1/sqrt2, 1/2, 1/sqrt2, 1, 1/sqrt2, 1]
    A_hat (rows = target):   row 0: [1/2, 1/(2 sqrt2), 0, 0, 0, 0]
                             row 1: [2 * 1/(2 sqrt2) = 1/sqrt2, 1/4, 1/(2 sqrt2), 0, 0, 0]
                             row 2: [0, 1/(2 sqrt2), 1/2, 0, 0, 0]
                             row 3: [0, 0, 0, 1, 0, 0]
                             row 4: [0, 0, 0, 1/sqrt2, 1/2, 0]
                             row 5: [0, 0, 0, 0, 0, 1]
    x[n, 0] = n + 1, x[n, 1] = -(n + 1) / 2, other channels 0;  W = 2 I with W[0, 1] = 1;  bias[0] = 0.5
    h = x W^T:  h[:, 0] = 2 x0 + x1 = 1.5 (n + 1) = [1.5, 3, 4.5, 6, 7.5, 9],  h[:, 1] = 2 x1 = -(n + 1)
    out = A_hat h + bias."""
    ei = torch.tensor([[0, 1, 1, 2, 2, 0, 3], [1, 0, 2, 1, 2, 1, 4]], dtype=torch.int64)
    x = torch.zeros(6, 128)
    x[:, 0] = torch.arange(1, 7, dtype=torch.float32)
    x[:, 1] = -torch.arange(1, 7, dtype=torch.float32) / 2
    w = 2 * torch.eye(128)
    w[0, 1] = 1.0
    b = torch.zeros(128)
    b[0] = 0.5
    out = torch.zeros(6, 128, dtype=torch.float64)
    # channel 0: 0.5*1.5 + 0.35355339*3 | 0.25*3 + 0.70710678*1.5 + 0.35355339*4.5 | 0.5*4.5 + 0.35355339*3 | 6 | 0.5*7.5 + 0.70710678*6 | 9, + 0.5
    out[:, 0] = torch.tensor([2.31066017, 3.90165043, 3.81066017, 6.5, 8.49264069, 9.5], dtype=torch.float64)
    # channel 1 = -(2/3) * (channel 0 - 0.5)
    out[:, 1] = torch.tensor([-1.20710678, -2.26776695, -2.20710678, -4.0, -5.32842712, -6.0], dtype=torch.float64)
    deg_inv_sqrt = torch.tensor([0.70710678, 0.5, 0.70710678, 1.0, 0.70710678, 1.0], dtype=torch.float64)
    return ei, x, w, b, out, deg_inv_sqrt
