"""Deterministic inputs / weights shared by the golden generator and the tests: the generators live in
echoglad_amd/synthetic.py (bench.py and the smoke run use them too and must not depend on the test tree); this module
re-exports them and holds the hand-computed known-answer case."""
from __future__ import annotations

import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


from echoglad_amd.synthetic import (AVERAGE_COORDS, fill_state_dict, initial_coords, synthetic_frames,  # noqa: E402,F401
                                    synthetic_node_feats)


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
