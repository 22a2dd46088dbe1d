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


def gcn_closed_form_families(n: int, seed: int = 0):
    """Graph families on which GCNConv (PyG 2.0.2 gcn_norm: one self loop per node, degree by target, d^-1/2 on both ends) has a
    CLOSED FORM that can be written down without running any aggregation code -- further anchors of the operator's semantics
    beside the hand-computed 6-node case above (PyG's own source is absent from /root/reference).  -> [(name, edge_index,
    A_hat as a function (x fp64 [n, c]) -> fp64 [n, c])]:
      complete graph K_n (both directions of every pair): deg + 1 = n everywhere       -> every row = column mean of x
      cycle C_n (i <-> i + 1 mod n): deg + 1 = 3                                       -> (x[i-1] + x[i] + x[i+1]) / 3
      star S_n (centre 0 <-> leaves): deg + 1 = n at the centre, 2 at a leaf           -> centre: x0 / n + sum(leaves) / sqrt(2 n),
                                                                                           leaf j: x_j / 2 + x0 / sqrt(2 n)"""
    import math
    idx = torch.arange(n, dtype=torch.int64)
    src, dst = torch.meshgrid(idx, idx, indexing="ij")
    off = src != dst
    complete = torch.stack([src[off], dst[off]])
    nxt = (idx + 1) % n
    cycle = torch.stack([torch.cat([idx, nxt]), torch.cat([nxt, idx])])
    leaves = idx[1:]
    zeros = torch.zeros_like(leaves)
    star = torch.stack([torch.cat([zeros, leaves]), torch.cat([leaves, zeros])])
    g = torch.Generator().manual_seed(seed)                       # the edge ORDER must not matter: shuffle every list
    out = []

    def shuffled(ei):
        return ei[:, torch.randperm(ei.shape[1], generator=g)]

    out.append(("complete", shuffled(complete), lambda x: x.mean(dim=0, keepdim=True).expand_as(x).clone()))
    out.append(("cycle", shuffled(cycle), lambda x: (torch.roll(x, 1, 0) + x + torch.roll(x, -1, 0)) / 3.0))

    def star_fn(x):
        y = x / 2.0 + x[0:1] / math.sqrt(2.0 * n)
        y[0] = x[0] / n + x[1:].sum(dim=0) / math.sqrt(2.0 * n)
        return y
    out.append(("star", shuffled(star), star_fn))
    return out
