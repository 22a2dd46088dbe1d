"""Helpers shared by the -m gpu parity tests."""
import numpy as np
import torch

from fixtures_util import fill_state_dict
from oracle import gnn_oracle as O
from echoglad_amd import nn as egnn
from echoglad_amd.topology import HierTopology, TopologySpec

DEV = "cuda:0"


def model_pair(frame, naux, layers, coord=False, main_only=False, seed=0, output_activation="logit", **extra):
    """(HIP model on the GPU, oracle model on the CPU) with identical trained-like weights."""
    kw = dict(frame_size=frame, gnn_dropout_p=0.5, classifier_dropout_p=0.5, node_embedding_dim=128,
              node_hidden_dim=128, num_output_channels=4, num_gnn_layers=layers, num_aux_graphs=naux,
              classifier_hidden_dim=32, use_coordinate_graph=coord, output_activation=output_activation,
              use_main_graph_only=main_only)
    kw.update(extra)
    ref = O.OracleHierarchicalPatchModel(**kw)
    fill_state_dict(ref, seed)
    hip = egnn.HierarchicalPatchModel(**kw)
    hip.load_state_dict(ref.state_dict(), strict=True)
    return hip.to(DEV).eval(), ref.eval()


def graph_tensors(frame, naux, batch, coord=False, main_only=False, conn=False, main_type="grid", aux_type="grid"):
    topo = HierTopology(TopologySpec(frame, naux, main_only, coord, conn, main_type, aux_type))
    ei = torch.from_numpy(topo.batched_edge_index(batch))
    nt = torch.from_numpy(np.tile(topo.node_type(), batch))
    bi = torch.arange(batch).repeat_interleave(topo.num_nodes)
    return topo, ei, nt, bi


def rand_rows(rows, seed, scale=1.0):
    rs = np.random.RandomState(seed)
    return torch.from_numpy((rs.standard_normal((rows, 128)) * scale).astype(np.float32))


def dense_ahat(topo: HierTopology) -> torch.Tensor:
    """fp64 D^-1/2 (A+I) D^-1/2 for one frame."""
    n = topo.num_nodes
    a = torch.zeros(n, n, dtype=torch.float64)
    e = topo.edge_index()
    a[e[1], e[0]] = 1.0
    a += torch.eye(n, dtype=torch.float64)
    dis = a.sum(1).pow(-0.5)
    return dis[:, None] * a * dis[None, :]


def assert_param_grads_close(hip, ref, tight=1e-3, loose=3e-2, flipped_channels=2):
    """Parameter gradients of a train step, HIP model against the CPU oracle (both after .backward()).  A train step is
    discontinuous where a pre-activation sits within rounding of the ReLU kink: the two evaluations may put it on different sides,
    which moves every gradient entry of that CHANNEL by a whole term -- measured against an fp64 run (tools/dbg_conn.py)
    either side can be the one that is off, by up to 5.5e-3 of the largest entry, in ONE channel of 128, while every other
    channel agrees to 4e-5.  So: at most `flipped_channels` channels (rows of a weight, entries of a vector; 2 % of them if that is
    more) of a parameter may miss the tight tolerance, and no entry the loose one.  (A bias in front of a train-mode BatchNorm has
    an exactly-zero gradient: both sides hold rounding noise.)"""
    rg = dict(ref.named_parameters())
    gmax = max(float(q.grad.abs().max()) for q in rg.values() if q.grad is not None)
    for name, p in hip.named_parameters():
        want = rg[name].grad
        assert p.grad is not None and want is not None, name
        scale = float(want.abs().max())
        if scale < 1e-5 * gmax:                      # analytically zero (the oracle's entries are rounding noise, the kernels return zeros)
            assert float(p.grad.abs().max()) <= 1e-4 * gmax, name
            continue
        err = ((p.grad.detach().cpu() - want).abs() / scale).reshape(want.shape[0], -1).max(dim=1).values      # per channel
        assert float(err.max()) < loose, (name, float(err.max()))
        bad = int((err > tight).sum())
        assert bad <= max(flipped_channels, err.numel() // 50), (name, bad, err.numel(), float(err.max()))
