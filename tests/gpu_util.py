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
