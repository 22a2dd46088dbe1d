"""Closed-form topology vs. the golden vectors captured from the reference builder
(reference src/core/datasets.py:1441-1584 run in-container by tests/golden/make_golden.py)."""
import json
import os

import numpy as np
import pytest

from echoglad_amd.topology import HierTopology, TopologySpec, commutative_edge_hash, get_topology


def _load(golden_dir):
    with open(os.path.join(golden_dir, "topology.json")) as f:
        return json.load(f)


def _spec_from_key(key):
    parts = key.split("_")
    F, A = int(parts[0][1:]), int(parts[1][1:])
    mo, co, cn = (int(p[2:]) for p in parts[2:5])
    return TopologySpec(F, A, bool(mo), bool(co), bool(cn), parts[5], parts[6])


def test_every_golden_config_matches(golden_dir):
    gold = _load(golden_dir)
    assert len(gold) >= 17
    for key, g in gold.items():
        topo = HierTopology(_spec_from_key(key))
        assert topo.num_nodes == g["num_nodes"], key
        assert topo.num_undirected_edges == g["num_undirected_edges"], key
        assert topo.edge_index().shape[1] == g["num_directed_edges"], key
        assert topo.edge_set_digest() == g["edge_sha256"], key
        assert {str(k): v for k, v in topo.degree_histogram().items()} == g["degree_hist"], key
        assert topo.num_valid_nodes == g["num_valid"], key


def test_default_config_numbers():
    """N = 72,020, E = 215,100 undirected, degree histogram at 224/7."""
    topo = get_topology(TopologySpec(224, 7))
    assert topo.num_nodes == 72020
    assert topo.num_undirected_edges == 215100
    assert topo.degree_histogram() == {3: 8, 4: 1392, 5: 52616, 6: 4, 7: 20, 8: 456, 9: 17524}
    assert [lv.base for lv in topo.aux_levels] == [0, 4, 20, 84, 340, 1364, 5460]
    assert topo.main.base == 21844
    assert topo.crop_rows[0] == 8 and len(topo.crop_rows) == 112


def test_full_edge_list_f8(golden_dir):
    und = np.load(os.path.join(golden_dir, "topo_f8_a2_edges.npy"))
    topo = HierTopology(TopologySpec(8, 2))
    e = topo.undirected_edges()
    mine = np.unique(np.stack([np.minimum(e[0], e[1]), np.maximum(e[0], e[1])], axis=1), axis=0)
    assert np.array_equal(mine, und)


def test_slice_quirks():
    # F=64 / naux=2: negative centre clamps to the whole 4x4 level, linked to the top-left 8x8 only
    t = HierTopology(TopologySpec(64, 2))
    assert t.crop_rows == [0, 1, 2, 3]
    # F=8 / naux=1: centre -1 wraps to the last row only
    t = HierTopology(TopologySpec(8, 1))
    assert t.crop_rows == [1]
    # F=448 / naux=7: only rows 80..127 link
    t = HierTopology(TopologySpec(448, 7))
    assert t.crop_rows[0] == 80 and t.crop_rows[-1] == 127


def test_edge_index_is_symmetric_sorted_and_loop_free():
    topo = HierTopology(TopologySpec(16, 3, use_coordinate_graph=True))
    ei = topo.edge_index()
    assert (ei[0] != ei[1]).all()
    fwd = set(map(tuple, ei.T.tolist()))
    assert all((b, a) in fwd for a, b in fwd)
    key = ei[0] * topo.num_nodes + ei[1]
    assert (np.diff(key) > 0).all()
    # coordinate nodes are an isolated K4
    cb = topo.coord_base
    touching = ei[:, (ei[0] >= cb) | (ei[1] >= cb)]
    assert touching.shape[1] == 12 and (touching >= cb).all()


def test_batched_edge_index_offsets():
    topo = HierTopology(TopologySpec(8, 2))
    e1 = topo.edge_index()
    e3 = topo.batched_edge_index(3)
    E = e1.shape[1]
    assert e3.shape == (2, 3 * E)
    for b in range(3):
        assert np.array_equal(e3[:, b * E:(b + 1) * E], e1 + b * topo.num_nodes)


def test_commutative_hash_is_order_independent_and_sensitive():
    topo = HierTopology(TopologySpec(16, 3))
    ei = topo.edge_index()
    perm = np.random.RandomState(0).permutation(ei.shape[1])
    assert commutative_edge_hash(ei) == commutative_edge_hash(ei[:, perm])
    bad = ei.copy()
    bad[1, 5] = (bad[1, 5] + 1) % topo.num_nodes
    assert commutative_edge_hash(ei) != commutative_edge_hash(bad)
    assert commutative_edge_hash(ei) != commutative_edge_hash(ei[::-1][:, :-1])


def test_deg_inv_sqrt_matches_degree():
    topo = HierTopology(TopologySpec(32, 4, use_coordinate_graph=True))
    d = topo.degree()
    assert np.allclose(topo.deg_inv_sqrt(), 1.0 / np.sqrt(d + 1.0), rtol=1e-7)
    assert (d[topo.coord_base:] == 3).all()


def test_structured_flag():
    assert HierTopology(TopologySpec(16, 3)).is_structured()
    assert HierTopology(TopologySpec(16, 3, use_connection_nodes=True)).is_structured()                 # connection nodes: pre-pass + stencil (round 4)
    assert HierTopology(TopologySpec(16, 3, main_graph_type="grid-diagonal")).is_structured()          # 8-neighbour stencil (round 4)
    assert HierTopology(TopologySpec(16, 3, aux_graph_type="grid-diagonal")).is_structured()
