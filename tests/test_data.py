"""CPU: dataset / label / collate host logic (echoglad_amd/data.py) against fixtures from the reference's DummyDataset."""
import os

import numpy as np
import pytest
import torch

from echoglad_amd import data
from echoglad_amd.topology import HierTopology, TopologySpec


@pytest.mark.parametrize("frame,naux,main_only", [(16, 3, False), (30, 3, False), (64, 2, False), (224, 7, False), (16, 2, True)])
def test_node_labels_match_reference(golden_dir, frame, naux, main_only):
    d = np.load(os.path.join(golden_dir, "labels.npz"))
    key = f"F{frame}_A{naux}_mo{int(main_only)}"
    for c, ones in zip(d[key + "_coords"], d[key + "_ones"]):
        y = data.node_labels(c, frame, naux, main_only)
        assert len(y) == int(d[key + "_len"]) and y.dtype == np.float32
        assert np.array_equal(np.nonzero(y)[0], ones), (c, np.nonzero(y)[0], ones)
        assert set(np.unique(y)) == {0.0, 1.0}


def test_node_labels_out_of_range_raises_like_numpy():
    with pytest.raises(IndexError):
        data.node_labels([16, 0], 16, 3)           # digitize -> bin p: out of bounds in the reference too
    with pytest.raises(IndexError):
        data.node_labels([0, -17], 16, 2, True)


def test_full_sample_matches_reference_under_the_same_seed(golden_dir):
    d = np.load(os.path.join(golden_dir, "labels.npz"))
    ds = data.SyntheticEchoDataset(num_aux_graphs=3, frame_size=16, use_coordinate_graph=True)
    np.random.seed(78)
    assert np.array_equal(np.random.randint(low=0, high=16, size=12), d["sample_draws"])
    np.random.seed(78)
    g = ds[0]
    assert np.array_equal(g.y.numpy(), d["sample_y"])
    assert np.array_equal(g.valid_labels.numpy(), d["sample_valid"])
    assert np.array_equal(g.node_type.numpy(), d["sample_node_type"]) and g.node_type.dtype == torch.float64
    assert np.allclose(g.node_coords.numpy(), d["sample_node_coords"])
    assert np.array_equal(g.node_coord_y.numpy(), d["sample_node_coord_y"])
    assert np.allclose([float(g.pix2mm_x), float(g.pix2mm_y)], d["sample_pix2mm"])
    assert list(g.x.shape) == list(d["sample_x_shape"])
    assert len(ds) == 100


@pytest.mark.parametrize("coord", [False, True])
def test_collate_is_a_disjoint_union(coord):
    ds = data.SyntheticEchoDataset(num_aux_graphs=2, frame_size=8, use_coordinate_graph=coord)
    samples = [ds[i] for i in range(3)]
    b1, b2 = data.collate(samples), data.collate(samples, ds.topology)
    n = ds.topology.num_nodes
    assert b1.x.shape == (3, 1, 8, 8) and b1.y.shape == (3 * (4 + 16 + 64), 4) and b1.node_type.shape == (3 * n,)
    assert torch.equal(b1.batch, torch.arange(3).repeat_interleave(n))
    # same edge multiset either way, every edge inside its own frame
    def canon(ei):
        return ei[:, np.lexsort((ei[1].numpy(), ei[0].numpy()))]
    assert torch.equal(canon(b1.edge_index), canon(b2.edge_index))
    assert bool(((b1.edge_index[0] // n) == (b1.edge_index[1] // n)).all())
    assert b1.pix2mm_x.shape == (3,)
    if coord:
        assert b1.node_coords.shape == (12, 2) and b1.node_coord_y.shape == (12, 2)
    else:
        assert not hasattr(b1, "node_coords")


def test_unet_variant_example_registers_the_references_state_dict(golden_dir):
    """echoglad_amd.examples.UNetNodeFeatureModel (INTEGRATION.md route A for the class configs/default.yml names) has the
    state_dict of the reference's UNETHierarchicalPatchModel -- every key, every shape (tests/golden/unet_state_keys.json, dumped
    from the reference class by make_golden.py) -- so a reference checkpoint loads strict=True (checkpointers.py:94-98)."""
    import json
    import os
    from echoglad_amd.examples import UNetNodeFeatureModel
    ref = json.load(open(os.path.join(golden_dir, "unet_state_keys.json")))
    m = UNetNodeFeatureModel(frame_size=224, num_aux_graphs=7, node_embedding_dim=128, node_hidden_dim=128, classifier_hidden_dim=32,
                             num_gnn_layers=3, output_activation="logit", use_coordinate_graph=True, gnn_dropout_p=0.5,
                             classifier_dropout_p=0.5)
    mine = {k: list(v.shape) for k, v in m.state_dict().items()}
    assert mine == ref["state_dict"]
    assert sum(p.numel() for p in m.parameters()) == ref["n_parameters"] == 8089074
    # the decoder's maps are the graph's levels, coarse to fine (CPU: stock torch modules, no HIP involved)
    import torch
    with torch.no_grad():
        maps = m.eval().decoder_maps(torch.zeros(1, 4, 224, 224))
    assert [tuple(t.shape[1:]) for t in maps] == [(512, 2, 2), (256, 4, 4), (128, 8, 8), (64, 16, 16), (32, 32, 32), (16, 64, 64),
                                                  (8, 128, 128), (4, 224, 224)]


def test_copy_batch_writes_into_the_static_tensors():
    """data.copy_batch_: a new collated batch arrives IN PLACE (what a captured training step reads), shapes must agree."""
    np.random.seed(3)
    ds = data.SyntheticEchoDataset(num_aux_graphs=3, frame_size=16, use_coordinate_graph=True)
    a = data.collate([ds[0], ds[1]], ds.topology)
    b = data.collate([ds[2], ds[3]], ds.topology)
    ptrs = {k: v.data_ptr() for k, v in vars(a).items() if torch.is_tensor(v)}
    assert not torch.equal(a.x, b.x)
    out = data.copy_batch_(a, b)
    assert out is a and all(getattr(a, k).data_ptr() == p for k, p in ptrs.items())
    for k, v in vars(b).items():
        if torch.is_tensor(v):
            assert torch.equal(getattr(a, k), v), k
    c = data.collate([ds[0]], ds.topology)
    with pytest.raises(ValueError):
        data.copy_batch_(a, c)


def test_copy_batch_never_copies_the_graph_tensors():
    """A captured step runs on the topology handle it was captured with: edge_index / batch / node_type are not copied (a
    pageable host-to-device copy of 7 - 55 MB per step otherwise, and a version bump that defeats the identity resolution) --
    the collate() constant the static tensor was moved from is recognised by identity, a fresh equal tensor is compared once, a
    different graph is refused instead of silently running on the old one."""
    np.random.seed(5)
    ds = data.SyntheticEchoDataset(num_aux_graphs=3, frame_size=16)
    static = data.collate([ds[0], ds[1]], ds.topology)
    static.edge_index, static.batch, static.node_type = static.edge_index.clone(), static.batch.clone(), static.node_type.clone()
    v0 = (static.edge_index._version, static.batch._version, static.node_type._version)
    calls = []
    orig = torch.equal
    try:
        torch.equal = lambda a, b: (calls.append(1), orig(a, b))[1]
        for k in (2, 4):
            data.copy_batch_(static, data.collate([ds[k], ds[k + 1]]))              # fresh (equal) graph tensors every time
    finally:
        torch.equal = orig
    assert len(calls) == 3                                                            # compared once per attribute, not per step
    assert (static.edge_index._version, static.batch._version, static.node_type._version) == v0
    other = data.collate([ds[0], ds[1]])
    other.edge_index = other.edge_index.flip(0)
    fresh = data.collate([ds[0], ds[1]], ds.topology)
    fresh.edge_index, fresh.batch, fresh.node_type = fresh.edge_index.clone(), fresh.batch.clone(), fresh.node_type.clone()
    with pytest.raises(ValueError, match="ONE\\s+topology"):
        data.copy_batch_(fresh, other)


def test_to_device_caches_only_what_collate_registered():
    np.random.seed(6)
    ds = data.SyntheticEchoDataset(num_aux_graphs=3, frame_size=16)
    before = len(data._CONST_ON_DEVICE)
    plain = data.collate([ds[0], ds[1]])                                              # fresh graph tensors: never cached
    data.to_device(plain, "meta")
    assert len(data._CONST_ON_DEVICE) == before and plain.edge_index.device.type == "meta"
    const = data.collate([ds[0], ds[1]], ds.topology)
    cpu_ei = const.edge_index
    data.to_device(const, "meta")
    again = data.to_device(data.collate([ds[2], ds[3]], ds.topology), "meta")
    assert len(data._CONST_ON_DEVICE) == before + 3 and again.edge_index is const.edge_index and data._is_collate_const(cpu_ei)
    for k in [k for k in data._CONST_ON_DEVICE if k[1] == "meta"]:
        del data._CONST_ON_DEVICE[k]


def test_collate_hands_out_the_constant_tensors_of_a_topology_again():
    """collate(..., topology): edge_index / batch / node_type depend on (topology, batch size) only -- built once, the SAME tensors
    for every later batch (the model resolves a known edge_index by identity; to_device moves it once per device), equal to what a
    fresh build gives; another batch size gets its own."""
    np.random.seed(4)
    ds = data.SyntheticEchoDataset(num_aux_graphs=3, frame_size=16)
    a = data.collate([ds[0], ds[1]], ds.topology)
    b = data.collate([ds[2], ds[3]], ds.topology)
    assert a.edge_index is b.edge_index and a.batch is b.batch and a.node_type is b.node_type
    n = ds.topology.num_nodes
    want = torch.cat([ds.edge_index, ds.edge_index + n], dim=1)
    assert torch.equal(a.edge_index, want) and torch.equal(a.batch, torch.arange(2).repeat_interleave(n))
    assert torch.equal(a.node_type, torch.cat([ds.node_type, ds.node_type]))
    c = data.collate([ds[0], ds[1], ds[2]], ds.topology)
    assert c.edge_index is not a.edge_index and c.edge_index.shape[1] == 3 * ds.edge_index.shape[1]
    assert ds.topology.batched_edge_index(2) is ds.topology.batched_edge_index(2)
    plain = data.collate([ds[0], ds[1]])                                   # without the topology: PyG's shifted copies, same values
    assert torch.equal(plain.edge_index, a.edge_index) and torch.equal(plain.node_type, a.node_type)
    moved = data.to_device(data.collate([ds[0], ds[1]], ds.topology), "cpu")
    assert moved.edge_index is a.edge_index                                # (already there: nothing to move)
