"""N>1 path on CPU: world_size-2 gloo run of the batch sharding + flat gradient all-reduce
(echoglad_amd/parallel.py) against a single-process run of the full batch."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from fixtures_util import fill_state_dict, synthetic_node_feats
from oracle import gnn_oracle as O
from echoglad_amd import parallel
from echoglad_amd.topology import HierTopology, TopologySpec


def test_shard_range_is_a_partition():
    for n in (1, 7, 8, 64, 257):
        for world in (1, 2, 3, 8):
            spans = [parallel.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def test_shard_frames_views():
    N = 84
    feats = torch.arange(4 * N * 2, dtype=torch.float32).view(4 * N, 2)
    coords = torch.arange(16 * 2, dtype=torch.float32).view(16, 2)
    s = parallel.shard_frames(1, 2, N, node_feats=feats, node_coords=coords, labels=feats[:, :1])
    assert s["frame_range"] == (2, 4)
    assert torch.equal(s["node_feats"], feats[2 * N:]) and torch.equal(s["node_coords"], coords[8:])
    assert s["labels"].shape[0] == 2 * N


def _model(frame, naux):
    m = O.OracleHierarchicalPatchModel(frame_size=frame, node_embedding_dim=128, node_hidden_dim=128,
                                       num_gnn_layers=2, num_aux_graphs=naux, classifier_hidden_dim=32,
                                       output_activation="logit")
    fill_state_dict(m, 3)
    return m.eval()          # eval-mode BN so that shard-wise and full-batch losses are identical functions


def _worker(rank, world, port, frame, naux, B, out_dir, mode="after"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    topo = HierTopology(TopologySpec(frame, naux))
    n = topo.num_nodes
    feats = synthetic_node_feats(B * n, 128, seed=5)
    shard = parallel.shard_frames(rank, world, n, node_feats=feats)
    lo, hi = shard["frame_range"]
    b = hi - lo
    model = _model(frame, naux)
    if rank != 0:                      # ranks start different; broadcast must fix that
        for p in model.parameters():
            p.data.add_(1.0)
    parallel.broadcast_parameters(model, src=0)
    ei = torch.from_numpy(topo.batched_edge_index(b))
    nt = torch.from_numpy(np.tile(topo.node_type(), b))
    if mode == "missing" and rank == 1:
        # withheld BEFORE the forward builds the graph: no AccumulateGrad node, the hook never fires on this rank, so the
        # bucket that holds this parameter completes inside backward on rank 0 and only in finish() on rank 1
        model.node_classifiers[3][8].bias.requires_grad_(False)
    logits, _ = model.forward_nodes(shard["node_feats"], ei, nt, b)
    loss = (logits ** 2).sum() / (B * n)            # global mean written as a sum of shard sums
    if mode == "missing":
        # (the parameter list must be the same on both ranks: the flag is re-enabled for the list, not for the graph)
        model.node_classifiers[3][8].bias.requires_grad_(True)
        red = parallel.GradientAllReducer(model.parameters(), average=False, bucket_bytes=16 << 10).attach_hooks()
        fired = []
        orig = red._launch
        red._launch = lambda bk: (fired.append(bk), orig(bk))[1]
        loss.backward()
        in_backward = len(fired)
        red.finish()
        order = [red._buckets.index(bk) for bk in fired]
        assert order == list(range(len(red._buckets))), order          # strictly in index order on every rank
        if rank == 0:
            assert in_backward >= len(red._buckets) - 1
        else:
            assert in_backward == 0                                     # bucket 0 never completed: everything waits for finish()
    elif mode == "after":                             # reduce once backward has returned
        loss.backward()
        red = parallel.GradientAllReducer(model.parameters(), average=False)
        pending = red.allreduce(async_op=True)
        pending.wait()
    else:                                           # overlapped: buckets fire from post-accumulate-grad hooks inside backward
        red = parallel.GradientAllReducer(model.parameters(), average=False, bucket_bytes=16 << 10).attach_hooks()
        assert len(red._buckets) >= 4
        fired = []
        orig = red._launch
        red._launch = lambda b: (fired.append(b), orig(b))[1]
        loss.backward()
        n_in_backward = len(fired)
        red.finish()
        assert n_in_backward >= len(red._buckets) - 1 and len(fired) == len(red._buckets)
        # the diagnostics bench.py --mode train reports at world > 1 (what one step sends; time behind the collectives)
        d = red.describe()
        assert d["collectives_per_step"] == len(red._buckets) == len(d["bytes_per_collective"]) and d["bucket_bytes_target"] == 16 << 10
        assert sum(d["bytes_per_collective"]) == d["bytes_per_step"] == 4 * sum(p.numel() for p in model.parameters())
        assert all(bb >= 16 << 10 for bb in d["bytes_per_collective"][:-1])
        assert red.finish_calls == 1 and red.collectives_issued == len(red._buckets)
        red.profile = True                              # CPU tensors: nothing to time, and nothing may break
        assert red.collective_wait_ms() is None
        # the classifier heads' bucket goes out first, the first GNN layer's last
        names = {id(p): k for k, p in model.named_parameters()}
        assert names[id(fired[0].params[0])].startswith("node_classifiers")
        assert any(names[id(p)].startswith("gnn_layers.0.") for p in fired[-1].params)
        # second step with the same reducer: counters were reset
        for p in model.parameters():
            p.grad = None
        logits, _ = model.forward_nodes(shard["node_feats"], ei, nt, b)
        ((logits ** 2).sum() / (B * n)).backward()
        red.finish()
    if rank == 0:
        torch.save({k: p.grad.clone() for k, p in model.named_parameters()}, os.path.join(out_dir, "grads.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["after", "overlapped", "missing"])
def test_two_rank_gradients_match_single_process(tmp_path, mode):
    frame, naux, B, world = 8, 2, 4, 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(world, port, frame, naux, B, str(tmp_path), mode), nprocs=world, join=True)
    got = torch.load(os.path.join(tmp_path, "grads.pt"))
    topo = HierTopology(TopologySpec(frame, naux))
    n = topo.num_nodes
    model = _model(frame, naux)
    feats = synthetic_node_feats(B * n, 128, seed=5)
    ei = torch.from_numpy(topo.batched_edge_index(B))
    nt = torch.from_numpy(np.tile(topo.node_type(), B))
    logits, _ = model.forward_nodes(feats, ei, nt, B)
    ((logits ** 2).sum() / (B * n)).backward()
    want = {k: p.grad.clone() for k, p in model.named_parameters()}
    if mode == "missing":
        # rank 1 contributed zeros for the withheld parameter: its reduced gradient is rank 0's shard alone
        # (round 2's reducer paired the wrong buckets here and returned silently wrong sums for EVERY bucket)
        half = B // 2
        for p in model.parameters():
            p.grad = None
        ei0 = torch.from_numpy(topo.batched_edge_index(half))
        logits, _ = model.forward_nodes(feats[:half * n], ei0, nt[:half * n], half)
        ((logits ** 2).sum() / (B * n)).backward()
        want["node_classifiers.3.8.bias"] = model.node_classifiers[3][8].bias.grad.clone()
    for k in want:
        assert torch.allclose(got[k], want[k], rtol=1e-4, atol=1e-6), k


def test_bench_starts_its_own_ranks_and_propagates_failure():
    """`python bench.py --gpus 2` without WORLD_SIZE starts 2 rank processes itself (before any GPU call in the parent).
    In this container there is no GPU, so both ranks refuse to run and the parent must exit non-zero.  With a GPU the same
    parent / child structure runs for real: tests/test_gpu_parallel.py::test_world_2_on_one_gpu_... (two ranks on one device over
    gloo) and ::test_world_2_parent_exits_with_the_worst_rank_code."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    if torch.cuda.is_available():
        pytest.skip("with a GPU: tests/test_gpu_parallel.py::test_world_2_* run the same parent with real ranks")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0
    assert r.stderr.count("needs a GPU") == 2, r.stderr
