"""-m gpu: the GPU branch of the gradient reducer (side stream + event + asynchronous RCCL all-reduce, echoglad_amd/parallel.py)
on ONE GPU: a one-rank "nccl" process group with ``force_collective=True`` runs exactly the calls the multi-GPU step makes
(replaces the reference's DataParallel gradient reduce, src/engine.py:105-110).  Each case runs in a child process: a process
group is process-global state the other tests must not see."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _child_env():
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    env["MASTER_ADDR"] = "127.0.0.1"
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return env


def test_reducer_hooks_on_rccl_world_1_leave_the_gradients_bit_identical():
    r = subprocess.run([sys.executable, os.path.abspath(__file__)], capture_output=True, text=True, timeout=600, env=_child_env())
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["backend"] == "nccl" and d["world_size"] == 1
    assert d["collectives_issued"] == d["buckets"] >= 2 and d["fired_inside_backward"] >= d["buckets"] - 1
    assert d["bit_identical"] is True and d["params_compared"] > 40 and d["side_stream_used"] is True


def test_bench_train_step_with_the_reducer_attached_on_one_gpu():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--mode", "train", "--batch", "4", "--steps", "3", "--warmup", "2",
                        "--force-collective", "--bucket-kb", "64"], capture_output=True, text=True, timeout=900, env=_child_env())
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    dist = d["distributed"]
    assert dist["backend"] == "nccl" and dist["world_size"] == 1 and dist["allreduce_check"] is True
    # 2 warm-up + 3 timed steps + 3 steps on this rank's own clock (the per-rank diagnostics)
    assert dist["gradient_buckets"] >= 1 and dist["gradient_collectives_issued"] == dist["gradient_buckets"] * 8
    # what the first multi-GPU run will be read by: collectives per step and their bytes, per-rank step time, time behind the collectives
    gc = dist["gradient_collectives"]
    assert gc["collectives_per_step"] == dist["gradient_buckets"] == len(gc["bytes_per_collective"]) and sum(gc["bytes_per_collective"]) == gc["bytes_per_step"] > 250_000
    pr = dist["per_rank"]
    assert len(pr) == 1 and pr[0]["rank"] == 0 and pr[0]["ms_per_step_alone"] > 0 and pr[0]["ms_in_finish_behind_collectives"] is not None
    assert 0 <= pr[0]["ms_in_finish_behind_collectives"] < pr[0]["ms_per_step_alone"]
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["final_loss"] == d["final_loss"]                                  # (not NaN)


def _main():
    import socket

    import torch
    import torch.distributed as dist

    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from fixtures_util import initial_coords, synthetic_node_feats
    from gpu_util import DEV, graph_tensors, model_pair
    from echoglad_amd.parallel import GradientAllReducer

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        os.environ["MASTER_PORT"] = str(s.getsockname()[1])
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(DEV))
    frame, naux, B = 32, 4, 3
    hip, _ = model_pair(frame, naux, 3, coord=True, seed=23)
    hip.train()
    topo, ei, nt, bi = graph_tensors(frame, naux, B, coord=True)
    x = synthetic_node_feats(B * topo.num_nodes, 128, seed=41).to(DEV)
    eid = ei.to(DEV)
    coords0 = initial_coords(B, frame).to(DEV)
    state = {k: v.clone() for k, v in hip.state_dict().items()}

    def step(reducer):
        hip.load_state_dict(state)
        for p in hip.parameters():
            p.grad = None
        torch.manual_seed(7)                                   # the dropout seeds come from the host RNG
        logits, coords = hip.forward_nodes(x, eid, B, coords0.clone())
        loss = (logits ** 2).mean() + (coords ** 2).mean() * 1e-3
        loss.backward()
        inside.append(len(fired))                              # buckets that went out from inside backward
        if reducer is not None:
            reducer.finish()
        torch.cuda.synchronize()
        return {k: p.grad.clone() for k, p in hip.named_parameters() if p.grad is not None}

    fired, inside = [], []
    plain = step(None)
    red = GradientAllReducer(hip.parameters(), bucket_bytes=32 << 10, force_collective=True).attach_hooks()
    orig = red._launch
    red._launch = lambda b: (fired.append(b), orig(b))[1]
    reduced = step(red)
    reduced2 = step(red)                                        # second step through the same reducer: counters reset
    same = set(plain) == set(reduced) and all(torch.equal(plain[k], reduced[k]) and torch.equal(plain[k], reduced2[k]) for k in plain)
    print(json.dumps({"backend": dist.get_backend(), "world_size": dist.get_world_size(), "buckets": len(red._buckets),
                      "collectives_issued": red.collectives_issued // 2, "fired_inside_backward": inside[1],
                      "bit_identical": bool(same), "params_compared": len(plain), "side_stream_used": red._side is not None}))
    dist.barrier()
    dist.destroy_process_group()


def _main_graphed():
    """Child process of test_graphed_train_step_in_its_data_parallel_form...: engine.GraphedTrainStep with a reducer on a one-rank
    RCCL group (graph = forward + backward; hook-less all-reduce and an ordinary Adam step outside it) against the same steps
    taken eagerly under the same dropout epochs."""
    import socket

    import torch
    import torch.distributed as dist

    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from fixtures_util import initial_coords, synthetic_node_feats
    from gpu_util import DEV, graph_tensors, model_pair
    from echoglad_amd import engine, ops
    from echoglad_amd.parallel import GradientAllReducer

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        os.environ["MASTER_PORT"] = str(s.getsockname()[1])
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(DEV))
    frame, naux, B, warm, replays = 32, 4, 3, 2, 3
    topo, ei, nt, bi = graph_tensors(frame, naux, B, coord=True)
    x = synthetic_node_feats(B * topo.num_nodes, 128, seed=41).to(DEV)
    eid = ei.to(DEV)
    coords0 = initial_coords(B, frame).to(DEV)

    def build():
        hip, _ = model_pair(frame, naux, 3, coord=True, seed=23)
        hip.train()
        params = list(hip.parameters())
        opt = torch.optim.Adam(params, lr=1e-3)                     # (NOT capturable: the update stays outside the graph)
        red = GradientAllReducer(params, bucket_bytes=32 << 10, force_collective=True)

        def loss_fn():
            logits, coords = hip.forward_nodes(x, eid, B, coords0.clone())
            return (logits ** 2).mean() + (coords ** 2).mean() * 1e-3, logits
        return hip, params, opt, red, loss_fn

    ops.dropout_epoch_set(0)
    hip_g, params_g, opt_g, red_g, loss_g = build()
    torch.manual_seed(77)
    step = engine.GraphedTrainStep(loss_g, opt_g, warmup=warm, reducer=red_g)
    e0 = ops.dropout_epoch()
    losses_g = [float(step()[0]) for _ in range(replays)]
    hooked_refused = False
    try:
        engine.GraphedTrainStep(loss_g, opt_g, warmup=1, reducer=GradientAllReducer(params_g, force_collective=True).attach_hooks())
    except ValueError:
        hooked_refused = True
    ops.dropout_epoch_set(e0)
    hip_e, params_e, opt_e, red_e, loss_e = build()
    torch.manual_seed(77)

    def eager():
        opt_e.zero_grad(set_to_none=True)
        out = loss_e()
        out[0].backward()
        red_e.allreduce()
        opt_e.step()
        return float(out[0].detach())
    for _ in range(warm):
        eager()
    rng = torch.get_rng_state()
    losses_e = []
    for k in range(replays):
        torch.set_rng_state(rng)
        ops.dropout_epoch_set(e0 + k + 1)
        losses_e.append(eager())
    torch.cuda.synchronize()
    same = all(torch.equal(a.detach(), b.detach()) for a, b in zip(params_g, params_e))
    print(json.dumps({"backend": dist.get_backend(), "world_size": dist.get_world_size(), "buckets": len(red_g._buckets),
                      "collectives_issued": red_g.collectives_issued, "steps": warm + replays, "losses_equal": losses_g == losses_e,
                      "params_bit_identical": bool(same), "replays_differ": len(set(losses_g)) == replays,
                      "hooked_reducer_refused": hooked_refused, "epoch_advanced": ops.dropout_epoch() - e0}))
    ops.dropout_epoch_set(0)
    dist.barrier()
    dist.destroy_process_group()


def test_graphed_train_step_in_its_data_parallel_form_on_rccl_world_1():
    """engine.GraphedTrainStep(reducer=...): the captured graph holds forward + backward, the gradient all-reduce (RCCL, hook-less)
    and the optimizer step follow every replay eagerly.  On a one-rank "nccl" group: bit-identical to the eager steps under the same
    dropout epochs, one collective per bucket and step, a reducer with attached hooks is refused."""
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "graphed"], capture_output=True, text=True, timeout=600, env=_child_env())
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["backend"] == "nccl" and d["world_size"] == 1 and d["buckets"] >= 2
    assert d["collectives_issued"] == d["buckets"] * d["steps"]
    assert d["losses_equal"] is True and d["params_bit_identical"] is True and d["replays_differ"] is True
    assert d["hooked_reducer_refused"] is True and d["epoch_advanced"] == 3


if __name__ == "__main__":
    _main_graphed() if sys.argv[1:] == ["graphed"] else _main()


def test_default_bench_takes_the_training_leg_on_every_rank():
    """`bench.py --gpus N` (N > 1) is the inference headline -- no collective in it -- plus a short configs[3] training
    measurement on ALL ranks, so that the driver's multi-GPU run of the default command also meets the gradient all-reduce.
    `--train-leg` runs that code at world size 1 on a one-rank RCCL group: the line carries other_configs.cfg4_train_dp1 with
    both timings (collectives from inside backward / after it), the per-rank diagnostics and the collectives' sizes."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--train-leg", "--steps", "3", "--warmup", "1", "--repeats", "0",
                        "--no-cpu-baseline", "--no-other-configs", "--details", "inline"], capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    js = [l for l in r.stdout.splitlines() if l.startswith("{")]          # (RCCL may print its version banner around the line)
    assert js, (r.stdout[-500:], r.stderr[-500:])
    line = json.loads(js[-1])
    leg = line["other_configs"]["cfg4_train_dp1"]
    assert "error" not in leg, leg
    assert leg["n_gpus"] == 1 and leg["steps"] == 8 and leg["ms_per_step"] > 0 and leg["ms_per_step_collectives_after_backward"] > 0
    assert leg["gradient_collectives"]["collectives_per_step"] >= 1 and leg["gradient_collectives"]["bytes_per_step"] > 0
    assert len(leg["per_rank"]) == 1 and leg["per_rank"][0]["ms_in_finish_behind_collectives"] is not None
    assert line["distributed"]["backend"] == "nccl" and line["distributed"]["allreduce_check"] is True
    assert leg["final_loss"] == leg["final_loss"] and abs(leg["final_loss"]) < 1e30


def _world2_on_one_gpu(extra_env=None, extra_args=()):
    """`bench.py --gpus 2 --backend gloo --share-device` from a parent bench process that never touches the GPU: it starts the two
    ranks itself (both on cuda:0; gloo reduces CUDA tensors through the host, RCCL refuses duplicate devices)."""
    env = _child_env()
    env.update(extra_env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--share-device",
                           "--steps", "5", "--warmup", "2", "--repeats", "1", "--no-cpu-baseline", "--details", "inline", *extra_args],
                          capture_output=True, text=True, timeout=1200, env=env, cwd=ROOT)


def test_world_2_on_one_gpu_runs_the_multi_rank_bench_path_on_real_kernels():
    """The N > 1 code path of the DEFAULT bench command on hardware (replaces src/engine.py:105-110 + dataloader_builder.py:17-22):
    two rank processes, each with its own shard of frames (seeded by global rank), barrier + max-over-ranks timing, every rank's
    logits digest bit-equal to what rank 0 alone computes for the same frames, and the configs[3] training leg on BOTH ranks with
    the GradientAllReducer's buckets fired from gradient hooks inside the real backward, strictly in index order, each on the
    side stream.  What this does NOT cover: RCCL and xGMI (gloo moves the 277 KB through the host) and any timing conclusion --
    the two ranks share one GPU."""
    r = _world2_on_one_gpu()
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-4000:]
    js = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(js) == 1, r.stdout[-1500:]                       # rank 0 alone prints
    d = json.loads(js[0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 16 and d["scaling"] == "weak"
    assert abs(d["value"] - 16 / (d["ms_per_step"] * 1e-3)) < 1e-2 * d["value"]
    dist = d["distributed"]
    assert dist["world_size"] == 2 and dist["backend"] == "gloo" and dist["share_device"] is True and dist["allreduce_check"] is True
    assert dist["shard_digests_equal_single_rank"] is True and dist["shards_checked"] == 1 and dist["shard_digest_mismatch_ranks"] == []
    leg = d["other_configs"]["cfg4_train_dp2"]
    assert "error" not in leg and not leg.get("skipped"), leg
    assert leg["n_gpus"] == 2 and leg["ms_per_step"] > 0 and leg["ms_per_step_collectives_after_backward"] > 0
    n_buckets = leg["gradient_collectives"]["collectives_per_step"]
    assert n_buckets >= 2 and len(leg["per_rank"]) == 2 and sorted(pr["rank"] for pr in leg["per_rank"]) == [0, 1]
    for pr in leg["per_rank"]:
        assert pr["bucket_fire_order"] == list(range(n_buckets)), pr
        assert pr["buckets_fired_inside_backward"] >= n_buckets - 1 and pr["collectives_on_side_stream"] is True, pr
        assert pr["ms_in_finish_behind_collectives"] is not None and pr["ms_per_step_alone"] > 0
        assert pr["final_loss"] == pr["final_loss"] and abs(pr["final_loss"]) < 1e30
    # the ranks train on different frames (their losses differ) but hold the same averaged gradients: after the same number of
    # steps from broadcast parameters both are finite and of the same magnitude
    a, b = (pr["final_loss"] for pr in leg["per_rank"])
    assert a != b and 0.2 < abs(a) / abs(b) < 5.0


def test_world_2_parent_exits_with_the_worst_rank_code():
    """One rank dies before the rendezvous: the parent must not hang in the other rank's wait and must exit non-zero."""
    r = _world2_on_one_gpu({"EG_BENCH_FAIL_RANK": "1", "EG_BENCH_KILL_AFTER": "5"}, ("--no-other-configs",))
    assert r.returncode == 3, (r.returncode, r.stderr[-2000:])
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
