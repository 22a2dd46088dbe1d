#!/usr/bin/env python3
"""Throughput of the EchoGLAD GNN hot path on MI355X: echo frames/s through the full GNN stack.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

A "step" is one pass of the hot path (3 fused GCN layers + the 4 classifier heads, eval mode,
reference src/core/models.py:428-490) over one batch of synthetic node features already
resident in HBM -> logits resident in HBM.  Workload = BASELINE.json configs[1]
(default.yml: 224x224 frame, 7 aux levels, 3 GNN layers, batch 8 per GPU).  Frames are
independent units, so ranks shard the batch with no data-path collective (weak scaling).

`--gpus N` with N > 1 and no WORLD_SIZE in the environment starts the N ranks itself (child processes, created before
this process touches the GPU; a failed rank makes the parent exit non-zero).  Under torch.distributed.run the
environment's RANK / LOCAL_RANK / WORLD_SIZE are used as they are.

Rank 0 prints ONE JSON line with `roofline` (dominant kernel, measured live with HIP events on
the launch stream), `cpu_baseline` (the CPU oracle = port of the reference PyG op sequence,
timed on this box's host cores on a bounded sample), `repeats` (the timed loop repeated) and — at N = 1 —
`other_configs`: BASELINE configs[2] (main grid only, batch 32), configs[4] (448x448, 8 aux levels, batch 8) and one
configs[3] training step (coordinate graph, batch 32), each timed the same way; they never enter `value`."""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import re
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_HBM_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
PEAK_F32_MFMA_TF = 157.3       # MI355X_MICROARCH.md: dense fp32 MFMA peak
C = 128


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=50)
    p.add_argument("--warmup", type=int, default=10)
    p.add_argument("--batch", type=int, default=8, help="frames per GPU per step")
    p.add_argument("--frame", type=int, default=224)
    p.add_argument("--naux", type=int, default=7)
    p.add_argument("--layers", type=int, default=3)
    p.add_argument("--main-only", action="store_true")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-other-configs", action="store_true", help="skip the configs[2] / [3] / [4] side measurements")
    p.add_argument("--cpu-frames", type=int, default=2, help="frames in the CPU-baseline sample")
    p.add_argument("--kernel-iters", type=int, default=30)
    p.add_argument("--repeats", type=int, default=5, help="extra repeats of the K-step timed loop (reported under `repeats`)")
    p.add_argument("--spinup", type=float, default=0.3, help="seconds of untimed steps before the warm-up steps")
    p.add_argument("--no-hip-graph", action="store_true", help="launch the 4 kernels of a step eagerly from Python")
    p.add_argument("--force-collective", action="store_true",
                   help="train mode: initialise torch.distributed (nccl = RCCL) and run the gradient reducer's collectives even "
                        "at world size 1 -- the single-GPU way through the code path the multi-GPU step takes")
    p.add_argument("--train-leg", action="store_true",
                   help="infer mode at world size 1: also run the short configs[3] training measurement the default run takes on ALL ranks "
                        "at world > 1 (other_configs.cfg4_train_dpN), on a one-rank RCCL group -- the single-GPU way through that code")
    p.add_argument("--bucket-kb", type=int, default=0,
                   help="train mode: size of a gradient all-reduce bucket in KiB (0: the reducer's default, max(total / 2, 64 KiB) "
                        "capped at 8 MiB); the first multi-GPU run can sweep it")
    p.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                   help="process-group backend at world > 1: nccl (= RCCL over xGMI, the production path) or gloo (collectives through "
                        "the host: the way to run world > 1 on ONE GPU together with --share-device, RCCL refuses duplicate devices)")
    p.add_argument("--share-device", action="store_true",
                   help="every rank uses cuda:0 (world > 1 on a single-GPU box: the N>1 code path -- shard seeds, digests, gradient hooks "
                        "on a side stream, the all-ranks training leg -- on real kernels; says nothing about RCCL or xGMI)")
    p.add_argument("--details", choices=["file", "stderr", "inline"], default="file",
                   help="where everything beyond the headline goes (other_configs, before / after the path, notes): a JSON file "
                        "(--details-path), one JSON line on stderr, or inline in the stdout line (the pre-round-6 format)")
    p.add_argument("--details-path", default=os.path.join("gpurun_out", "bench_details.json"))
    p.add_argument("--no-graph-replay", action="store_true", help="--mode train: skip the HIP-graph replay of the step (A/B scripts)")
    p.add_argument("--mode", choices=["infer", "train"], default="infer",
                   help="infer (default): BASELINE configs[1], the metric's configuration.  train: one full training "
                        "step of the GNN stack on configs[3] (coordinate graph; SURVEY 8d), reported under its own metric name")
    return p.parse_args()


# ---------------------------------------------------------------------------------------------------------------------
# stdout carries ONE line: the JSON result.  Native libraries write there too (RCCL prints a five-line version banner to
# stdout the first time a communicator is created -- block-buffered, so behind a pipe it lands AFTER the result line): file
# descriptor 1 is pointed at stderr for the whole run and the result goes to a private copy of the real stdout.
# ---------------------------------------------------------------------------------------------------------------------
_REAL_STDOUT = None


def guard_stdout() -> None:
    global _REAL_STDOUT
    if _REAL_STDOUT is None:
        sys.stdout.flush()
        _REAL_STDOUT = os.dup(1)
        os.dup2(2, 1)


def emit(result: dict) -> None:
    line = (json.dumps(result) + "\n").encode()
    if _REAL_STDOUT is None:
        sys.stdout.write(line.decode())
        sys.stdout.flush()
        return
    sys.stdout.flush()
    while line:
        line = line[os.write(_REAL_STDOUT, line):]


# ---------------------------------------------------------------------------------------------------------------------
# N ranks on one node without torchrun
# ---------------------------------------------------------------------------------------------------------------------
def spawn_ranks(n: int) -> int:
    """Start n copies of this command, one per GPU, BEFORE this process makes any GPU call (a process that initialised
    the GPU must never exec / be replaced; children are plain subprocesses).  Returns the worst exit code."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    worst = 0
    deadline = None
    while procs:
        for p in list(procs):
            rc = p.poll()
            if rc is None:
                continue
            procs.remove(p)
            if rc != 0:
                worst = worst or rc
                # a dead rank would leave the others in a collective (or the rendezvous) forever
                deadline = deadline or time.time() + float(os.environ.get("EG_BENCH_KILL_AFTER", "30"))
        if deadline is not None and time.time() > deadline:
            for p in procs:
                p.kill()
            worst = worst or 1
            break
        time.sleep(0.05)
    return worst


# ---------------------------------------------------------------------------------------------------------------------
# models / workloads
# ---------------------------------------------------------------------------------------------------------------------
def model_kwargs(frame, naux, layers, main_only=False, coord=False, conn=False):
    drop = float(os.environ.get("EG_BENCH_DROPOUT", "0.5"))          # diagnostic knob; the reported workloads use 0.5
    return dict(frame_size=frame, gnn_dropout_p=drop, classifier_dropout_p=drop, node_embedding_dim=C,
                node_hidden_dim=C, num_output_channels=4, num_gnn_layers=layers, num_aux_graphs=naux,
                gnn_jk_mode="last", classifier_hidden_dim=32, residual=True, use_coordinate_graph=coord,
                output_activation="logit", use_main_graph_only=main_only, use_connection_nodes=conn)


def build_model(kw, device, train=False):
    from echoglad_amd import nn as egnn
    from echoglad_amd.synthetic import fill_state_dict
    model = egnn.HierarchicalPatchModel(**kw)
    fill_state_dict(model, seed=200)          # glorot-like weights, trained-like BN stats (seed: default.yml:24)
    model = model.to(device)
    return model.train() if train else model.eval()


def stack_work(topo, layers):
    """algorithmic bytes and FLOPs of the whole stack per frame."""
    n, e_dir = topo.num_nodes, 2 * topo.num_undirected_edges
    return n * (layers * 1024 + 528), n * (layers * 2 * C * C + 36992) + layers * (e_dir + n) * 2 * C


def infer_workload(frame, naux, layers, main_only, B, device, rank, hip_graph=True, graph_type="grid", conn=False):
    import torch
    from echoglad_amd.topology import TopologySpec, get_topology
    from echoglad_amd.synthetic import synthetic_node_feats
    kw = model_kwargs(frame, naux, layers, main_only, conn=conn)
    model = build_model(kw, device)
    model.enable_hip_graph(hip_graph)
    topo = get_topology(TopologySpec(frame, naux, main_only, False, conn, graph_type, graph_type))
    # this rank's shard of the global batch: frames [rank*B, (rank+1)*B) — synthetic N(0,1) node features
    feats = synthetic_node_feats(B * topo.num_nodes, C, seed=200 + rank).to(device)
    edge_index = torch.from_numpy(topo.batched_edge_index(B)).to(device)

    def step():
        with torch.no_grad():
            return model.forward_nodes(feats, edge_index, B)[0]

    return model, kw, topo, feats, edge_index, step


def train_workload(frame, naux, layers, B, device, world, rank, force_collective=False, bucket_kb=0, capturable=False):
    """forward + losses + backward + gradient all-reduce + Adam on the stack's parameters,
    node features [B*N,128] resident in HBM, B frames per GPU (32 in BASELINE's cfg4), coordinate graph on."""
    import numpy as np
    import torch
    from echoglad_amd import engine, losses
    from echoglad_amd.data import node_labels
    from echoglad_amd.parallel import GradientAllReducer, broadcast_parameters
    from echoglad_amd.topology import TopologySpec, get_topology
    from echoglad_amd.synthetic import initial_coords, synthetic_node_feats
    model = build_model(model_kwargs(frame, naux, layers, coord=True), device, train=True)
    topo = get_topology(TopologySpec(frame, naux, False, True))
    N, n_valid = topo.num_nodes, topo.num_valid_nodes
    feats = synthetic_node_feats(B * N, C, seed=200 + rank).to(device)
    edge_index = torch.from_numpy(topo.batched_edge_index(B)).to(device)
    coords0 = initial_coords(B, frame).to(device)
    rs = np.random.RandomState(300 + rank)
    y = torch.from_numpy(np.stack([np.stack([node_labels(rs.randint(0, frame, 2), frame, naux) for _ in range(4)], 1)
                                   for _ in range(B)])).reshape(B * n_valid, 4).to(device)
    valid = torch.ones_like(y)
    coord_y = torch.from_numpy(rs.uniform(0, frame - 1, (B * 4, 2)).astype(np.float32)).to(device)
    crit = {"bce": losses.WeightedBCEWithLogitsLoss("none", 9000, 1),
            "elm": losses.ExpectedLandmarkMSE(10, B, frame, naux), "coordinate": engine.MSE(1)}
    params = list(model.parameters())
    # the update as ONE launch (echoglad_amd.optim.Adam: torch's fused-Adam arithmetic, the step count on the device);
    # EG_BENCH_ADAM=torch: torch.optim.Adam(fused=True) -- 4 launches for the 73 parameter tensors
    if os.environ.get("EG_BENCH_ADAM", "eg") == "eg":
        from echoglad_amd.optim import Adam
        opt = Adam(params, lr=1e-4)
    else:
        try:
            opt = torch.optim.Adam(params, lr=1e-4, fused=os.environ.get("EG_BENCH_FUSED_ADAM", "1") != "0", capturable=capturable)
        except (RuntimeError, TypeError):
            opt = torch.optim.Adam(params, lr=1e-4, capturable=capturable)
    reducer = None
    if world > 1 or force_collective:
        broadcast_parameters(model)
        reducer = GradientAllReducer(params, force_collective=force_collective,
                                     bucket_bytes=(bucket_kb << 10) if bucket_kb > 0 else None)
        reducer.attach_hooks()
        reducer.profile = False          # (two timing events per step while on: switched on around the measured steps only)

    def step():
        preds, coord_preds = model.forward_nodes(feats, edge_index, B, coords0)      # (the model does not write into its coordinate input)
        ls = engine.compute_loss(crit, preds, y, coord_preds, coord_y, valid, B)
        loss = engine.total_loss(ls)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        if reducer is not None:
            reducer.finish()
        opt.step()
        return loss

    def loss_fn():                         # (what engine.GraphedTrainStep captures: the same step without the optimizer calls)
        preds, coord_preds = model.forward_nodes(feats, edge_index, B, coords0)
        return engine.total_loss(engine.compute_loss(crit, preds, y, coord_preds, coord_y, valid, B))

    step.reducer = reducer
    step.graphed = (lambda warmup=2: engine.GraphedTrainStep(loss_fn, opt, warmup=warmup)) if capturable and reducer is None else None
    return step, topo


def logits_digest(t):
    """Bit-exact fingerprint of a logits tensor (sha256 of its bytes)."""
    return hashlib.sha256(t.detach().contiguous().cpu().numpy().tobytes()).hexdigest()[:16]


def timed_loop(step, steps, world, device):
    """K steps bracketed by barrier + synchronize on both sides; max over ranks."""
    import torch

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = step()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed, out


def train_layer_roofline(B, topo, device, layers, step=None):
    """The dominant kernel of the training step -- the train-forward layer kernel (k_gcn_layer_ps<false, false, 1>: aggregation,
    128x128 node update, BatchNorm partial sums; 3 launches per step) -- timed live with HIP events on the launch stream INSIDE
    real training steps (`step`: eg_debug_layer_timing_* puts an event pair around every layer-kernel launch, also those issued
    from autograd nodes), so `avg_launch_ms` is what a kernel trace of the step shows.  Algorithmic bytes per launch: read x,
    write z, write A_hat x (kept for dW) = 3 * B * N * 512 B; layers 2, 3 of a step also read the child sums of x."""
    import torch
    from echoglad_amd import ops
    rows = B * topo.num_nodes
    dx_ms = None
    if step is not None:
        n_steps = 6
        with ops.layer_timing(256) as tm:
            for _ in range(n_steps):
                step()
            torch.cuda.synchronize()
        ms = tm.mean_ms("ps_train_fwd")
        dx_ms = tm.mean_ms("ps_dx")
        n_timed = sum(1 for k, _ in tm.launches if k == "ps_train_fwd")
        timing = (f"HIP events around the kernel's launches inside {n_steps} real training steps "
                  f"({n_timed} launches: layer 1 pulls child rows, layers 2-3 read child sums)")
    if step is None or ms is None:
        g = ops.Graph.topo(topo.spec.frame_size, topo.spec.num_aux_graphs, False, True, device=device)
        x = torch.randn(rows, C, device=device)
        W = torch.randn(C, C, device=device) * 0.08
        one, zero = torch.ones(C, device=device), torch.zeros(C, device=device)
        kin = ops.new_kidsum(g, B)
        forms = [dict()] + ([dict(kidsum_in=kin)] * (layers - 1) if kin is not None else [dict()] * (layers - 1))

        def launches():
            for f in forms:
                ops.gcn_layer_train_fwd(g, B, x, W, zero, one, zero, None, None, None, 1e-5, True, 0.0, 0, True, want_out=False, **f)

        ms = time_steps(launches, iters=10, warm=2) / len(forms)
        timing = "the kernel alone on fresh rows (+ the 2 tiny launches that reduce its BatchNorm partial sums, ~25 us)"
    n, e_dir = topo.num_nodes, 2 * topo.num_undirected_edges
    bytes_alg = 3 * rows * C * 4
    flops = B * (n * 2 * C * C + (e_dir + n) * 2 * C)
    tf, gbs = flops / (ms * 1e-3) / 1e12, bytes_alg / (ms * 1e-3) / 1e9
    rf = {"bound": "hbm", "kernel": "k_gcn_layer_ps<false, false, 1> (train-forward layer kernel, 3 launches per step)",
          "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(gbs / PEAK_HBM_GBS, 4),
          "avg_launch_ms": round(ms, 4), "avg_launch_timing": timing, "algorithmic_bytes_per_launch": bytes_alg,
          "dx_launch_ms": None if dx_ms is None else round(dx_ms, 4),
          "mfma": {"achieved": round(tf, 2), "peak": PEAK_F32_MFMA_TF, "unit": "TFLOP/s", "frac": round(tf / PEAK_F32_MFMA_TF, 4),
                   "frac_mfma_only": round(B * n * 2 * C * C / (ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TF, 4)}}
    pm, src = _pmc_summary(r"r\d+_train_pmc\.json", TRAIN_KERNEL_SOURCES)
    rf["traffic"], rf["mfma_busy_frac"], rf["traffic_source"] = None, None, src
    if pm is not None:
        for k, v in pm.items():
            if isinstance(v, dict) and k.startswith("k_gcn_layer_ps<false, false, 1"):
                rf["traffic"] = int(v.get("hbm_bytes_per_launch", 0)) or None
                rf["mfma_busy_frac"] = round(v["mfma_busy_frac"], 4) if "mfma_busy_frac" in v else None
    return rf


def main_train(args, world, rank, device, dist_info):
    import torch
    B = args.batch
    step, topo = train_workload(args.frame, args.naux, args.layers, B, device, world, rank, args.force_collective, args.bucket_kb)
    for _ in range(max(args.warmup, 2)):
        loss = step()
    if step.reducer is not None:
        step.reducer.profile = True                          # (events from here on: the measured steps)
    elapsed, loss = timed_loop(step, args.steps, world, device)
    # per rank: its own clock over the same K steps (the reported time is the maximum), and how long its compute stream sat in
    # finish() behind the gradient collectives -- what to look at first when the scaling curve bends
    t0 = time.perf_counter()
    torch.cuda.synchronize()
    own = []
    for _ in range(min(args.steps, 5)):
        t0 = time.perf_counter()
        step()
        torch.cuda.synchronize()
        own.append(1e3 * (time.perf_counter() - t0))
    wait_ms = step.reducer.collective_wait_ms() if step.reducer is not None else None
    per_rank = [{"rank": rank, "ms_per_step_alone": round(min(own), 3), "ms_in_finish_behind_collectives": None if wait_ms is None else round(wait_ms, 4)}]
    if world > 1:
        gathered = [None] * world
        torch.distributed.all_gather_object(gathered, per_rank[0])
        per_rank = gathered
    if rank == 0:
        fps = world * B * args.steps / elapsed
        sb, sf = stack_work(topo, args.layers)
        out = {
            "metric": "echo frames/sec, one full training step of the GNN stack (fwd + losses + bwd + all-reduce + Adam)",
            "value": round(fps, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": max(args.warmup, 2),
            "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"configs[3]: {args.frame}x{args.frame} frame, {args.naux} aux levels + 4 coordinate nodes, "
                                   f"num_gnn_layers={args.layers}, batch={B} per GPU, train mode (batch-stat BN, dropout 0.5), "
                                   "losses: weighted BCE + expected-landmark MSE + coordinate MSE, Adam",
                       "nodes_per_frame": topo.num_nodes, "global_batch": B * world,
                       "parallelism": f"dp{world} (batch-sharded frames, gradient all-reduce overlapped with backward)"},
            # train-step floor = 3 x the forward's algorithmic bytes / FLOPs
            "stack_hbm_frac": round(fps / world * 3 * sb / 1e9 / PEAK_HBM_GBS, 4),
            "stack_mfma_frac": round(fps / world * 3 * sf / 1e12 / PEAK_F32_MFMA_TF, 4), "final_loss": float(loss.detach()),
            "distributed": dict(dist_info, gradient_collectives_issued=(step.reducer.collectives_issued if step.reducer else 0),
                                gradient_buckets=(len(step.reducer._buckets) if step.reducer else 0),
                                gradient_collectives=(step.reducer.describe() if step.reducer else None),
                                per_rank=per_rank,
                                note="ms_in_finish_behind_collectives = mean time per step the compute stream waited in "
                                     "GradientAllReducer.finish() for the bucket all-reduces (issued from inside backward on a "
                                     "side stream); ms_per_step_alone = this rank's own synchronised step time")}
        if world == 1 and not args.no_other_configs:
            try:
                out["roofline"] = train_layer_roofline(B, topo, device, args.layers, step)
            except Exception as ex:
                out["roofline"] = {"error": repr(ex)}
        if world == 1 and step.reducer is None and not args.no_graph_replay:
            del step
            import gc
            gc.collect()
            torch.cuda.empty_cache()
            out["hip_graph_replay"] = graphed_train_ms(args, device, B, n=max(8, min(args.steps, 50)))
        emit(out)               # (--mode train is a tool's command, not the driver's: one inline line)


def train_leg_all_ranks(args, world, rank, device):
    """The default (inference) run has no collective at all -- frames are independent.  So that a multi-GPU run of the DEFAULT
    command still shows what the gradient all-reduce costs over xGMI, every rank also takes a short configs[3] training
    measurement (batch 32 per GPU, weak scaling): K = 8 steps timed like the headline (barrier + synchronize on both sides, max
    over ranks), each rank's own-clock step time and the time its compute stream waited in finish() behind the collectives,
    and the same K steps with the bucket all-reduces issued AFTER backward instead of from inside it (hooks detached): the
    difference is what the overlap buys -- or costs, next to persistent one-workgroup-per-CU launches.
    Every rank must call this (collectives); rank 0 gets the dictionary, the others None."""
    import gc
    import torch
    B, K = 32, 8
    # Agree across ranks BEFORE the first collective step: a rank that cannot run the step at all (out of memory, an unsupported
    # shape) would otherwise leave the others inside an all-reduce until the process group's timeout aborts the job -- and the
    # headline line with it.  Every rank takes one local step without any collective, then one MIN all-reduce of an ok flag
    # (which every rank reaches: the probe is wrapped) decides for all of them.
    ok, err = 1, None
    try:
        probe, _ = train_workload(224, 7, args.layers, B, device, 1, rank)
        probe()
        torch.cuda.synchronize()
        del probe
    except Exception as ex:
        ok, err = 0, repr(ex)
    gc.collect()
    torch.cuda.empty_cache()
    flag = torch.tensor([ok], dtype=torch.int32, device=device)
    if world > 1:
        torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MIN)
    if int(flag.item()) == 0:
        return {"skipped": True, "error": err or "another rank failed its local probe step"} if rank == 0 else None
    step, topo = train_workload(224, 7, args.layers, B, device, world, rank, force_collective=(world == 1), bucket_kb=args.bucket_kb)
    red = step.reducer
    for _ in range(3):
        loss = step()
    red.profile = True
    elapsed, loss = timed_loop(step, K, world, device)
    torch.cuda.synchronize()
    own = []
    for _ in range(3):
        t0 = time.perf_counter()
        step()
        torch.cuda.synchronize()
        own.append(1e3 * (time.perf_counter() - t0))
    wait_ms = red.collective_wait_ms()
    red.trace = []                                       # one traced step: which buckets left from inside backward, in which order
    step()
    torch.cuda.synchronize()
    tr, red.trace = red.trace, None
    red.detach_hooks()                                   # same buckets, issued from finish(): nothing overlaps the backward
    for _ in range(2):
        step()
    red.collective_wait_ms()
    elapsed_after, _ = timed_loop(step, K, world, device)
    wait_after = red.collective_wait_ms()
    mine = {"rank": rank, "ms_per_step_alone": round(min(own), 3), "ms_in_finish_behind_collectives": None if wait_ms is None else round(wait_ms, 4),
            "ms_in_finish_collectives_after_backward": None if wait_after is None else round(wait_after, 4),
            "bucket_fire_order": [b for b, _, _ in tr], "buckets_fired_inside_backward": sum(1 for _, h, _ in tr if h),
            "collectives_on_side_stream": bool(tr) and all(sd for _, _, sd in tr), "final_loss": float(loss.detach())}
    per_rank = [mine]
    if world > 1:
        per_rank = [None] * world
        torch.distributed.all_gather_object(per_rank, mine)
    out = None
    if rank == 0:
        out = {"workload": f"configs[3]: 224x224, 7 aux levels + coordinate graph, batch {B} per GPU x {world} GPUs (weak scaling), one training "
                           "step (fwd + 3 losses + bwd + gradient all-reduce + Adam); frames are independent: the all-reduce is the only collective",
               "n_gpus": world, "steps": K, "ms_per_step": round(1e3 * elapsed / K, 3), "frames_s": round(world * B * K / elapsed, 1),
               "ms_per_step_collectives_after_backward": round(1e3 * elapsed_after / K, 3), "scaling": "weak",
               "gradient_collectives": red.describe(), "per_rank": per_rank, "final_loss": float(loss.detach()),
               "note": "ms_per_step: bucket all-reduces issued from inside backward on a side stream; ..._after_backward: the same buckets "
                       "issued from finish(); per_rank: own-clock step time and the time the compute stream waited behind the collectives"}
    del step, red
    gc.collect()
    torch.cuda.empty_cache()
    return out


def cpu_baseline(args, kw, state_dict):
    """The oracle (a port of the reference's PyG op sequence: gcn_norm recomputed per layer,
    x W^T, index_select gather, scale, index_add_ scatter, BN, ReLU, residual, 4 classifier MLPs)
    on the host cores, on a bounded sample of the same workload."""
    import numpy as np
    import torch
    from oracle import gnn_oracle as O
    from echoglad_amd.topology import TopologySpec, get_topology
    from echoglad_amd.synthetic import synthetic_node_feats
    avail = os.cpu_count() or 1
    ref = O.OracleHierarchicalPatchModel(**kw)
    ref.load_state_dict({k: v.cpu() for k, v in state_dict.items()}, strict=True)
    ref.eval()
    topo = get_topology(TopologySpec(args.frame, args.naux, args.main_only))
    B = args.cpu_frames
    ei = torch.from_numpy(topo.batched_edge_index(B))
    nt = torch.from_numpy(np.tile(topo.node_type(), B))
    feats = synthetic_node_feats(B * topo.num_nodes, C, seed=200)
    # torch's CPU index_add_/index_select stop scaling (and regress) with very many threads, so a few
    # thread counts are probed once and the fastest is used for the timed runs
    probe = {}
    with torch.no_grad():
        for th in sorted({min(avail, t) for t in (8, 16, 32, 64)}):
            torch.set_num_threads(th)
            ref.forward_nodes(feats, ei, nt, B)                  # warm-up at this thread count
            t0 = time.perf_counter()
            ref.forward_nodes(feats, ei, nt, B)
            probe[th] = time.perf_counter() - t0
            if sum(probe.values()) > 25.0:
                break
        cores = min(probe, key=probe.get)
        torch.set_num_threads(cores)
        times = [probe[cores]]
        t_end = time.perf_counter() + 10.0
        while len(times) < 5 and time.perf_counter() < t_end:
            t0 = time.perf_counter()
            out, _ = ref.forward_nodes(feats, ei, nt, B)
            times.append(time.perf_counter() - t0)
        out, _ = ref.forward_nodes(feats, ei, nt, B)
    best = min(times)
    return {"value": round(B / best, 3), "unit": "frames/s", "cores": cores, "kind": "port", "sample_batch": B,
            "sample_short": f"{B} frames of the workload, min of {len(times)} runs",
            "sample": f"{B} frames of the same workload (F={args.frame}, naux={args.naux}, L={args.layers}), "
                      f"min of {len(times)} runs after warm-up, torch CPU fp32, {cores} threads "
                      f"(fastest of the probed thread counts {sorted(probe)} on {avail} host cores)"}, out, feats, ei


DOMINANT_KERNEL_SOURCES = ("gcn_layer_ps.hip", "seg_wide.h", "tile.h", "common.h", "graph.hip", "conn.hip")
TRAIN_KERNEL_SOURCES = ("gcn_layer.hip", "gcn_layer_ps.hip", "train.hip", "bn_act_tiles.hip", "cls_train.hip", "train_common.h",
                        "seg_wide.h", "tile.h", "common.h", "graph.hip", "conn.hip", "coord.hip", "coord_common.h", "coord_mlp.hip", "heatmap.hip",
                        "adam.hip")


def kernel_source_digest(sources=DOMINANT_KERNEL_SOURCES) -> str:
    """sha256 over the sources of the dominant kernel (the chained layer kernel, its device helpers and the topology tables
    it reads): ties a committed PMC summary to the code it was measured on (the GPU box has no .git).  With
    TRAIN_KERNEL_SOURCES: the same for the kernels of the training step."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "echoglad_amd", "csrc")
    for f in sources:
        h.update(f.encode())
        h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def _pmc_summary(pattern: str, sources):
    """Newest profiles/<pattern> (r{NN}_pmc.json: the inference step; r{NN}_train_pmc.json: the training step) whose
    `kernel_source_digest` equals the current sources'.  Files of the other kind, or any other *_pmc.json that happens to
    lie in profiles/, are never candidates.  Returns (summary | None, source-or-reason)."""
    pdir = os.path.join(ROOT, "profiles")
    rx = re.compile(pattern)
    cands = sorted([f for f in os.listdir(pdir) if rx.fullmatch(f)], reverse=True) if os.path.isdir(pdir) else []
    if not cands:
        return None, f"no profiles/{pattern}"
    want = kernel_source_digest(sources)
    try:
        pm = json.load(open(os.path.join(pdir, cands[0])))
    except Exception as ex:
        return None, f"profiles/{cands[0]}: {ex!r}"
    if pm.get("kernel_source_digest") != want:
        return None, (f"profiles/{cands[0]} is stale (measured on kernel sources {pm.get('kernel_source_digest')}, "
                      f"current {want}): re-run tools/profile_round.sh")
    return pm, "profiles/" + cands[0]


def pmc_traffic(kernel_key: str):
    """HBM bytes per launch of the dominant kernel from the newest profiles/r{NN}_pmc.json (tools/profile_round.sh:
    FETCH_SIZE and WRITE_SIZE in separate passes, FETCH_SIZE doubled per MI355X_MICROARCH.md §HBM).  A summary measured
    on other kernel sources is NOT used: `traffic` stays null and the source says why."""
    pm, src = _pmc_summary(r"r\d+_pmc\.json", DOMINANT_KERNEL_SOURCES)
    if pm is None:
        return None, src
    try:
        return int(pm[kernel_key]["hbm_bytes_per_launch"]), src
    except Exception as ex:
        return None, f"{src}: {ex!r}"


def train_pmc_traffic():
    """HBM bytes of one configs[3] training step (batch 32 per GPU) from the newest profiles/r{NN}_train_pmc.json
    (tools/profile_train_pmc.sh; per-kernel bytes per launch x launches per step), same staleness rule."""
    pm, src = _pmc_summary(r"r\d+_train_pmc\.json", TRAIN_KERNEL_SOURCES)
    if pm is None:
        return None, src
    try:
        return int(pm["hbm_bytes_per_step"]), src
    except Exception as ex:
        return None, f"{src}: {ex!r}"


def time_steps(step, iters=20, warm=5):
    """ms per call of `step` with HIP events on the current stream."""
    import torch
    for _ in range(warm):
        step()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        step()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def _infer_entry(args, device, what, frame, naux, main_only, B, **kw):
    """One inference side configuration timed like the headline (HIP-graph replay, inputs resident in HBM)."""
    import gc
    import torch
    try:
        model, _, topo, feats, ei, step = infer_workload(frame, naux, args.layers, main_only, B, device, 0, **kw)
        ms = time_steps(step, iters=20, warm=5)
        graph, _ = model._resolver.resolve(ei, feats.shape[0])
        sb, sf = stack_work(topo, args.layers)
        fps = B / (ms * 1e-3)
        out = {"workload": what, "ms_per_step": round(ms, 4), "frames_s": round(fps, 1),
               "mfma_frac": round(fps * sf / 1e12 / PEAK_F32_MFMA_TF, 4), "hbm_frac": round(fps * sb / 1e9 / PEAK_HBM_GBS, 4),
               "nodes_per_frame": topo.num_nodes, "stencil_handle": bool(graph.structured)}
        del model, feats, ei, step
    except Exception as ex:
        out = {"workload": what, "error": repr(ex)}
    gc.collect()
    torch.cuda.empty_cache()
    return out


def pyg_surface_entry(args, device, B=8):
    """INTEGRATION route B timed: the reference's forward loop (models.py:426-435 layer call + residual, :485 node-type
    filter, :488-490 four heads + cat) written as the reference writes it, over this package's torch_geometric-shaped modules
    (Sequential('x, edge_index', [GCNConv, BatchNorm1d, Dropout, ReLU]) -> one fused launch per layer); the residual add, the
    row filter and the heads stay the torch ops the unchanged models.py issues.  Eval mode, configs[1] shape."""
    import gc
    import torch
    what = f"configs[1] via the reference-shaped loop (route B), batch {B}, eval"
    try:
        model, _, topo, feats, ei, _ = infer_workload(224, 7, args.layers, False, B, device, 0, hip_graph=False)
        n, n_valid = topo.num_nodes, topo.num_valid_nodes
        node_type = torch.from_numpy(__import__("numpy").tile(topo.node_type(), B)).to(device)
        keep = torch.nonzero(node_type == 0).squeeze(1)               # (static per topology: the reference re-derives it with a host sync)
        from echoglad_amd import nn as egnn

        def layers_only():
            hidden = [feats]
            for i in range(args.layers):
                h = model.gnn_layers[i](hidden[i], ei)
                h = h + hidden[i]
                hidden.append(h)
            return hidden[-1]

        def step():
            h = layers_only()
            h = h[keep]
            return torch.cat([clf(h) for clf in model.node_classifiers], dim=1).squeeze(1)

        graph, gb = egnn._SHARED_RESOLVER.resolve(ei, feats.shape[0])
        with torch.no_grad():
            l0, p0 = graph.layer_launches, graph.ps_launches
            got = step()
            per_step = (graph.layer_launches - l0, graph.ps_launches - p0)
            ms = time_steps(step, iters=20, warm=5)
            ms_layers = time_steps(layers_only, iters=20, warm=3)
            forced = os.environ.get("EG_SEQ_FUSED")              # (a caller's / knob matrix's own setting survives this entry)
            os.environ["EG_SEQ_FUSED"] = "0"
            try:
                ms_unfused = time_steps(step, iters=10, warm=3)
            finally:
                if forced is None:
                    del os.environ["EG_SEQ_FUSED"]
                else:
                    os.environ["EG_SEQ_FUSED"] = forced
            same = float((got - model.forward_nodes(feats, ei, B)[0]).abs().max())
        sb, sf = stack_work(topo, args.layers)
        fps = B / (ms * 1e-3)
        out = {"workload": what, "ms_per_step": round(ms, 4), "frames_s": round(fps, 1),
               "mfma_frac": round(fps * sf / 1e12 / PEAK_F32_MFMA_TF, 4), "hbm_frac": round(fps * sb / 1e9 / PEAK_HBM_GBS, 4),
               "ms_layers_and_residual_adds": round(ms_layers, 4), "ms_per_step_module_by_module": round(ms_unfused, 4),
               "launches": {"fused_layer_launches_per_step": per_step[0], "of_them_producer_consumer": per_step[1],
                            "torch": f"{args.layers} residual adds, 1 row gather, 4 heads x (3 Linear + 2 BatchNorm1d + 2 ReLU) + cat"},
               "stencil_handle": bool(graph.structured), "max_abs_diff_vs_route_A": same}
        if forced == "0":
            out["note"] = "EG_SEQ_FUSED=0 was set for the whole run: ms_per_step and the launch counts are the module-by-module route too"
        del model, feats, ei
    except Exception as ex:
        out = {"workload": what, "error": repr(ex)}
    gc.collect()
    torch.cuda.empty_cache()
    return out


def train_entry(args, device, B, what, roofline=False):
    import gc
    import torch
    try:
        step, topo = train_workload(224, 7, args.layers, B, device, 1, 0)
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 8
        for _ in range(n):
            step()
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / n
        sb, sf = stack_work(topo, args.layers)
        fps = B / (ms * 1e-3)
        out = {"workload": what, "ms_per_step": round(ms, 3), "frames_s": round(fps, 1),
               "mfma_frac": round(fps * 3 * sf / 1e12 / PEAK_F32_MFMA_TF, 4), "hbm_frac": round(fps * 3 * sb / 1e9 / PEAK_HBM_GBS, 4),
               "nodes_per_frame": topo.num_nodes, "floor": "3 x forward bytes / FLOPs (SURVEY 8d)"}
        if roofline:
            tb, tsrc = train_pmc_traffic()
            out.update({"traffic_bytes_per_step": tb, "traffic_source": tsrc, "algorithmic_bytes_per_step": 3 * sb * B})
            out["roofline"] = train_layer_roofline(B, topo, device, args.layers, step)
        del step
    except Exception as ex:
        out = {"workload": what, "error": repr(ex)}
    gc.collect()
    torch.cuda.empty_cache()
    if "error" not in out:
        out["hip_graph_replay"] = graphed_train_ms(args, device, B)
    return out


def graphed_train_ms(args, device, B, n=8):
    """The same training step as ONE HIP graph (echoglad_amd.engine.GraphedTrainStep: forward, criteria, backward and a capturable
    fused Adam captured once; the graph's first node bumps the device's dropout epoch, so every replay draws fresh masks)."""
    import gc
    import torch
    try:
        step, _ = train_workload(224, 7, args.layers, B, device, 1, 0, capturable=True)
        g = step.graphed(2)
        for _ in range(2):
            g()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            g()
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / n
        res = {"ms_per_step": round(ms, 3), "frames_s": round(B / (ms * 1e-3), 1), "final_loss": float(g.outputs[0])}
        del g, step
    except Exception as ex:
        res = {"error": repr(ex)}
    gc.collect()
    torch.cuda.empty_cache()
    return res


def documented_graphed_loop_ms(args, device, B=1, n=24):
    """INTEGRATION.md section E as written, timed: frames -> embedder (1x1 conv) -> model(x=...) -> three criteria -> backward ->
    Adam as ONE HIP graph over a static collated batch, every step preceded by ``data.copy_batch_(static, next batch)`` with
    batches fresh from ``collate`` on the host (pageable memory, as a DataLoader hands them over).  What graphed_train_ms leaves out:
    the node-feature packing in front of the stack and the per-step host-to-device copies of x / y / valid_labels."""
    import gc
    import torch
    try:
        from echoglad_amd import data, engine, losses
        frame, naux = 224, 7
        model = build_model(model_kwargs(frame, naux, args.layers, coord=True), device, train=True)
        emb = torch.nn.Conv2d(1, C, kernel_size=1).to(device)
        ds = data.SyntheticEchoDataset(num_aux_graphs=naux, frame_size=frame, use_coordinate_graph=True)
        host = [data.collate([ds[i * B + j] for j in range(B)], ds.topology) for i in range(4)]
        import copy
        static = data.to_device(copy.copy(host[0]), device)
        crit = {"bce": losses.WeightedBCEWithLogitsLoss("none", 9000, 1), "elm": losses.ExpectedLandmarkMSE(10, B, frame, naux),
                "coordinate": engine.MSE(1)}
        params = list(model.parameters()) + list(emb.parameters())
        if os.environ.get("EG_BENCH_ADAM", "eg") == "eg":
            from echoglad_amd.optim import Adam
            opt = Adam(params, lr=1e-4)
        else:
            opt = torch.optim.Adam(params, lr=1e-4, fused=True, capturable=True)
        md = {"embedder": emb, "landmark": model}

        def loss_fn():                     # (copy_batch_ writes every new batch's landmark guesses into static.node_coords; the model does not write there)
            preds, cp = engine.forward_batch(md, static, True)
            return engine.total_loss(engine.compute_loss(crit, preds, static.y, cp, static.node_coord_y, static.valid_labels, B))

        step = engine.GraphedTrainStep(loss_fn, opt, warmup=2)
        v0 = static.edge_index._version
        for k in range(3):
            data.copy_batch_(static, host[k % 4]); step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(n):
            data.copy_batch_(static, host[k % 4])
            out = step()
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / n
        res = {"ms_per_step": round(ms, 3), "frames_s": round(B / (ms * 1e-3), 1), "final_loss": float(out[0]),
               "edge_index_untouched": static.edge_index._version == v0,
               "what": "INTEGRATION.md E: copy_batch_(static, fresh host batch) + one graph launch per step (embedder, packing, stack, "
                       "3 criteria, backward, Adam as one launch)"}
        del step, model, emb, static, host
    except Exception as ex:
        res = {"error": repr(ex)}
    gc.collect()
    torch.cuda.empty_cache()
    return res


def other_configs(args, device):
    """The BASELINE configs the metric is not quoted on, timed like the headline (HIP-graph replay, inputs in HBM);
    reported beside it, never inside `value`.  (What each entry is: DESIGN.md section 4; the strings here stay short so that the
    one JSON line fits the driver's tail window.)"""
    import gc
    import torch
    out = {}
    out["cfg3"] = _infer_entry(args, device, "configs[2]: main grid only, 224x224, batch 32, eval", 224, 7, True, 32)
    out["cfg5"] = _infer_entry(args, device, "configs[4]: 448x448, 8 aux levels, batch 8 per GPU, eval", 448, 8, False, 8)
    out["cfg2_diagonal"] = _infer_entry(args, device, "configs[1] shape, 'grid-diagonal' levels, batch 8, eval", 224, 7, False, 8,
                                        graph_type="grid-diagonal")
    out["cfg2_connection_nodes"] = _infer_entry(args, device, "configs[1] shape, use_connection_nodes, batch 8, eval", 224, 7, False, 8,
                                                conn=True)
    # the reference's own operating point: batch_size 1 (configs/default.yml:27)
    b1 = _infer_entry(args, device, "configs[1] shape at batch 1 (default.yml:27), eval", 224, 7, False, 1)
    if "ms_per_step" in b1:
        b1["us_per_frame"] = round(1e3 * b1["ms_per_step"], 1)
    out["cfg2_b1"] = b1
    out["cfg2_pyg_surface"] = pyg_surface_entry(args, device)
    what = "configs[1] through a CSR handle (arbitrary edge_index), batch 8, eval"
    try:
        from echoglad_amd import ops
        model, kw, topo, feats, ei, _ = infer_workload(224, 7, args.layers, False, 8, device, 0, hip_graph=False)
        g = ops.Graph.csr(ei, feats.shape[0])
        folded, packed = model._folded_layers(), model._packed_classifier()

        def csr_step():
            h = feats
            for i, (w, sc, sh) in enumerate(folded):
                h = ops.gcn_layer_fwd(g, 1, h, w, sc, sh, h, relu=i < len(folded) - 1)
            return ops.classifier_fwd(h, 8, topo.num_nodes, 0, topo.num_nodes, packed)

        with torch.no_grad():
            ms = time_steps(csr_step, iters=20, warm=5)
            same = float((csr_step() - model.forward_nodes(feats, ei, 8)[0]).abs().max())
        sb, sf = stack_work(topo, args.layers)
        fps = 8 / (ms * 1e-3)
        out["cfg2_csr_fallback"] = {"workload": what, "ms_per_step": round(ms, 4), "frames_s": round(fps, 1),
                                    "mfma_frac": round(fps * sf / 1e12 / PEAK_F32_MFMA_TF, 4),
                                    "hbm_frac": round(fps * (sb + 8 * 2 * topo.num_undirected_edges * args.layers) / 1e9 / PEAK_HBM_GBS, 4),
                                    "max_abs_diff_vs_stencil_path": same}
        del model, feats, ei, g
    except Exception as ex:
        out["cfg2_csr_fallback"] = {"workload": what, "error": repr(ex)}
    gc.collect()
    torch.cuda.empty_cache()
    out["cfg4_train"] = train_entry(args, device, 32, "configs[3]: 224/7 + coordinate graph, batch 32 per GPU, one train step", roofline=True)
    out["cfg4_train_b1"] = train_entry(args, device, 1, "configs[3] shape at batch 1 (default.yml:27), one train step")
    out["cfg4_train_b1"]["documented_graphed_loop"] = documented_graphed_loop_ms(args, device)
    return out


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # no GPU call has happened in this process; the ranks are children, this process only waits for them
        sys.exit(spawn_ranks(args.gpus))

    guard_stdout()                       # from here on only emit() reaches the real stdout
    import warnings
    warnings.filterwarnings("ignore", message="The given NumPy array is not writable")      # (cached read-only edge lists, only read)
    import numpy as np  # noqa: F401
    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    if os.environ.get("EG_BENCH_FAIL_RANK") == str(rank) and world > 1:
        raise SystemExit(3)               # (test hook: one rank dies before the rendezvous; the parent must report it)
    dev_index = 0 if args.share_device else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    dist_info = {"world_size": 1, "backend": None, "allreduce_check": None}
    use_dist = world > 1 or (args.force_collective and args.mode == "train") or (args.train_leg and args.mode == "infer")
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:                                              # --force-collective on one GPU: a one-rank RCCL group
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ.setdefault("MASTER_PORT", str(sk.getsockname()[1]))
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        import datetime
        # (a bounded timeout: a collective that never completes must end the run with an error, not hold a node for ten minutes)
        if args.share_device and args.backend == "nccl" and world > 1:
            raise SystemExit("--share-device needs --backend gloo: RCCL refuses two ranks on one device")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=device, timeout=datetime.timedelta(seconds=300))      # "nccl" IS RCCL on ROCm
        else:
            dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=300))      # (CUDA tensors are reduced through the host)
        one = torch.ones(1, device=device)
        dist.all_reduce(one)
        dist_info = {"world_size": dist.get_world_size(), "backend": dist.get_backend(),
                     "allreduce_check": float(one.item()) == float(world), "share_device": bool(args.share_device)}
        if not dist_info["allreduce_check"]:
            raise SystemExit(f"rank {rank}: all-reduce of ones over {world} ranks gave {float(one.item())}")
    if args.gpus != world and rank == 0:
        print(f"warning: --gpus {args.gpus} != WORLD_SIZE {world}", file=sys.stderr)
    try:
        if args.mode == "train":
            main_train(args, world, rank, device, dist_info)
        else:
            main_infer(args, world, rank, device, dist_info)
    finally:
        if use_dist:
            torch.distributed.barrier()
            torch.distributed.destroy_process_group()


def main_infer(args, world, rank, device, dist_info):
    import torch
    from echoglad_amd import ops
    B = args.batch
    model, kw, topo, feats, edge_index, step = infer_workload(args.frame, args.naux, args.layers, args.main_only, B, device,
                                                              rank, hip_graph=not args.no_hip_graph)
    N = topo.num_nodes
    # clock / allocator spin-up before the W warm-up steps (untimed): freshly started boxes showed one-off
    # stalls of tens of ms (first graph replays, allocator growth, DVFS ramp) that would land in a short timed loop
    t_spin = time.perf_counter()
    while time.perf_counter() - t_spin < args.spinup:
        out = step()
        torch.cuda.synchronize()
    # ... and one un-synchronised burst as deep as the timed loop, so that whatever the runtime grows the first
    # time that many launches are queued (observed: a one-off ~45 ms stall under torch.distributed.run) is paid here
    for _ in range(min(max(args.steps + args.warmup, 32), 512)):
        out = step()
    torch.cuda.synchronize()
    for _ in range(args.warmup):
        out = step()
    elapsed, out = timed_loop(step, args.steps, world, device)          # THE timed region: exactly K steps
    frames_per_s = world * B * args.steps / elapsed
    rep = [1e3 * timed_loop(step, args.steps, world, device)[0] / args.steps for _ in range(max(args.repeats, 0))]
    # the same step launched eagerly from Python (4 launches + queue resets per step), timed the same way: what a caller without
    # enable_hip_graph() gets
    eager_ms = None
    if not args.no_hip_graph:
        model.use_hip_graph = False                  # (the captured graph stays where it is)
        for _ in range(max(args.warmup, 3)):
            step()
        eager_ms = 1e3 * timed_loop(step, args.steps, world, device)[0] / args.steps
        model.use_hip_graph = True
        out = step()

    # ---- every rank's logits for ITS shard must be what one GPU alone computes for the same frames (frames are seeded by global
    # rank index; the kernels are bitwise deterministic): rank 0 recomputes every other shard and compares digests
    if world > 1:
        digest = logits_digest(out)
        digests = [None] * world
        torch.distributed.all_gather_object(digests, digest)
        if rank == 0:
            from echoglad_amd.synthetic import synthetic_node_feats
            bad = []
            for r in range(1, world):
                fr = synthetic_node_feats(B * N, C, seed=200 + r).to(device)
                with torch.no_grad():
                    lone = model.forward_nodes(fr, edge_index, B)[0]
                if logits_digest(lone) != digests[r]:
                    bad.append(r)
                del fr
            dist_info = dict(dist_info, shard_digests_equal_single_rank=not bad, shards_checked=world - 1,
                             shard_digest_mismatch_ranks=bad)
            if bad:
                print(f"warning: ranks {bad} computed other logits for their shard than rank 0 does for the same frames", file=sys.stderr)
    # ---- world > 1 (or --train-leg): the short training measurement every rank takes part in (the only collective of this framework)
    train_leg = None
    if (args.train_leg or (world > 1 and not args.no_other_configs)) and args.frame == 224 and args.naux == 7 and not args.main_only \
            and os.environ.get("EG_BENCH_TRAIN_LEG", "1") != "0":
        try:
            train_leg = train_leg_all_ranks(args, world, rank, device)
        except Exception as ex:                               # (a failure here must not cost the headline line)
            train_leg = {"error": repr(ex)}
    if rank != 0:
        return
    # ---- dominant kernel: the fused GCN layer, timed with HIP events on the launch stream
    graph, gb = model._resolver.resolve(edge_index, feats.shape[0])
    w, scale, shift = model._folded_layers()[0]
    buf = torch.empty_like(feats)
    chained = bool(model.chain_layers and graph.kidsum_rows > 0 and args.layers > 1)
    if chained:
        # the step's layers run chained (eg_gcn_layer_fwd_chain): time the three forms the step launches
        # (first: writes child sums; middle: reads + writes; last: reads) and report their mean, which is what
        # a kernel trace shows as the average duration of k_gcn_layer_ps
        ka, kb = model._kidsum_buffers(graph, gb)
        forms = [dict(kidsum_out=ka)] + [dict(kidsum_in=ka, kidsum_out=kb)] * max(args.layers - 2, 0)
        fused_cls = bool(model.fuse_classifier) and topo.n_conn == 0 and topo.num_valid_nodes == N
        if not fused_cls:
            forms.append(dict(kidsum_in=ka))          # (with the classifier fused, the last layer is k_gcn_layer_ps<true>)
        kname = "k_gcn_layer_ps<false>" if fused_cls else "k_gcn_layer_ps"
    else:
        forms = [dict()]
        flat = graph.structured and graph.kidsum_rows == 0 and graph.fused_classifier_ok       # single-level topology
        kname = "k_gcn_layer_ps<false>" if flat else ("k_gcn_layer<AGG_STENCIL>" if graph.structured else "k_gcn_layer<AGG_CSR>")

    def layer_launches():
        for f in forms:
            ops.gcn_layer_fwd(graph, gb, feats, w, scale, shift, feats, relu=True, out=buf, **f)

    timing_note = "each form of the launch on the step's input, back to back (HIP events around the loop)"
    if chained and fused_cls and args.layers >= 2 and model.jk is None:
        # The step itself, launch by launch: layer 1 -> ... -> last layer + heads, every launch reading what the one before it wrote
        # (as the replayed graph does), HIP events between the launches; the dominant kernel's average duration is the mean over the
        # step's plain-layer launches.  (Each form timed alone on the same input measured 4 - 6 % more than the same launches take
        # inside the step -- profiles/r04_infer_step_sequence.txt -- and would not be what a kernel trace of this command shows.)
        folded, packed = model._folded_layers(), model._packed_classifier()
        bufs = [torch.empty_like(feats) for _ in range(2)]
        n_plain = args.layers - 1

        def step_launches(events=None):
            x = feats
            for i in range(n_plain):
                wi, si, hi = folded[i]
                if events is not None:
                    events[i].record()
                x = ops.gcn_layer_fwd(graph, gb, x, wi, si, hi, x, relu=True, out=bufs[i & 1], kidsum_in=(ka, kb)[(i + 1) & 1] if i > 0 else None,
                                      kidsum_out=(ka, kb)[i & 1])
            if events is not None:
                events[n_plain].record()
            wl, sl, hl = folded[args.layers - 1]
            return ops.gcn_layer_cls_fwd(graph, gb, x, wl, sl, hl, x, False, packed, sigmoid=False, kidsum_in=(ka, kb)[(n_plain + 1) & 1])

        for _ in range(3):
            step_launches()
        torch.cuda.synchronize()
        evs = [[torch.cuda.Event(enable_timing=True) for _ in range(n_plain + 1)] for _ in range(args.kernel_iters)]
        for it in range(args.kernel_iters):
            step_launches(evs[it])
        torch.cuda.synchronize()
        layer_ms = sum(e[i].elapsed_time(e[i + 1]) for e in evs for i in range(n_plain)) / (n_plain * args.kernel_iters)
        timing_note = ("the step's launches in sequence (each reads what the one before it wrote), HIP events between them: mean over the "
                       f"step's {n_plain} plain-layer launches x {args.kernel_iters} steps")
        del bufs
    else:
        layer_ms = time_steps(layer_launches, iters=args.kernel_iters, warm=3) / len(forms)
    e_dir = 2 * topo.num_undirected_edges
    flops = B * (N * 2 * C * C + (e_dir + N) * 2 * C)             # SURVEY §8(d) per-layer FLOPs
    bytes_alg = B * N * 2 * C * 4                                  # read x once + write out once
    tf = flops / (layer_ms * 1e-3) / 1e12
    gbs = bytes_alg / (layer_ms * 1e-3) / 1e9
    traffic, traffic_src = (None, "measured only for the default workload")
    if args.frame == 224 and args.naux == 7 and B == 8 and not args.main_only:
        traffic, traffic_src = pmc_traffic("k_gcn_layer")
    roofline = {"bound": "mfma", "kernel": kname,
                "achieved": round(tf, 2), "peak": PEAK_F32_MFMA_TF, "unit": "TFLOP/s",
                "frac": round(tf / PEAK_F32_MFMA_TF, 4),
                # `frac` books SURVEY 8(d)'s per-layer FLOPs, which include the aggregation's FMAs (vector ALU work); the 128x128
                # node update alone -- what the MFMA pipe executes -- against the same roof:
                "frac_mfma_only": round(B * N * 2 * C * C / (layer_ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TF, 4),
                "traffic": traffic, "traffic_source": traffic_src,
                "avg_launch_ms": round(layer_ms, 4), "avg_launch_timing": timing_note,
                "hbm": {"achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                        "frac": round(gbs / PEAK_HBM_GBS, 4), "algorithmic_bytes_per_launch": bytes_alg}}
    stack_bytes, stack_flops = stack_work(topo, args.layers)
    result = {
        "metric": "echo frames/sec through full GNN stack (224x224 default hier-graph); MAE parity",
        "value": round(frames_per_s, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 4),
        "eager_ms_per_step": None if eager_ms is None else round(eager_ms, 4), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"configs[1] default.yml: {args.frame}x{args.frame} frame, "
                               f"{'main grid only' if args.main_only else str(args.naux) + ' aux levels'}, "
                               f"num_gnn_layers={args.layers}, batch={B} per GPU, eval mode, "
                               "node features [B*N,128] in HBM -> logits [B*N_valid,4] in HBM",
                   "workload_short": f"configs[1] default.yml: {args.frame}x{args.frame}, "
                                     f"{'main grid only' if args.main_only else str(args.naux) + ' aux levels'}, L={args.layers}, "
                                     f"batch {B}/GPU, eval",
                   "nodes_per_frame": N, "directed_edges_per_frame": e_dir, "global_batch": B * world,
                   "parallelism": f"dp{world} (batch-sharded frames, no data-path collective)",
                   "launch": "eager" if args.no_hip_graph else "hipGraph replay of the step's kernels",
                   "kernels_per_step": ("2 x k_gcn_layer_ps<false> (chained layers) + k_gcn_layer_ps<true> (last layer + "
                                        "classifier heads)") if (graph.kidsum_rows > 0 and model.chain_layers and
                                                                  model.fuse_classifier) else "3 layer launches + k_classifier"},
        "distributed": dist_info,
        "repeats": ({"ms_per_step": {"min": round(min(rep), 4), "median": round(sorted(rep)[len(rep) // 2], 4),
                                     "max": round(max(rep), 4)}, "n": len(rep),
                     "note": "the K-step timed loop run again n times after the reported one"} if rep else None),
        "stack_hbm_frac": round(frames_per_s / world * stack_bytes / 1e9 / PEAK_HBM_GBS, 4),
        "stack_mfma_frac": round(frames_per_s / world * stack_flops / 1e12 / PEAK_F32_MFMA_TF, 4),
        "roofline": roofline,
    }
    # ---- the rows right after the path, timed beside it; NOT part of `value`
    try:
        from echoglad_amd import evaluators as EV, losses as LS
        lg = out.detach().reshape(B * (out.shape[0] // B), 4).contiguous()
        n_rows = lg.shape[0] // B
        lv = LS.level_grids(args.frame, args.naux, args.main_only)
        yl = torch.zeros_like(lg)
        yl.view(B, n_rows, 4)[:, [st + (s // 2) * s + s // 2 for st, s in lv if st + s * s <= n_rows], :] = 1.0
        vl = torch.ones_like(lg)
        elm = LS.ExpectedLandmarkMSE(10, B, args.frame, args.naux, args.main_only)
        bce = LS.WeightedBCEWithLogitsLoss("none", 9000, 1)

        def loss_step():
            x = lg.detach().requires_grad_(True)
            (elm.compute(x, yl, vl) + bce.compute(x, yl, vl)).backward()

        if not args.main_only:
            pm = [torch.randn(B, C, 2 ** g, 2 ** g, device=device) for g in range(1, args.naux + 1)]
            pm.append(torch.randn(B, C, args.frame, args.frame, device=device))
            pack_ms = round(time_steps(lambda: ops.pack_levels(pm, B, N, 0), 20, 3), 4)
            result["before_path"] = {"pack_levels_ms": pack_ms,
                                     "pack_levels_GBs": round(2 * B * N * C * 4 / (pack_ms * 1e-3) / 1e9, 1),
                                     "note": "SURVEY f-1: NCHW level maps -> node-major [B*N,128] in one launch (reads + writes B*N*512 B)"}
            # the UNet variant's tail: relu(1x1 conv to 128 channels) of every decoder map fused into the packing
            chans = [512, 256, 128, 64, 32, 16, 8, 4][-(args.naux + 1):]
            fm = [torch.randn(B, c, m.shape[2], m.shape[3], device=device) for c, m in zip(chans, pm)]
            ws = [torch.randn(C, c, device=device) * (1.0 / c) ** 0.5 for c in chans]
            bs = [torch.zeros(C, device=device) for _ in chans]
            with torch.no_grad():
                fused_ms = round(time_steps(lambda: ops.conv1x1_relu_pack_levels(fm, ws, bs, B, N, 0), 20, 3), 4)
                unfused_ms = round(time_steps(lambda: ops.pack_levels(
                    [torch.relu(torch.nn.functional.conv2d(f, w.view(C, -1, 1, 1), b)) for f, w, b in zip(fm, ws, bs)], B, N, 0), 10, 2), 4)
            in_bytes = sum(f.numel() * 4 for f in fm)
            result["before_path"].update({
                "conv1x1_relu_pack_ms": fused_ms, "conv1x1_relu_then_pack_levels_ms": unfused_ms,
                "conv1x1_relu_pack_GBs": round((in_bytes + B * N * C * 4) / (fused_ms * 1e-3) / 1e9, 1),
                "note_fused": "models.py:707-710 + :726-756 in one launch (reads the decoder maps, writes B*N*512 B) next to torch "
                              "conv2d + relu per level followed by eg_pack_levels"})
            del pm, fm, ws, bs
            # SURVEY 8(d), context: frame -> logits END TO END through the UNet variant the reference's default.yml names
            # (echoglad_amd/examples.py: stock PyTorch-ROCm convolutions in front, the fused tail, then the stack above)
            if args.frame == 224 and args.naux == 7 and not args.no_other_configs:
                from echoglad_amd.examples import UNetNodeFeatureModel
                um = UNetNodeFeatureModel(**kw).to(device).eval()
                fr = torch.randn(B, 4, args.frame, args.frame, device=device)
                frs = [torch.randn(B, 4, args.frame, args.frame, device=device) for _ in range(3)]
                with torch.no_grad():
                    e2e = time_steps(lambda: um(x=fr, edge_index=edge_index), 10, 3)
                    front = time_steps(lambda: um.decoder_maps(fr), 10, 3)
                    mp = um.decoder_maps(fr)
                    tail = time_steps(lambda: um.pack_node_features_linear(mp, list(um.linears), B), 10, 3)
                    # ... and through model(x=...) with enable_hip_graph(True): the tail writes the model's static node-feature
                    # buffer, the stack replays ONE captured graph for every new batch of frames (3 different frame tensors in turn)
                    um.enable_hip_graph(True)
                    turn = [0]

                    def replayed():
                        turn[0] += 1
                        return um(x=frs[turn[0] % 3], edge_index=edge_index)

                    e2e_replay = time_steps(replayed, 12, 3)
                    captures = um.hip_graph_captures
                result["before_path"]["end_to_end"] = {
                    "frame_to_logits_ms": round(e2e, 4), "frames_s": round(B / (e2e * 1e-3), 1), "unet_front_end_ms": round(front, 4),
                    "tail_1x1conv_relu_pack_ms": round(tail, 4),
                    "frame_to_logits_ms_stack_replayed": round(e2e_replay, 4), "stack_graph_captures": captures,
                    "note": "context only (SURVEY 8d): UNet encoder / decoder on stock PyTorch-ROCm (MIOpen), not part of the hot path; "
                            "eager launches, batch " + str(B) + "; _stack_replayed: model(x=new frames) with enable_hip_graph(True)"}
                del frs
                del um, fr, mp
        result["after_path"] = {"landmark_decode_ms": round(time_steps(lambda: EV.decode_landmarks(lg, B, args.frame, yl, vl), 20, 3), 4),
                                "losses_fwd_bwd_ms": round(time_steps(loss_step, 20, 3), 4),
                                "note": "softmax-expected + hard-argmax landmark decode of the step's logits, and "
                                        "ExpectedLandmarkMSE + WeightedBCEWithLogits forward+backward, on the device"}
    except Exception as ex:                                   # never lose the bench line over the side measurement
        result["after_path"] = {"error": repr(ex)}
    if not args.no_cpu_baseline:
        cb, cpu_out, cpu_feats, cpu_ei = cpu_baseline(args, kw, model.state_dict())
        result["cpu_baseline"] = cb
        # parity of the measured path on the CPU sample (same inputs): logits + landmark indices
        from oracle import gnn_oracle as O
        with torch.no_grad():
            got = model.forward_nodes(cpu_feats.to(device), cpu_ei.to(device), args.cpu_frames)[0].cpu()
        result["parity"] = {"max_abs_err_vs_oracle": float((got - cpu_out).abs().max()),
                            "landmark_argmax_equal": bool(torch.equal(
                                O.landmark_argmax(got, args.cpu_frames, args.frame),
                                O.landmark_argmax(cpu_out, args.cpu_frames, args.frame)))}
    if world == 1 and not args.no_other_configs:
        del buf
        result["other_configs"] = other_configs(args, device)
    if train_leg is not None:
        result.setdefault("other_configs", {})[f"cfg4_train_dp{world}"] = train_leg
    emit_with_details(args, result)


def _short(text, n):
    return text if len(text) <= n else text[:n - 1] + "~"


def headline_of(result: dict, details_where: str) -> dict:
    """The stdout line: the contract's keys + roofline + cpu_baseline + parity + the summary numbers of the other BASELINE
    configs, short enough (<= ~1.5 KB) that the last 2000 characters of the run's output hold the WHOLE object.  Everything
    else (per-config entries, notes, before / after the path, per-rank diagnostics) goes to `details`."""
    h = {k: result[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "eager_ms_per_step",
                                "higher_is_better", "scaling", "vs_baseline", "dtype", "data") if k in result}
    cfg = result.get("config", {})
    h["config"] = {"workload": cfg.get("workload_short") or _short(cfg.get("workload", ""), 96), "global_batch": cfg.get("global_batch"),
                   "parallelism": cfg.get("parallelism", "").split(" ")[0], "launch": " ".join(cfg.get("launch", "").split(" ")[:2])}
    rf = result.get("roofline", {})
    h["roofline"] = {k: rf.get(k) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_ms")}
    if isinstance(rf.get("hbm"), dict):
        h["roofline"]["hbm_frac"] = rf["hbm"].get("frac")
    h["stack_mfma_frac"], h["stack_hbm_frac"] = result.get("stack_mfma_frac"), result.get("stack_hbm_frac")
    if "cpu_baseline" in result:
        cb = result["cpu_baseline"]
        h["cpu_baseline"] = {"value": cb["value"], "unit": cb["unit"], "cores": cb["cores"], "kind": cb["kind"],
                             "sample_batch": cb.get("sample_batch"),
                             "sample": cb.get("sample_short") or _short(cb["sample"], 40)}
    if "parity" in result:
        h["parity"] = dict(result["parity"], max_abs_err_vs_oracle=float(f"{result['parity']['max_abs_err_vs_oracle']:.3e}"))
    rp = result.get("repeats")
    h["repeats"] = None if not rp else dict(rp["ms_per_step"], n=rp["n"])
    d = result.get("distributed") or {}
    h["distributed"] = {k: d[k] for k in ("world_size", "backend", "share_device", "shard_digests_equal_single_rank") if k in d}
    oc = result.get("other_configs") or {}
    t = oc.get("cfg4_train")
    if isinstance(t, dict) and "ms_per_step" in t:
        h["cfg4_train"] = {"ms_per_step": t["ms_per_step"], "mfma_frac": t.get("mfma_frac"), "hbm_frac": t.get("hbm_frac"),
                           "traffic_bytes_per_step": t.get("traffic_bytes_per_step"),
                           "algorithmic_bytes_per_step": t.get("algorithmic_bytes_per_step")}
    brief = {}
    for k, v in oc.items():
        if isinstance(v, dict) and k not in ("cfg4_train", "cfg2_diagonal", "cfg2_connection_nodes"):
            brief[k] = v.get("ms_per_step", "error" if "error" in v else None)
            if k.startswith("cfg4_train_b1") and isinstance(v.get("hip_graph_replay"), dict):
                brief[k + "_replayed"] = v["hip_graph_replay"].get("ms_per_step")
    if brief:
        h["other_ms_per_step"] = brief
    h["details"] = details_where
    return h


def emit_with_details(args, result: dict) -> None:
    """stdout: the headline line.  The full result: --details file (default: gpurun_out/bench_details.json, which gpurun merges
    back; a short pointer on stderr), stderr (one JSON line, written BEFORE the stdout line), or inline (one long stdout line)."""
    if args.details == "inline":
        emit(result)
        return
    where = "stderr"
    if args.details == "file":
        path = args.details_path if os.path.isabs(args.details_path) else os.path.join(ROOT, args.details_path)
        try:
            os.makedirs(os.path.dirname(path), exist_ok=True)
            with open(path, "w") as f:
                json.dump(result, f)
            where = os.path.relpath(path, ROOT)
        except OSError as ex:
            where = "stderr"
            print(f"bench.py: cannot write {path} ({ex!r}); details follow on stderr", file=sys.stderr)
    if where == "stderr":
        sys.stderr.write(json.dumps({"bench_details": result}) + "\n")
        sys.stderr.flush()
    emit(headline_of(result, where))


if __name__ == "__main__":
    main()
