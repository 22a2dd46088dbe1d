"""Diagnostic: soak test for launch-to-launch determinism of the fused layer kernels at FULL grid size (the register-reuse
hazard of DESIGN.md section 5 item 14 only showed with every CU busy).  Runs each form `reps` times and compares bits; then a whole training step `reps / 10` times.
usage (GPU box, repo root): python3 tools/tools_determinism.py [reps]"""
import os
import sys

R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "tests"))
sys.path.insert(0, os.path.join(R, "tests", "golden"))
import torch  # noqa: E402

from echoglad_amd import ops  # noqa: E402
from echoglad_amd.synthetic import synthetic_node_feats  # noqa: E402

DEV = "cuda:0"


def soak(frame, naux, B, mode, reps, main_only=False, diag=False, conn=False):
    g = ops.Graph.topo(frame, naux, main_only, diag_main=diag, diag_aux=diag and not main_only, use_connection_nodes=conn)
    n = g.num_nodes
    x = synthetic_node_feats(B * n, 128, seed=1).to(DEV)
    w = (synthetic_node_feats(128, 128, seed=2) * 0.1).to(DEV)
    sc, sh = torch.ones(128, device=DEV), torch.zeros(128, device=DEV)
    chained = g.kidsum_rows > 0
    ka = ops.new_kidsum(g, B) if chained else None
    kb = ops.new_kidsum(g, B) if chained else None
    first, bad = None, 0
    for _ in range(reps):
        if chained:
            h = ops.gcn_layer_fwd(g, B, x, w, sc, sh, x, relu=True, kidsum_out=ka)
            out = ops.gcn_layer_fwd(g, B, h, w, sc, sh, h, relu=True, kidsum_in=ka, kidsum_out=kb)
        else:
            out = ops.gcn_layer_fwd(g, B, x, w, sc, sh, x, relu=True)
        if first is None:
            first = out.clone()
        elif not torch.equal(out, first):
            bad += 1
    print(f"{frame}x{frame} naux={naux} main_only={main_only} diagonal={diag} connection_nodes={conn} B={B} {mode}: {bad} of {reps - 1} "
          "launches differ from the first", flush=True)
    return bad


def soak_csr(reps, B=8):
    """configs[1]'s batch as a plain edge_index: the CSR handle's clustered tiles (round 5) and, on the stencil handle, the
    unchained producer/consumer launch that plain calls take since round 5."""
    from echoglad_amd.topology import TopologySpec, get_topology
    topo = get_topology(TopologySpec(224, 7, False, False))
    ei = torch.from_numpy(topo.batched_edge_index(B)).to(DEV)
    n = B * topo.num_nodes
    x = synthetic_node_feats(n, 128, seed=1).to(DEV)
    w = (synthetic_node_feats(128, 128, seed=2) * 0.1).to(DEV)
    total = 0
    for label, g, gb in (("CSR handle (clustered tiles)", ops.Graph.csr(ei, n), 1), ("stencil handle, plain call", ops.Graph.topo(224, 7), B)):
        first, bad = None, 0
        for _ in range(reps):
            out = ops.gcn_layer_fwd(g, gb, x, w, None, None, x, relu=True)
            if first is None:
                first = out.clone()
            elif not torch.equal(out, first):
                bad += 1
        print(f"224x224 naux=7 B={B} {label}: {bad} of {reps - 1} launches differ from the first", flush=True)
        total += bad
    return total


def soak_train(reps):
    """One configs[3] training step (224/7 + coordinate graph, batch 32, dropout 0.5) repeated from the same state and host
    RNG seed: logits, coordinates and every parameter gradient must come out bit-identical."""
    import copy
    from echoglad_amd.synthetic import initial_coords
    from gpu_util import graph_tensors, model_pair
    frame, naux, B = 224, 7, 32
    hip, _ = model_pair(frame, naux, 3, coord=True, seed=17)
    hip.train()
    topo, ei, nt, bi = graph_tensors(frame, naux, B, coord=True)
    x = synthetic_node_feats(B * topo.num_nodes, 128, seed=31).to(DEV)
    coords0 = initial_coords(B, frame).to(DEV)
    eid = ei.to(DEV)
    state = copy.deepcopy(hip.state_dict())
    first, bad = None, 0
    for _ in range(reps):
        hip.load_state_dict(state)
        hip.zero_grad(set_to_none=True)
        torch.manual_seed(5)
        logits, c = hip.forward_nodes(x, eid, B, coords0.clone())
        ((logits ** 2).mean() + (c ** 2).mean() * 1e-3).backward()
        cur = [logits.detach().clone(), c.detach().clone()] + [p.grad.detach().clone() for p in hip.parameters()]
        if first is None:
            first = cur
        elif not all(torch.equal(a, b) for a, b in zip(cur, first)):
            bad += 1
    print(f"train step 224/7+coord B=32: {bad} of {reps - 1} repetitions differ from the first")
    return bad


def soak_graphed(replays, runs=3):
    """engine.GraphedTrainStep on configs[3] at batch 32 (bench.py's training workload, capturable fused Adam): `runs` independent
    captures from the same initial state and host seed, `replays` replays each -- loss of the last replay and every parameter must
    come out bit-identical (the graph's epoch node gives replay k the masks of epoch k in every run)."""
    import gc
    import hashlib
    import bench
    dev = torch.device("cuda", 0)
    digests = []
    for _ in range(runs):
        ops.dropout_epoch_set(0)
        torch.manual_seed(1234)
        step, _ = bench.train_workload(224, 7, 3, 32, dev, 1, 0, capturable=True)
        torch.manual_seed(99)
        g = step.graphed(1)
        for _ in range(replays):
            out = g()
        torch.cuda.synchronize()
        h = hashlib.sha256()
        h.update(out[0].detach().cpu().numpy().tobytes())
        for q in g.optimizer.param_groups[0]["params"]:
            h.update(q.detach().cpu().numpy().tobytes())
        digests.append(h.hexdigest()[:16])
        del g, step
        gc.collect(); torch.cuda.empty_cache()
    ops.dropout_epoch_set(0)
    bad = sum(d != digests[0] for d in digests[1:])
    print(f"graphed train step 224/7+coord B=32: {runs} captures x {replays} replays (+ 1 eager warm-up step each), digests {digests}: "
          f"{bad} of {runs - 1} runs differ from the first")
    return bad


if __name__ == "__main__":
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    total = 0
    for mode in ("f32",):
        total += soak(224, 7, 8, mode, reps)
        total += soak(224, 7, 32, mode, reps, main_only=True)
        total += soak(448, 8, 8, mode, max(reps // 4, 2))
        total += soak(224, 7, 8, mode, reps, diag=True)                 # 'grid-diagonal' levels (round 4)
        total += soak(224, 7, 8, mode, reps, conn=True)                 # connection nodes: pre-pass + stencil
        total += soak(224, 7, 8, mode, max(reps // 2, 2), diag=True, conn=True)
    total += soak_csr(reps)
    total += soak_train(max(reps // 10, 3))
    total += soak_graphed(max(reps // 10, 3))
    sys.exit(1 if total else 0)
