"""Diagnostic: soak test for launch-to-launch determinism of the fused layer kernels at FULL grid size (the register-reuse
hazard of DESIGN.md section 5 item 14 only showed with every CU busy).  Runs each form `reps` times and compares bits.
usage (GPU box, repo root): python3 tools/tools_determinism.py [reps]"""
import os
import sys

R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "tests"))
sys.path.insert(0, os.path.join(R, "tests", "golden"))
import torch  # noqa: E402

from echoglad_amd import ops  # noqa: E402
from fixtures_util import synthetic_node_feats  # noqa: E402

DEV = "cuda:0"


def soak(frame, naux, B, mode, reps, main_only=False):
    g = ops.Graph.topo(frame, naux, main_only)
    g.set_precision(mode)
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
    g.set_precision("f32")
    print(f"{frame}x{frame} naux={naux} main_only={main_only} B={B} {mode}: {bad} of {reps - 1} launches differ from the first")
    return bad


if __name__ == "__main__":
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    total = 0
    for mode in ("f32", "bf16x3", "bf16x6"):
        total += soak(224, 7, 8, mode, reps)
        total += soak(224, 7, 32, mode, reps, main_only=True)
        total += soak(448, 8, 8, mode, max(reps // 4, 2))
    sys.exit(1 if total else 0)
