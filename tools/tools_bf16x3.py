"""Diagnostic: the opt-in bf16x3 / bf16x6 layer products (eg_graph_set_precision) against the exact fp32 path on the headline
workload: max-abs error of the logits, arg-max equality per (frame, level, channel), step time.
usage (GPU box, repo root): python3 tools/tools_bf16x3.py [batch]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import torch  # noqa: E402

import bench  # noqa: E402
from echoglad_amd import nn as egnn  # noqa: E402


def run(B, precision):
    model, kw, topo, feats, ei, step = bench.infer_workload(224, 7, 3, False, B, torch.device("cuda", 0), 0)
    graph, _ = model._resolver.resolve(ei, feats.shape[0])
    graph.set_precision(precision or "f32")
    for _ in range(20):
        out = step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(100):
        out = step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 10.0
    return out.clone(), ms, topo


if __name__ == "__main__":
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    from echoglad_amd import losses
    ref, ms_ref, topo = run(B, None)
    n = topo.num_valid_nodes
    print(f"f32   : {ms_ref:.4f} ms/step ({B / ms_ref * 1e3:.0f} frames/s), max |logit| {ref.abs().max().item():.3f}")
    for mode in ("bf16x3", "bf16x6"):
        got, ms_bf, _ = run(B, mode)
        err = (got - ref).abs().max().item()
        same = True
        for st, s in losses.level_grids(224, 7, False):
            a = ref.view(B, n, 4)[:, st:st + s * s, :].argmax(1)
            b = got.view(B, n, 4)[:, st:st + s * s, :].argmax(1)
            same = same and bool((a == b).all())
        print(f"{mode}: {ms_bf:.4f} ms/step ({B / ms_bf * 1e3:.0f} frames/s), max|logit diff vs f32| = {err:.3e}, per-level argmax equal: {same}")
    run(B, None)                                   # leave the shared handle in the exact mode
