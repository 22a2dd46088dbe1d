"""Diagnostic: ms per step of the inference stack (HIP-graph replay) and of one training step by batch size at configs[1] /
configs[3]'s shape -- the fixed cost per step (launches, prologues, pipeline fill / drain; host time for training) against the
cost per frame.  usage (GPU box, repo root): python3 tools/tools_batch_sweep.py"""
import gc
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402

dev = torch.device("cuda", 0)
print("# inference, 224x224 / 7 aux levels / 3 layers, eval, HIP-graph replay (ms per step, us per frame, frames/s)")
rows = []
for B in (1, 2, 4, 8, 16, 32):
    model, _, topo, feats, ei, step = bench.infer_workload(224, 7, 3, False, B, dev, 0)
    ms = bench.time_steps(step, iters=100, warm=20)
    rows.append((B, ms))
    print(f"B = {B:2d}: {ms:7.4f} ms   {1e3 * ms / B:7.1f} us / frame   {B / (ms * 1e-3):8.0f} frames/s", flush=True)
    del model, feats, ei, step
    gc.collect(); torch.cuda.empty_cache()
b = (rows[-1][1] - rows[0][1]) / (rows[-1][0] - rows[0][0])
print(f"# least fixed cost: t(B) ~ {rows[0][1] - b:.3f} ms + {b:.4f} ms * B  (from B = 1 and B = 32)")
print("# training step, 224x224 / 7 aux levels + coordinate graph, dropout 0.5, 3 losses, fused Adam (ms per step; host issue time | engine.GraphedTrainStep)")
for B in (1, 2, 4, 8, 16, 32):
    step, topo = bench.train_workload(224, 7, 3, B, dev, 1, 0)
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    n = 30
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    del step
    gc.collect(); torch.cuda.empty_cache()
    g = bench.graphed_train_ms(bench.argparse.Namespace(layers=3), dev, B, n=30)          # the same step replayed from ONE HIP graph
    print(f"B = {B:2d}: {1e3 * t_all / n:7.3f} ms   host issue {1e3 * t_issue / n:6.3f} ms   {B / (t_all / n):7.0f} frames/s"
          f"   | HIP-graph replay {g.get('ms_per_step', float('nan')):7.3f} ms   {g.get('frames_s', float('nan')):7.0f} frames/s", flush=True)
