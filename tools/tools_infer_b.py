"""Diagnostic: ms per step of the inference stack (HIP-graph replay) at a few batch sizes, for same-box A/Bs of library builds
(ECHOGLAD_LIB honoured).  usage: python3 tools/tools_infer_b.py [B ...]"""
import gc
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402

dev = torch.device("cuda", 0)
for B in [int(a) for a in sys.argv[1:]] or [1, 2, 8]:
    model, _, topo, feats, ei, step = bench.infer_workload(224, 7, 3, False, B, dev, 0)
    best = min(bench.time_steps(step, iters=200, warm=30) for _ in range(3))
    print(f"B = {B:2d}: {best:7.4f} ms   {1e3 * best / B:7.1f} us / frame", flush=True)
    del model, feats, ei, step
    gc.collect(); torch.cuda.empty_cache()
