"""Diagnostic: mean duration of the dX launches (k_gcn_layer_ps MODE 2 / 3) and of the whole step inside real training steps at configs[3],
for same-box A/Bs of library builds (ECHOGLAD_LIB) and of EG_SUMS_DOWN.  usage: python3 tools/dx_ab.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from echoglad_amd import ops
dev = torch.device("cuda", 0)
step, topo = bench.train_workload(224, 7, 3, 32, dev, 1, 0)
for _ in range(4):
    step()
torch.cuda.synchronize()
with ops.layer_timing(256) as tm:
    for _ in range(6):
        step()
    torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    step()
torch.cuda.synchronize()
print(f"lib={os.environ.get('ECHOGLAD_LIB', 'base')} EG_SUMS_DOWN={os.environ.get('EG_SUMS_DOWN', '1')}: dX launch {1e3 * tm.mean_ms('ps_dx'):.1f} us, "
      f"train-forward launch {1e3 * tm.mean_ms('ps_train_fwd'):.1f} us, step {1e3 * (time.perf_counter() - t0) / 20:.3f} ms")
