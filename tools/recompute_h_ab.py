"""cfg4_train (224/7, batch 32, coordinate graph) step time with nn.ROUTES.heads_recompute_h on / off, interleaved, same process."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from echoglad_amd import nn as egnn

dev = torch.device("cuda:0")
step, topo = bench.train_workload(224, 7, 3, 32, dev, 1, 0)


def run(k):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(k):
        step()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / k


for rep in range(4):
    for on in (True, False):
        egnn.ROUTES.heads_recompute_h = on
        run(3)
        print(f"rep {rep} heads_recompute_h={int(on)}: {run(20):.3f} ms/step", flush=True)
