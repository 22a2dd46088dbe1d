"""Diagnostic: where the HOST spends a training step (cProfile over N steps of bench.py's train workload, batch EG_B [1]).
At batch 1 -- the reference's own setting (configs/default.yml:27) -- the step is host-bound: ~135 launches and the autograd
graph around them take longer to issue than the GPU needs to run them."""
import cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

B = int(os.environ.get("EG_B", "1"))
N = int(os.environ.get("EG_STEPS", "60"))
dev = torch.device("cuda", 0)
step, topo = bench.train_workload(224, 7, 3, B, dev, 1, 0)
for _ in range(5):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(N):
    step()
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"batch {B}: {1e3 * t_all / N:.3f} ms per step wall, host issue time {1e3 * t_issue / N:.3f} ms per step")
pr = cProfile.Profile()
pr.enable()
for _ in range(N):
    step()
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(45)
print(s.getvalue()[:9000])
