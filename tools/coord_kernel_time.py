"""Time of the coordinate update's single-workgroup launches at small batches (hip events around 300 back-to-back calls; the kernels are
longer than the launch rate, so this is kernel + hand-over).  usage: ECHOGLAD_LIB=... python3 tools/coord_kernel_time.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from echoglad_amd import ops
from echoglad_amd.nn import _MLP_NAMES
from echoglad_amd.topology import TopologySpec, get_topology
_T = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests")
sys.path[:0] = [_T, os.path.join(_T, "golden"), os.path.dirname(_T)]
from gpu_util import model_pair, rand_rows

DEV = torch.device("cuda", 0)
frame, naux = 224, 7
topo = get_topology(TopologySpec(frame, naux, False, True))
n, main_base, coord_base = topo.num_nodes, topo.main.base, topo.coord_base
hip, _ = model_pair(16, 3, 1, coord=True, seed=3)
mlp = hip.node_coordinate_mlp[0]
hip.train()
cfg, params = hip._coord_mlp_train_cfg(mlp)
P = dict(cfg)
P.update({k: q.detach().contiguous() for k, q in zip(_MLP_NAMES, params)})
P.update(seed1=101, seed2=202)


def timed(fn, it=300):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / it


for B in (1, 8, 16):
    rs = np.random.RandomState(5 + B)
    h = rand_rows(B * n, seed=21).to(DEV)
    c0 = torch.from_numpy(rs.uniform(0.5, frame - 1.5, (B * 4, 2)).astype(np.float32)).to(DEV)
    (new, _), lm, saved = ops.coord_update_fwd(h, c0, B, n, coord_base, main_base, P, True, frame, True)
    dx = rand_rows(B * n, seed=23).to(DEV)
    dnew = torch.from_numpy(rs.standard_normal((B * 4, 2)).astype(np.float32)).to(DEV)
    lz = rand_rows(B * n, seed=25).to(DEV)
    lbn = torch.cat([lz.mean(0), 1.0 / lz.std(0), torch.ones(128, device=DEV) * 0.9, torch.zeros(128, device=DEV) + 0.05]).contiguous()
    t_f = timed(lambda: ops.coord_update_fwd(h, c0, B, n, coord_base, main_base, P, True, frame, True))
    t_f0 = timed(lambda: ops.coord_update_fwd(h, c0, B, n, coord_base, main_base, P, True, frame, True, resample=False))
    t_b = timed(lambda: ops.coord_update_bwd(dx, dnew, h, new, lm, c0, B, n, coord_base, main_base, P, frame, saved, True, lower=(lz, lbn, True, 0.3, 999)))
    t_b0 = timed(lambda: ops.coord_update_bwd(dx, dnew, h, new, lm, c0, B, n, coord_base, main_base, P, frame, saved, True))
    t_m = timed(lambda: ops.coord_mlp_bwd(dnew, lm, c0, B, P, frame, saved, True, True, out_rows=(dx, n, coord_base), accumulate=True))
    print(f"lib={os.environ.get('ECHOGLAD_LIB', 'base')} B={B}: fwd+resample {t_f:.1f} us, fwd {t_f0:.1f}, bwd+resample+sums {t_b:.1f}, bwd+resample {t_b0:.1f}, "
          f"mlp bwd {t_m:.1f}", flush=True)
