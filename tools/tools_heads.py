"""Timing of the train-mode classifier heads alone (forward, backward) at BASELINE configs[3]'s shape: 224/7 + coordinate nodes,
batch 32.  Run under rocprofv3 --kernel-trace --stats for the per-kernel split."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from echoglad_amd import ops

DEV = "cuda:0"
B = int(os.environ.get("B", "32"))
n, n_valid = 72024, 72020
torch.manual_seed(0)
h = torch.randn(B * n, 128, device=DEV)
f = lambda *s: (torch.rand(*s, device=DEV) - 0.5) * 0.3
P = dict(w1=f(128, 128), b1=f(128), gamma1=f(128) + 1, beta1=f(128), w2=f(4, 16, 32), b2=f(64), gamma2=f(64) + 1, beta2=f(64),
         w3=f(64), b3=f(4), running_mean1=None, running_var1=None, running_mean2=None, running_var2=None, eps1=1e-5, eps2=1e-5,
         momentum1=None, momentum2=None, p1=0.5, p2=0.5, seed1=11, seed2=12)


def t(fn, it=10, warm=2):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(it):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it


logits, z1, z2, bn = ops.classifier_train_fwd(h, B, n, 0, n_valid, P, False)
dl = torch.randn_like(logits)
print("heads forward   %.3f ms" % t(lambda: ops.classifier_train_fwd(h, B, n, 0, n_valid, P, False)))
print("heads backward  %.3f ms" % t(lambda: ops.classifier_bwd(dl, h, B, n, 0, n_valid, P, z1, z2, bn, True)))
