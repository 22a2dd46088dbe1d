"""The heads' backward (eg_classifier_bwd / eg_classifier_bwd_sums) alone at BASELINE configs[3]'s shape: time per call and
sha256 digests of everything it returns -- run once per setting of EG_FB_STAGGER (the knob is read once per process) and
compare the digests: the staggered kernel claims bit-identical results.
    B=32 python tools/tools_fb.py            (small shapes for the bit comparison: SHAPES=1)"""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from echoglad_amd import ops

DEV = "cuda:0"


def digest(t):
    return hashlib.sha256(t.detach().cpu().contiguous().numpy().tobytes()).hexdigest()[:16]


def params(seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    f = lambda *s: ((torch.rand(*s, generator=g) - 0.5) * 0.3).to(DEV)
    return dict(w1=f(128, 128), b1=f(128), gamma1=f(128) + 1, beta1=f(128), w2=f(4, 16, 32), b2=f(64), gamma2=f(64) + 1, beta2=f(64),
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


def case(B, n, lo, n_valid, timing):
    g = torch.Generator(device="cpu").manual_seed(B * 1000 + n)
    h = torch.randn(B * n, 128, generator=g).to(DEV)
    lz = torch.randn(B * n, 128, generator=g).to(DEV)
    P = params(n)
    lbn = torch.cat([torch.randn(128, generator=g) * 0.1, torch.rand(128, generator=g) + 0.5, torch.zeros(256)]).to(DEV)
    lgamma, lbeta = (torch.rand(128, generator=g) + 0.5).to(DEV), (torch.randn(128, generator=g) * 0.1).to(DEV)
    logits, z1, z2, bn = ops.classifier_train_fwd(h, B, n, lo, n_valid, P, False)
    dl = torch.randn(logits.shape, generator=g).to(DEV)
    dh, grads = ops.classifier_bwd(dl, h, B, n, lo, n_valid, P, z1, z2, bn, True)
    dh2, grads2, sums = ops.classifier_bwd(dl, h, B, n, lo, n_valid, P, z1, z2, bn, True, layer=(lz, lbn, lgamma, lbeta, True, 0.5, 77))
    torch.cuda.synchronize()
    print(f"B={B} n={n} lo={lo} n_valid={n_valid}: dh {digest(dh)} grads {digest(grads)} | sums form: dh {digest(dh2)} grads {digest(grads2)} "
          f"sums {digest(sums) if sums is not None else None}  finite {bool(torch.isfinite(dh).all())}")
    if timing:
        print("  heads backward              %.3f ms" % t(lambda: ops.classifier_bwd(dl, h, B, n, lo, n_valid, P, z1, z2, bn, True)))
        print("  heads backward + layer sums %.3f ms" % t(lambda: ops.classifier_bwd(dl, h, B, n, lo, n_valid, P, z1, z2, bn, True,
                                                                                      layer=(lz, lbn, lgamma, lbeta, True, 0.5, 77))))


print("EG_FB_STAGGER =", os.environ.get("EG_FB_STAGGER", "(default)"))
if not os.environ.get("FB_TIMING_ONLY"):
    for B, n, lo, nv in ((2, 341, 0, 341), (3, 1000, 7, 901), (1, 72024, 0, 72020), (2, 5000, 4, 4993), (5, 70, 3, 64)):
        case(B, n, lo, nv, False)
case(int(os.environ.get("B", "32")), 72024, 0, 72020, True)
