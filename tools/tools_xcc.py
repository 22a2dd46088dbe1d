import os, sys, ctypes as ct
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from echoglad_amd import _lib
lib = _lib.load()
for nb in (64, 768, 2048):
    out = torch.zeros(2 * nb, dtype=torch.int32, device="cuda")
    lib.eg_debug_xcc(ct.c_void_p(out.data_ptr()), nb, ct.c_void_p(torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    x = out[:nb].cpu().numpy()
    print("nblocks", nb, "xcc histogram", np.bincount(x, minlength=8).tolist())
    print("  first 32 xcc ids:", x[:32].tolist())
    print("  blockIdx%8 == xcc (up to a rotation)?", [int(((np.arange(nb) + r) % 8 == x).mean() * 100) for r in range(8)])
