"""timing-only ablations of ffgp_potrf_diag128_v4 (libraries built with -DFFGP_D4_DBG=1 / 2: helpers idle in [A] / wave 0 without its
pivots; results are wrong): n = 128 factorisations back to back, us per call by wall clock.  FFGP_LIB=... python tools/diag4_ablate.py"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from fidelityfusion_amd import _lib

dev = torch.device("cuda:0")
h = _lib.handle(0)
lib = _lib.lib
n = 128
g = torch.Generator(device=dev).manual_seed(n)
R = torch.randn(n, 64, dtype=torch.float64, device=dev, generator=g)
S = R @ R.T / 64.0
S.diagonal().add_(2.0)
Ws = [S.clone() for _ in range(200)]
for rep in range(3):
    for W in Ws:
        W.copy_(S)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for W in Ws:
        lib.ffgp_potrf_rows(h, C.c_void_p(W.data_ptr()), n, n, n)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / len(Ws) * 1e6
print(os.environ.get("FFGP_LIB", "libffgp.so"), "n=128 potrf call %.1f us (synchronous call: kernel + launch + status read-back)" % dt)
