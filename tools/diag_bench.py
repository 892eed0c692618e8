"""Diagonal-block kernel A/B: the barrier version (diag_v2 = 0) against the pipelined one (1; 2 = helpers kept off wave 0's
SIMD).  Times ffgp_potrf at n = 128 (one block: launch + kernel), a chain of 32 blocks (n = 4096 with nothing but the chain
visible in the small trailing updates) and the C2 / N = 8192 / C3 factorisations, and checks L L^T = A each time."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _devlib  # noqa: F401,E402  (development build: these switches are not in the shipped library)
import torch

from fidelityfusion_amd import _lib

dev = torch.device("cuda:0")
h = _lib.handle(0)
lib = _lib.lib


def spd(n, seed=0):
    g = torch.Generator(device=dev).manual_seed(seed)
    X = torch.rand((n, 8), generator=g, device=dev, dtype=torch.float64)
    d2 = torch.cdist(X, X) ** 2
    return torch.exp(-0.5 * d2) + 0.37 * torch.eye(n, device=dev, dtype=torch.float64)


FACTORS = {}


def run(n, mode, reps):
    _lib.set_option("diag_v2", mode, 0)
    A = spd(n)
    W = A.clone()
    rc = lib.ffgp_potrf(h, C.c_void_p(W.data_ptr()), n, n)
    assert rc == 0, rc
    L = torch.tril(W)
    FACTORS[(n, mode)] = L.clone()
    err = float((L @ L.T - A).abs().max())
    best = 1e9
    for _ in range(reps):
        W.copy_(A)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        lib.ffgp_potrf(h, C.c_void_p(W.data_ptr()), n, n)
        best = min(best, time.perf_counter() - t0)
    return best * 1e6, err


for n in (128, 256, 1024, 4096, 8192, 16384):
    out = []
    for mode in (0, 3, 1, 4):   # 0: barrier version, 3: round-3 pipelined kernel, 1: the same with the DP-ALU DPP pivot step, 4: round-4 kernel (default)
        us, err = run(n, mode, 20 if n <= 4096 else 5)
        out.append("v2=%d: %9.1f us (|LL^T-A| %.1e)" % (mode, us, err))
    dmax = float((FACTORS[(n, 4)] - FACTORS[(n, 3)]).abs().max())
    print("potrf n=%5d  " % n + "   ".join(out) + "   max |L(4) - L(3)| = %.1e" % dmax, flush=True)
    FACTORS.clear()
_lib.set_option("diag_v2", 4, 0)
