"""ffgp_potrf_diag128_v4 (two barriers per 16-column stage) against v3: factor and Dinv against LAPACK at several sizes (partial
blocks, not-PD pivots), then interleaved timings of the forward likelihood.  python tools/diag4_ab.py"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from bench import synthetic_xy
from fidelityfusion_amd import _lib
from fidelityfusion_amd import functional as F

dev = torch.device("cuda:0")
h = _lib.handle(0)
lib = _lib.lib


def spd(n, seed=0):
    g = torch.Generator(device=dev).manual_seed(seed)
    R = torch.randn(n, 64, dtype=torch.float64, device=dev, generator=g)
    S = R @ R.T / 64.0
    S.diagonal().add_(2.0)
    return S


def potrf(S):
    n = S.shape[0]
    ld = (n + 1) // 2 * 2 + 2
    W = torch.zeros((n, ld), dtype=torch.float64, device=dev)
    W[:, :n] = S
    rc = lib.ffgp_potrf_rows(h, C.c_void_p(W.data_ptr()), n, n, ld)
    torch.cuda.synchronize()
    return rc, W[:, :n]


worst = 0.0
for n in (1, 5, 16, 17, 31, 100, 128, 129, 200, 255, 256, 300, 640, 1000, 1537, 3000, 4300):
    S = spd(n, n)
    Lref = torch.linalg.cholesky(S)
    for v4 in (0, 1):
        _lib.set_option("diag_v4", v4, 0)
        rc, W = potrf(S)
        err = float((W.tril() - Lref).abs().max() / Lref.abs().max())
        worst = max(worst, err)
        assert rc == 0 and err < 1e-11, (n, v4, rc, err)
    # a failing pivot is reported at its index
    for bad in sorted({1, max(1, n // 2), n}):
        S2 = S.clone()
        piv = float(Lref[bad - 1, bad - 1]) ** 2
        S2[bad - 1, bad - 1] -= piv + 0.5
        for v4 in (1,):
            _lib.set_option("diag_v4", v4, 0)
            rc, _ = potrf(S2)
            assert rc == bad, (n, bad, rc, v4)
print("factor vs LAPACK ok, worst rel err %.2e" % worst)
# the gradient path reads Dinv (TRTRI from the block inverses): likelihood + gradients against v3
for n in (300, 1000, 2500):
    X, Y = synthetic_xy(n, 5, 2, seed=1)
    Xd, Yd = torch.tensor(X, device=dev), torch.tensor(Y, device=dev)
    out = {}
    for v4 in (0, 1):
        _lib.set_option("diag_v4", v4, 0)
        w = torch.ones(5, dtype=torch.float64, device=dev, requires_grad=True)
        amp = torch.ones(1, dtype=torch.float64, device=dev, requires_grad=True)
        dadd = torch.tensor([0.3], dtype=torch.float64, device=dev, requires_grad=True)
        v = F.nlml(Xd, Yd, w, amp, diag_add=dadd, clamp=1e-30)
        v.backward()
        out[v4] = (v.detach().clone(), w.grad.clone(), amp.grad.clone(), dadd.grad.clone())
    for vv in (1,):
        for a, b in zip(out[0], out[vv]):
            assert float((a - b).abs().max() / b.abs().max()) < 1e-10, (n, vv, a, b)
print("likelihood + gradients agree with v3")
for n, D in ((128, 5), (1024, 8), (2048, 8), (4096, 8), (8192, 8), (16384, 16)):
    X, Y = synthetic_xy(n, D, 1, seed=0)
    Xd, Yd = torch.tensor(X, device=dev), torch.tensor(Y, device=dev)
    w = torch.ones(D, dtype=torch.float64, device=dev)
    amp = torch.ones(1, dtype=torch.float64, device=dev)
    dadd = torch.tensor([np.exp(-1.0) + 1e-6], dtype=torch.float64, device=dev)
    res = {0: [], 1: []}
    reps = 20 if n <= 4096 else 6
    for rnd in range(4):
        for v4 in (0, 1):
            _lib.set_option("diag_v4", v4, 0)
            with torch.no_grad():
                for _ in range(2):
                    F.nlml(Xd, Yd, w, amp, diag_add=dadd, clamp=1e-30)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(reps):
                    F.nlml(Xd, Yd, w, amp, diag_add=dadd, clamp=1e-30)
                torch.cuda.synchronize()
            res[v4].append((time.perf_counter() - t0) / reps * 1e3)
    print("N=%6d forward: v3 %.3f ms   v4 %.3f ms (%+.1f %%)" % (n, min(res[0]), min(res[1]), (min(res[1]) / min(res[0]) - 1) * 100), flush=True)
_lib.set_option("diag_v4", 1, 0)
