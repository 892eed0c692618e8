"""Per-launch time of the chain's panel-update shape (lower trapezoid m x n, K-major operands, 32 x 32 / 64 x 64 tiles) as a function of
K: how much of the 8-10 us is fixed cost (launch, first loads, epilogue) and how much the k loop.  python tools/small_gemm_latency.py"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from fidelityfusion_amd import _lib

h = _lib.handle(0)
_lib.bind_stream(h, 0)
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
A = torch.rand((9000, 128), generator=g, device=dev, dtype=torch.float64) - 0.5
Cm = torch.zeros((9000, 640), device=dev, dtype=torch.float64)
p = lambda t: C.c_void_p(t.data_ptr())
torch.cuda.synchronize()
for m in (1000, 3000, 8000):
    for n in (128, 384, 512):
        row = []
        for k in (4, 16, 64, 128):
            def run(reps):
                for _ in range(reps):
                    rc = _lib.lib.ffgp_gemm(h, 0, 0, 1, 0, p(A), 128, p(A), 128, p(Cm), 640, m, n, k, -1.0, 1.0)
                    assert rc == 0
            run(20)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run(200)
            torch.cuda.synchronize()
            row.append("k=%3d %5.2f us" % (k, (time.perf_counter() - t0) / 200 * 1e6))
        print("m=%4d n=%3d  %s" % (m, n, "   ".join(row)), flush=True)
