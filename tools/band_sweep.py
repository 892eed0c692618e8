"""GEMM tile order: band height 2^band_log2 tile rows -- SYRK standalone at a few sizes, then the whole C3 forward."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _devlib  # noqa: F401,E402  (development build: these switches are not in the shipped library)
import torch

from fidelityfusion_amd import _lib
from tools.gemm_bench import p, timeit

h = _lib.handle(0)
_lib.bind_stream(h, 0)
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
k = 512
A = torch.rand((16384, k), generator=g, device=dev, dtype=torch.float64) - 0.5
Cm = torch.zeros((16384, 16384), device=dev, dtype=torch.float64)
for bl in (1, 2, 3, 4, 5):
    _lib.set_option("band_log2", bl, 0)
    row = []
    for m in (15872, 12288, 8192, 4096):
        fn = lambda: _lib.lib.ffgp_gemm(h, 0, 0, 1, 0, p(A), k, p(A), k, p(Cm), 16384, m, m, k, -1.0, 1.0)
        fn()
        tmin, _ = timeit(fn, rounds=7)
        row.append("m=%d %.3f ms %.1f TF/s" % (m, tmin, m * (m + 1) * k / tmin / 1e9))
    B = torch.rand((8192, 2048), generator=g, device=dev, dtype=torch.float64)
    fn = lambda: _lib.lib.ffgp_gemm(h, 0, 0, 0, 0, p(B), 2048, p(B), 2048, p(Cm), 16384, 8192, 8192, 2048, 1.0, 0.0)
    fn()
    tmin, _ = timeit(fn, rounds=5)
    row.append("gemm 8192^2x2048 %.1f TF/s" % (2 * 8192.0 * 8192 * 2048 / tmin / 1e9))
    print("band 2^%d: %s" % (bl, " | ".join(row)))
_lib.set_option("band_log2", 3, 0)
