"""Cost of taking CUs away from the trailing update: the SYRK kernel on a CU-masked stream (development probe)."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from fidelityfusion_amd import _lib

hip = C.CDLL("libamdhip64.so")
h = _lib.handle(0)
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
k = 512
A = torch.rand((16384, k), generator=g, device=dev, dtype=torch.float64) - 0.5
Cm = torch.zeros((16384, 16384), device=dev, dtype=torch.float64)
torch.cuda.synchronize()


def masked_stream(drop_per_xcc):
    m = (C.c_uint32 * 8)(*([0xffffffff] * 8))
    for x in range(8):
        for c in range(drop_per_xcc):
            bit = c * 8 + x          # bit i <-> (XCC i % 8, CU i // 8)
            m[bit // 32] &= ~(1 << (bit % 32))
    s = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), 8, m)
    assert rc == 0, rc
    return s


def p(t):
    return C.c_void_p(t.data_ptr())


for drop in (None, 0, 1, 2):
    if drop is None:
        s = C.c_void_p()
        assert hip.hipStreamCreate(C.byref(s)) == 0
        name = "plain stream"
    else:
        s = masked_stream(drop)
        name = "masked, %d CU(s) per XCD removed" % drop
    _lib.lib.ffgp_set_stream(h, s)
    row = []
    for m in (16384, 12288, 8192, 4096):
        fn = lambda: _lib.lib.ffgp_gemm(h, 0, 0, 1, 0, p(A), k, p(A), k, p(Cm), 16384, m, m, k, -1.0, 1.0)
        fn()
        hip.hipStreamSynchronize(s)
        best = 1e9
        for _ in range(5):
            t0 = time.perf_counter()
            for _ in range(4):
                fn()
            hip.hipStreamSynchronize(s)
            best = min(best, (time.perf_counter() - t0) / 4)
        row.append("m=%d %.3f ms (%.1f TF/s)" % (m, best * 1e3, m * (m + 1) * k / best / 1e12))
    print("%-36s %s" % (name, "  ".join(row)))
_lib.lib.ffgp_set_stream(h, None)
