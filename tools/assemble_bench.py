"""Covariance assembly: time, TB/s of K written (lower-only: 4 N^2 bytes) and the largest relative difference from a torch
evaluation of the same formula."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fidelityfusion_amd import _lib

dev = torch.device("cuda:0")
h = _lib.handle(0)
_lib.bind_stream(h, 0)
p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
for kv in os.environ.get("FFGP_ASM_OPTS", "").split(","):     # e.g. FFGP_ASM_OPTS=asm_mm=0  or  asm_mm_grid=512
    if kv:
        _lib.check(_lib.lib.ffgp_set_option(h, kv.split("=")[0].encode(), float(kv.split("=")[1])), "ffgp_set_option")
for nbytes in (1 << 31,):
    t = torch.empty(nbytes // 8, device=dev, dtype=torch.float64)
    ts = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); t.fill_(1.5); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1))
    print("torch fill_ of %d MB: %.3f ms = %.2f TB/s written" % (nbytes >> 20, min(ts), nbytes / min(ts) / 1e9), flush=True)
    del t
for (n, D, lower, pad) in [(16384, 16, 1, 0), (16384, 16, 0, 0), (8192, 8, 1, 0)] + [(int(v), 8, 1, 0) for v in os.environ.get('FFGP_ASM_SIZES', '').split(',') if v]:
    g = torch.Generator(device=dev).manual_seed(0)
    X = torch.rand((n, D), generator=g, device=dev, dtype=torch.float64)
    w = torch.rand((D,), generator=g, device=dev, dtype=torch.float64) + 0.5
    amp = torch.tensor([1.3], device=dev, dtype=torch.float64)
    dadd = torch.tensor([0.37], device=dev, dtype=torch.float64)
    outs = []
    line = "assemble n=%5d ld=n+%2d D=%2d %s: " % (n, pad, D, "lower" if lower else "full ")
    for mode in (0,):
        K = torch.zeros((n, n + pad), device=dev, dtype=torch.float64)
        fn = lambda: _lib.lib.ffgp_assemble(h, p(X), n, p(X), n, D, p(w), p(amp), 1e-30, p(dadd), None, 0, None, 0, 0.0, 0.0, p(K), n + pad, lower, 0, 1.0)
        fn()
        ts = []
        for _ in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record(); e1.synchronize()
            ts.append(e0.elapsed_time(e1))
        byt = (4.0 if lower else 8.0) * n * n
        line += "%.3f ms %.2f TB/s   " % (min(ts), byt / min(ts) / 1e9)
        outs.append(torch.tril(K[:, :n]) if lower else K[:, :n])
    d2 = torch.cdist(X * w, X * w) ** 2
    ref = torch.tril(amp * torch.exp(-0.5 * d2)) if lower else amp * torch.exp(-0.5 * d2)
    ref.diagonal().copy_(amp + dadd)
    line += "max rel |K - torch| %.2e" % float(((outs[0] - ref).abs() / ref.abs().clamp_min(1e-300)).max())
    print(line, flush=True)
