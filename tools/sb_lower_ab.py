"""Band reduction on the lower triangle only (option `sb_lower`, sy2sb_av_sym): A/B of eigh and of the sy2sb stage on the shipped
library, with residual / orthogonality of both.  python tools/sb_lower_ab.py [N ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from fidelityfusion_amd import _lib
from fidelityfusion_amd import eigh as E

dev = "cuda:0"
sizes = [int(a) for a in sys.argv[1:]] or [1024, 2048, 4096, 8192]
for _kv in os.environ.get("FFGP_OPTS", "").split(","):      # e.g. FFGP_OPTS=sb_sym_wg=4096
    if _kv:
        _lib.set_option(_kv.split("=")[0], float(_kv.split("=")[1]), 0)
for n in sizes:
    g = torch.Generator(device=dev).manual_seed(0)
    X = torch.rand((n, 8), generator=g, device=dev, dtype=torch.float64)
    d = torch.cdist(X, X)
    K = torch.exp(-0.5 * d * d)
    del d
    nrm = float(torch.linalg.matrix_norm(K))

    def timed(fn, reps=3):
        fn()
        torch.cuda.synchronize()
        best, out = 1e9, None
        for _ in range(reps):
            t0 = time.perf_counter()
            out = fn()
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        return best * 1e3, out

    row = []
    evs = []
    for lower in (0, 1, 0, 1):
        _lib.set_option("sb_lower", float(lower), 0)
        t, (ev, U) = timed(lambda: E.eigh(K))
        ts, _ = timed(lambda: E.sy2sb(K), 2) if n % 64 == 0 else (float("nan"), None)
        rec = float(torch.linalg.matrix_norm((U * ev) @ U.T - K)) / nrm
        orth = float((U.T @ U - torch.eye(n, device=dev, dtype=torch.float64)).abs().max())
        evs.append(ev)
        row.append("sb_lower=%d eigh %.1f ms sy2sb %.1f ms res %.1e orth %.1e" % (lower, t, ts, rec, orth))
        del U
    print("N = %5d | %s | max |dlam| / lam_max %.1e" % (n, " | ".join(row), float((evs[0] - evs[1]).abs().max() / evs[0].abs().max())), flush=True)
_lib.set_option("sb_lower", 1.0, 0)
