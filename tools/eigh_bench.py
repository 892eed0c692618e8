"""eigh of a kernel matrix: the library's two-stage solver (ffgp_syevd) against rocSOLVER (torch.linalg.eigh): time, accuracy and
the time of every stage (events around the stage entry points)."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _devlib  # noqa: F401,E402  (development build: these switches are not in the shipped library)
import torch
from fidelityfusion_amd import eigh as E
from fidelityfusion_amd import _lib as _opt_lib
for _kv in os.environ.get("FFGP_EIGH_OPTS", "").split(","):     # e.g. FFGP_EIGH_OPTS=q2_wave4=0
    if _kv:
        _opt_lib.check(_opt_lib.lib.ffgp_set_option(_opt_lib.handle(0), _kv.split("=")[0].encode(), float(_kv.split("=")[1])), "ffgp_set_option")

dev = "cuda:0"
sizes = [1024, 2048, 4096] + ([8192] if "full" in sys.argv else []) + ([16384] if "big" in sys.argv else [])
for n in sizes:
    D, ls = 8, 1.0
    g = torch.Generator(device=dev).manual_seed(0)
    X = torch.rand((n, D), generator=g, device=dev, dtype=torch.float64)
    d = torch.cdist(X / ls, X / ls)
    K = torch.exp(-0.5 * d * d)
    del d

    def timed(fn, reps=2):
        fn()
        torch.cuda.synchronize()
        best = 1e9
        out = None
        for _ in range(reps):
            t0 = time.perf_counter()
            out = fn()
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        return best * 1e3, out
    from fidelityfusion_amd import _lib
    _lib.set_option("eig_overlap", 1)
    t_ovl, _ = timed(lambda: E.eigh(K))
    _lib.set_option("eig_overlap", 0)
    t_own, (ev, U) = timed(lambda: E.eigh(K))
    t_roc, (ev_r, U_r) = timed(lambda: torch.linalg.eigh(K))
    nrm = float(torch.linalg.matrix_norm(K))
    rec = float(torch.linalg.matrix_norm((U * ev) @ U.T - K)) / nrm
    rec_r = float(torch.linalg.matrix_norm((U_r * ev_r) @ U_r.T - K)) / nrm
    orth = float((U.T @ U - torch.eye(n, device=dev, dtype=torch.float64)).abs().max())
    print("eigh n=%5d D=%d: own %8.1f ms (overlapped driver %8.1f ms)  rocSOLVER %8.1f ms   |K-ULU'|/|K| own %.1e roc %.1e  orth %.1e  max|dlam|/lam_max %.1e"
          % (n, D, t_own, t_ovl, t_roc, rec, rec_r, orth, float((ev - ev_r).abs().max() / ev_r.abs().max())), flush=True)
    if n % 64 == 0:
        t1, (AB, Y) = timed(lambda: E.sy2sb(K), 1)
        t2, (dd, ee, refl) = timed(lambda: E.sb2st(AB), 1)
        t3, (W, Z) = timed(lambda: E.stedc(dd, ee), 1)
        t4, _ = timed(lambda: E.ormq2(refl, Z), 1)
        t5, _ = timed(lambda: E.ormq1(Y, Z), 1)
        print("      stages (incl. buffer copies): sy2sb %.1f  sb2st %.1f  stedc %.1f  ormq2 %.1f  ormq1 %.1f ms" % (t1, t2, t3, t4, t5), flush=True)
        del AB, Y, refl, Z
    del K, U, U_r
    torch.cuda.empty_cache()
