"""eigh of a kernel matrix: the library's own block Jacobi (eigh.jacobi_eigh) against rocSOLVER (torch.linalg.eigh), time and accuracy."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fidelityfusion_amd.eigh import jacobi_eigh

dev = "cuda:0"
for (n, D, ls) in [(1024, 8, 1.0), (2048, 8, 1.0), (4096, 8, 1.0)] + ([(8192, 8, 1.0)] if "full" in sys.argv else []):
    g = torch.Generator(device=dev).manual_seed(0)
    X = torch.rand((n, D), generator=g, device=dev, dtype=torch.float64)
    d = torch.cdist(X / ls, X / ls)
    K = torch.exp(-0.5 * d * d)
    del d

    def timed(fn, reps=2):
        fn()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(reps):
            t0 = time.perf_counter()
            out = fn()
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        return best * 1e3, out
    t_own, (ev, U) = timed(lambda: jacobi_eigh(K), reps=1)
    t_roc, (ev_r, U_r) = timed(lambda: torch.linalg.eigh(K))
    nrm = float(torch.linalg.matrix_norm(K))
    rec = float(torch.linalg.matrix_norm((U * ev) @ U.T - K)) / nrm
    rec_r = float(torch.linalg.matrix_norm((U_r * ev_r) @ U_r.T - K)) / nrm
    orth = float((U.T @ U - torch.eye(n, device=dev, dtype=torch.float64)).abs().max())
    print("eigh n=%5d D=%d ls=%.1f: own %8.1f ms  rocSOLVER %8.1f ms   |K-ULU'|/|K| own %.1e roc %.1e  orth %.1e  max|dlam|/lam_max %.1e"
          % (n, D, ls, t_own, t_roc, rec, rec_r, orth, float((ev - ev_r).abs().max() / ev_r.abs().max())), flush=True)
    del K, U, U_r
    torch.cuda.empty_cache()
