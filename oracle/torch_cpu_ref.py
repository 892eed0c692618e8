"""torch-CPU restatement of the reference's cigp likelihood -- TEST / BASELINE INFRASTRUCTURE, NOT PRODUCT CODE.

This is the `cpu_ref` of SURVEY.md section 8(d) / BASELINE.md section 3: the same torch-CPU operator sequence
the reference executes for one `cigp.negative_log_likelihood` with an `ARDKernel` (and, through autograd, for
`loss.backward()`), written out so that it can be timed on the GPU box's host cores next to the HIP path.  The
reference's own Python never travels to the GPU box; this file is pinned against fixtures captured from the
imported reference (tests/test_oracle_golden.py::test_torch_cpu_ref_*).  Only tests/ and bench.py's `cpu_baseline`
leg import it; nothing under fidelityfusion_amd/ does.

Operator sequence (file:line relative to the reference root):
    GaussianProcess/kernel.py:98-105   ell = |length_scales| + 1e-9;  K = |signal_variance| * exp(-1/2 cdist(x/ell, x/ell)^2)
    GaussianProcess/cigp_v10.py:57-58  Sigma = K + exp(log_beta)^-1 * I + 1e-6 * I
    GaussianProcess/cigp_v10.py:61     L = torch.linalg.cholesky(Sigma)
    GaussianProcess/cigp_v10.py:63     Gamma = triangular_solve(Y, L, upper=False)   (solve_triangular: same LAPACK dtrsm)
    GaussianProcess/cigp_v10.py:67-69  nll = 1/2 sum(Gamma^2) + d sum(log diag L) + 1/2 N d log(2 * 3.1415); returns -nll
"""
import time

import torch

EPS = 1e-9          # GaussianProcess/kernel.py:21
JITTER = 1e-6       # GaussianProcess/cigp_v10.py:13
PI_TRUNC = 3.1415   # GaussianProcess/cigp_v10.py:15


def cigp_ll(X, Y, length_scales, signal_variance, log_beta, stages=None, keep=None):
    """+LL (what cigp.negative_log_likelihood returns, cigp_v10.py:69) as a differentiable torch-CPU scalar.
    stages: optional dict that receives per-stage wall seconds (assemble / potrf / trsm).
    keep: optional dict that receives the factor `L` (detached) -- `cigp_forward` re-derives the very same Sigma and factor
    (cigp_v10.py:31-35 against :57-61, no y_var), so a posterior check can reuse it instead of a second dpotrf."""
    t0 = time.perf_counter()
    ell = torch.abs(length_scales) + EPS                                     # kernel.py:98
    sq = torch.cdist(X / ell, X / ell, p=2) ** 2                             # kernel.py:100-104
    K = torch.abs(signal_variance) * torch.exp(-0.5 * sq)                    # kernel.py:105
    n, d = Y.shape
    eye = torch.eye(n, dtype=X.dtype)
    Sigma = K + log_beta.exp().pow(-1) * eye + JITTER * eye                  # cigp_v10.py:57-58
    t1 = time.perf_counter()
    L = torch.linalg.cholesky(Sigma)                                         # cigp_v10.py:61
    t2 = time.perf_counter()
    Gamma = torch.linalg.solve_triangular(L, Y, upper=False)                 # cigp_v10.py:63
    t3 = time.perf_counter()
    nll = 0.5 * (Gamma ** 2).sum() + L.diag().log().sum() * d \
        + 0.5 * n * torch.log(2 * torch.tensor(PI_TRUNC, dtype=X.dtype)) * d  # cigp_v10.py:67-68
    if stages is not None:
        stages.update(assemble=t1 - t0, potrf=t2 - t1, trsm=t3 - t2)
    if keep is not None:
        keep["L"] = L.detach()
    return -nll                                                              # cigp_v10.py:69


def ard_kernel(x1, x2, length_scales, signal_variance):
    """GaussianProcess/kernel.py:98-105"""
    ell = torch.abs(length_scales) + EPS
    return torch.abs(signal_variance) * torch.exp(-0.5 * torch.cdist(x1 / ell, x2 / ell, p=2) ** 2)


def cigp_forward(X, Y, Xs, length_scales, signal_variance, log_beta, L=None):
    """(mean, var) of cigp.forward, GaussianProcess/cigp_v10.py:24-48: the noise scalar is added to EVERY entry of the full
    covariance (:44) and y_var is ignored.  L: the factor of Sigma when the caller already holds it (same Sigma as the
    likelihood's when there is no y_var)."""
    if L is None:
        n = X.shape[0]
        eye = torch.eye(n, dtype=X.dtype)
        Sigma = ard_kernel(X, X, length_scales, signal_variance) + log_beta.exp().pow(-1) * eye + JITTER * eye   # :31-32
        L = torch.linalg.cholesky(Sigma)                                                                         # :35
    kx = ard_kernel(X, Xs, length_scales, signal_variance)                                                       # :34
    LinvKx = torch.linalg.solve_triangular(L, kx, upper=False)                                                   # :36
    mean = kx.t() @ torch.cholesky_solve(Y, L)                                                                   # :39
    var = ard_kernel(Xs, Xs, length_scales, signal_variance) - LinvKx.t() @ LinvKx                               # :41
    return mean, var + log_beta.exp().pow(-1)                                                                    # :44


def cigp_ll_and_grads(X, Y, length_scales, signal_variance, log_beta):
    """(+LL, {name: d(+LL)/d(name)}) through torch autograd, as `loss.backward()` at FidelityFusion_Models/ResGP.py:84-87."""
    ls = length_scales.clone().requires_grad_(True)
    sv = signal_variance.clone().requires_grad_(True)
    lb = log_beta.clone().requires_grad_(True)
    Yr = Y.clone().requires_grad_(True)
    ll = cigp_ll(X, Yr, ls, sv, lb)
    ll.backward()
    return ll.detach(), {"length_scales": ls.grad, "signal_variance": sv.grad, "log_beta": lb.grad, "Y": Yr.grad}


def time_cigp(X, Y, length_scales, signal_variance, log_beta, repeats=3, with_backward=True, budget_s=None, keep=None):
    """1 warm-up + min of `repeats` of the forward, and (optionally) of forward + backward.  budget_s bounds the
    total wall time: measurements that would not fit are skipped (None).  Returns a dict of seconds and the LL.
    keep: optional dict that receives the factor `L` of the warm-up run and, when a backward ran, `grads` (the autograd
    gradients of +LL: length_scales, signal_variance, log_beta, Y) -- the parity columns of bench.py."""
    out = {"fwd_s": None, "fwd_bwd_s": None, "stages_s": None, "ll": None}
    t_start = time.perf_counter()

    def left():
        return None if budget_s is None else budget_s - (time.perf_counter() - t_start)

    with torch.no_grad():
        t0 = time.perf_counter()
        ll = cigp_ll(X, Y, length_scales, signal_variance, log_beta, keep=keep)   # warm-up (also the value)
        first = time.perf_counter() - t0
        out["ll"] = float(ll)
        best, st_best = first, None
        for _ in range(repeats):
            if left() is not None and left() < 1.2 * first:
                break
            st = {}
            t0 = time.perf_counter()
            cigp_ll(X, Y, length_scales, signal_variance, log_beta, stages=st)
            dt = time.perf_counter() - t0
            if dt <= best or st_best is None:
                best, st_best = min(best, dt), st
        out["fwd_s"], out["stages_s"] = best, st_best
    if with_backward:
        runs = []                                                             # first run = warm-up when there is time for more
        for _ in range(repeats + 1):
            need = 5.0 * best if not runs else 1.2 * runs[-1]                 # autograd ~ 4-5x the forward
            if left() is not None and left() < need:
                break
            t0 = time.perf_counter()
            _, gr = cigp_ll_and_grads(X, Y, length_scales, signal_variance, log_beta)
            runs.append(time.perf_counter() - t0)
            if keep is not None:
                keep["grads"] = gr
        out["fwd_bwd_s"] = min(runs[1:]) if len(runs) > 1 else (runs[0] if runs else None)
    return out


def host_description():
    """CPU model, logical cores, torch threads and BLAS vendor of this host."""
    import os
    import re
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    cfg = torch.__config__.show()
    blas = "MKL" if re.search(r"USE_MKL=(ON|1)", cfg) or "BLAS_INFO=mkl" in cfg else \
        ("OpenBLAS" if "open" in cfg.lower() and "blas" in cfg.lower() else "unknown")
    return {"cpu_model": model, "logical_cores": os.cpu_count(), "torch_threads": torch.get_num_threads(), "blas": blas,
            "torch": torch.__version__}
