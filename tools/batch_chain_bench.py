"""F equal-shape blocks, three ways: one call after the other, overlapped through streams
(functional.concurrent_blocks), and as ONE factorisation chain (cigp_v10.negative_log_likelihood_many -> ffgp_nlml_fused_batch);
forward only and with gradients.  python tools/batch_chain_bench.py [N [F]]   (default 4096 8: eight C2 blocks)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from bench import synthetic_xy
from fidelityfusion_amd import functional as F
from fidelityfusion_amd import kernel
from fidelityfusion_amd.cigp_v10 import cigp, negative_log_likelihood_many

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
nF = int(sys.argv[2]) if len(sys.argv) > 2 else 8
for kv in sys.argv[3:]:                      # further arguments: key=value library options (e.g. batch_refine=0)
    from fidelityfusion_amd import _lib
    k, v = kv.split("=")
    _lib.set_option(k, float(v), 0)
    print("option", k, "=", v)
D, d = 8, 1
dev = torch.device("cuda", 0)
torch.set_default_dtype(torch.float64)
models, xs, ys = [], [], []
for f in range(nF):
    X, Y = synthetic_xy(n, D, d, seed=f)
    models.append(cigp(kernel.ARDKernel(D), 1.0).to(dev))
    xs.append(torch.tensor(X, device=dev))
    ys.append(torch.tensor(Y, device=dev))


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best * 1e3, out


def one_by_one(grad):
    def run():
        vals = [m.negative_log_likelihood(x, y) for m, x, y in zip(models, xs, ys)]
        if grad:
            torch.stack(vals).sum().backward()
        return torch.stack([v.detach() for v in vals])
    return run


def overlapped(grad):
    def run():
        vals = []
        with F.concurrent_blocks(nslots=2) as cb:
            for i, (m, x, y) in enumerate(zip(models, xs, ys)):
                with cb.slot(i):
                    w, amp, clamp = m.kernel.effective()
                    vals.append(-F.nlml(x, y, w, amp, diag_add=m.log_beta.exp().pow(-1) + 1e-6, clamp=clamp, pi_const=3.1415, **F._slot_args()))
        if grad:
            torch.stack(vals).sum().backward()
        return torch.stack([v.detach() for v in vals])
    return run


def chained(grad):
    def run():
        vals = negative_log_likelihood_many(models, xs, ys)
        if grad:
            vals.sum().backward()
        return vals.detach()
    return run


flops = nF * (n ** 3 / 3.0 + n * n * d + 2.0 * n * n * D)
for grad in (False, True):
    ctx = torch.enable_grad() if grad else torch.no_grad()
    with ctx:
        res = {}
        for name, mk in (("one after the other", one_by_one), ("overlapped (streams)", overlapped), ("one chain (batch)", chained)):
            ms, out = timed(mk(grad))
            res[name] = (ms, out)
            print("%d x N=%d %-8s %-22s %8.3f ms   %6.1f TFLOP/s" % (nF, n, "fwd+grad" if grad else "forward", name, ms,
                                                                    flops * (3.0 if grad else 1.0) / ms / 1e9), flush=True)
        a, b = res["one after the other"][1], res["one chain (batch)"][1]
        print("   values of the chained call == the individual calls, bit for bit:", bool(torch.equal(a, b)))
