"""Blocks of DIFFERENT sizes on one GPU: one call after the other, overlapped through streams
(functional.concurrent_blocks), through cigp_v10.negative_log_likelihood_many (the ragged shared chain when the library has it)
and the largest member alone.  python tools/ragged_probe.py 8192,4096,2048,1024 [d [grad]]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from bench import synthetic_xy
from fidelityfusion_amd import functional as F
from fidelityfusion_amd import kernel
from fidelityfusion_amd.cigp_v10 import cigp, negative_log_likelihood_many

sizes = [int(s) for s in (sys.argv[1] if len(sys.argv) > 1 else "8192,4096,2048,1024").split(",")]
d = int(sys.argv[2]) if len(sys.argv) > 2 else 1
grad = len(sys.argv) > 3 and sys.argv[3] == "grad"
D = 8
dev = torch.device("cuda", 0)
torch.set_default_dtype(torch.float64)
for kv in os.environ.get("FFGP_OPTS", "").split(","):      # library options for A/B runs: FFGP_OPTS=grad_lanes=1
    if kv:
        from fidelityfusion_amd import _lib
        _lib.set_option(kv.split("=")[0], float(kv.split("=")[1]), 0)
        print("option", kv)
models, xs, ys = [], [], []
for f, n in enumerate(sizes):
    X, Y = synthetic_xy(n, D, d, seed=f)
    models.append(cigp(kernel.ARDKernel(D), 1.0).to(dev))
    xs.append(torch.tensor(X, device=dev))
    ys.append(torch.tensor(Y, device=dev))


def timed(fn, reps=7):
    fn()
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best * 1e3, out


def finish(vals):
    if grad:
        torch.stack(list(vals)).sum().backward()
    return torch.stack([v.detach() for v in vals])


def one_by_one(idx):
    return lambda: finish([models[i].negative_log_likelihood(xs[i], ys[i]) for i in idx])


def overlapped(nslots, la):
    def run():
        vals = []
        with F.concurrent_blocks(nslots=nslots, lookahead=la) as cb:
            for i, (m, x, y) in enumerate(zip(models, xs, ys)):
                with cb.slot(i):
                    w, amp, clamp = m.kernel.effective()
                    vals.append(-F.nlml(x, y, w, amp, diag_add=m.log_beta.exp().pow(-1) + 1e-6, clamp=clamp, pi_const=3.1415, **F._slot_args()))
        return finish(vals)
    return run


def many():
    return finish(list(negative_log_likelihood_many(models, xs, ys)))


ctx = torch.enable_grad() if grad else torch.no_grad()
with ctx:
    big = max(range(len(sizes)), key=lambda i: sizes[i])
    t_big, _ = timed(one_by_one([big]))
    print("sizes %s d=%d %s" % (sizes, d, "fwd+grad" if grad else "forward"))
    print("  largest member alone (N=%d)        %8.3f ms" % (sizes[big], t_big), flush=True)
    t_seq, ref = timed(one_by_one(range(len(sizes))))
    print("  one after the other                 %8.3f ms  (%.2fx largest)" % (t_seq, t_seq / t_big), flush=True)
    for la in (False, True):
        t_ov, out = timed(overlapped(len(sizes), la))
        print("  overlapped, %d slots, lookahead=%d    %8.3f ms  (%.2fx largest)  equal=%s" % (len(sizes), la, t_ov, t_ov / t_big,
                                                                                            bool(torch.equal(out, ref))), flush=True)
    t_m, out = timed(many)
    print("  negative_log_likelihood_many        %8.3f ms  (%.2fx largest)  equal=%s" % (t_m, t_m / t_big, bool(torch.equal(out, ref))),
          flush=True)
