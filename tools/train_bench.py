"""K Adam steps per call (cigp_v10.train_many -> ffgp_train_raw) against the per-step loop through the drop-in modules
(gate: 200 steps at N = 128 in <= 20 ms).  python tools/train_bench.py [steps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from bench import synthetic_xy
from fidelityfusion_amd import kernel
from fidelityfusion_amd.cigp_v10 import cigp, train_many

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
for kv in os.environ.get("FFGP_OPTS", "").split(","):      # library options for A/B runs: FFGP_OPTS=small_finish=1
    if kv:
        from fidelityfusion_amd import _lib
        _lib.set_option(kv.split("=")[0], float(kv.split("=")[1]), 0)
        print("option", kv)
dev = torch.device("cuda", 0)
torch.set_default_dtype(torch.float64)


def make(n, D, d, F=1):
    ms, xs, ys = [], [], []
    for f in range(F):
        X, Y = synthetic_xy(n, D, d, seed=f)
        ms.append(cigp(kernel.ARDKernel(D), 1.0).to(dev))
        xs.append(torch.tensor(X, device=dev))
        ys.append(torch.tensor(Y, device=dev))
    return ms, xs, ys


def loop(ms, xs, ys, k):
    for m, x, y in zip(ms, xs, ys):
        opt = torch.optim.Adam(m.parameters(), lr=1e-2)
        for _ in range(k):
            opt.zero_grad()
            loss = -m.negative_log_likelihood(x, y)
            loss.backward()
            opt.step()


for n, D, d, F in ((32, 5, 1, 1), (128, 5, 1, 1), (256, 5, 1, 1), (512, 5, 1, 1), (1024, 8, 1, 1), (128, 5, 1, 8), (64, 5, 1, 16), (300, 5, 1, 3), (512, 5, 1, 4), (1024, 8, 1, 4)):
    ms, xs, ys = make(n, D, d, F)
    train_many(ms, xs, ys, 5)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    train_many(ms, xs, ys, steps)
    torch.cuda.synchronize()
    t_many = time.perf_counter() - t0
    ms2, _, _ = make(n, D, d, F)
    k_loop = max(10, steps // 4)
    loop(ms2, xs, ys, 3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    loop(ms2, xs, ys, k_loop)
    torch.cuda.synchronize()
    t_loop = (time.perf_counter() - t0) * steps / k_loop
    print("N=%5d D=%d d=%d F=%2d  %d steps: train_many %8.2f ms (%.3f ms/step/model)   per-step loop %8.2f ms (%.3f)   x%.1f"
          % (n, D, d, F, steps, t_many * 1e3, t_many * 1e3 / steps / F, t_loop * 1e3, t_loop * 1e3 / steps / F, t_loop / t_many), flush=True)
