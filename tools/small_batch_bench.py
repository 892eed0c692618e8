"""F independent small cigp models: one training step each, individually (F library calls) against negative_log_likelihood_many
(one call): python tools/small_batch_bench.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fidelityfusion_amd import kernel
from fidelityfusion_amd.cigp_v10 import cigp, negative_log_likelihood_many
torch.set_default_dtype(torch.float64)
dev = torch.device("cuda", 0)
for nF, n in ((8, 32), (8, 64), (8, 128), (16, 64), (4, 128)):
    ms = [cigp(kernel.ARDKernel(2), 1.0).to(dev) for _ in range(nF)]
    xs = [torch.rand(n, 2, device=dev) for _ in range(nF)]
    ys = [torch.sin(x.sum(1, keepdim=True)) for x in xs]
    params = [p for m in ms for p in m.parameters()]

    def loop():
        for p in params:
            p.grad = None
        for m, x, y in zip(ms, xs, ys):
            (-m.negative_log_likelihood(x, y)).backward()

    def many():
        for p in params:
            p.grad = None
        (-negative_log_likelihood_many(ms, xs, ys).sum()).backward()
    out = []
    for fn in (loop, many):
        for _ in range(10):
            fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(200):
            fn()
        torch.cuda.synchronize(); out.append((time.perf_counter() - t0) / 200 * 1e3)
    print("F=%2d models of N=%3d: one step of each, one after the other %.3f ms, batched %.3f ms (%.3f ms per model)" % (nF, n, out[0], out[1], out[1] / nF))
