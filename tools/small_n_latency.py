"""Per-call latency of the drop-in cigp at the small sizes the reference's own demos use (forward+backward, predict)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from fidelityfusion_amd import kernel
from fidelityfusion_amd.cigp_v10 import cigp

torch.set_default_dtype(torch.float64)
for dev in ("cuda", "cpu"):
    for n in (32, 128, 256, 512, 1024, 2048):
        X = torch.rand(n, 2, device=dev)
        Y = torch.sin(X.sum(1, keepdim=True)) + 0.05 * torch.rand(n, 1, device=dev)
        Xs = torch.rand(100, 2, device=dev)
        m = cigp(kernel.ARDKernel(2), 1.0).to(dev)

        def step():
            for p in m.parameters():
                p.grad = None
            (-m.negative_log_likelihood(X, Y)).backward()

        def pred():
            with torch.no_grad():
                m._pcache.clear()
                m(X, Y, Xs)

        out = []
        for fn in (step, pred):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                fn()
            torch.cuda.synchronize()
            out.append((time.perf_counter() - t0) / 20 * 1e3)
        print("tensors on %-4s n=%4d: NLL fwd+bwd %.3f ms, predict(100 pts, refactor) %.3f ms" % (dev, n, out[0], out[1]))
