import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time, torch, numpy as np
from fidelityfusion_amd import kernel
from fidelityfusion_amd.cigp_v10 import cigp
torch.set_default_dtype(torch.float64)
dev="cuda:0"
KERNELS = {"ARD": lambda: kernel.ARDKernel(16), "Sum(Linear, Matern52)": lambda: kernel.SumKernel(kernel.LinearKernel(16), kernel.MaternKernel(16))}
for n, kname in ((8192, "ARD"), (16384, "ARD"), (16384, "Sum(Linear, Matern52)")):
    g = torch.Generator(device=dev).manual_seed(0)
    X = torch.rand((n, 16), generator=g, device=dev); Y = torch.randn((n, 1), generator=g, device=dev)
    Xs = torch.rand((256, 16), generator=g, device=dev)
    m = cigp(KERNELS[kname](), 1.0).to(dev)
    with torch.no_grad():
        m(X, Y, Xs); torch.cuda.synchronize()
        m._pcache.clear()
        t0 = time.perf_counter(); m(X, Y, Xs); torch.cuda.synchronize(); t1 = time.perf_counter()
        ts = []
        for _ in range(5):
            t2 = time.perf_counter(); m(X, Y, Xs); torch.cuda.synchronize(); ts.append(time.perf_counter() - t2)
        post = m._pcache.posterior
        Xn = torch.rand((64, 16), generator=g, device=dev); Yn = torch.randn((64, 1), generator=g, device=dev)
        post.append(Xn[:8], Yn[:8]); torch.cuda.synchronize()
        t3 = time.perf_counter(); post.append(Xn[8:], Yn[8:]); torch.cuda.synchronize(); t4 = time.perf_counter()
    print("N=%d %s: first query (factor) %.1f ms, repeated query (nt=256) %.2f ms, append 56 points %.2f ms" % (n, kname, (t1 - t0) * 1e3, min(ts) * 1e3, (t4 - t3) * 1e3))

# acquisition-optimiser step (Bayesian_optimization/acq.py:50-62): UCB of the posterior at 64 moving query points,
# backward to the points -- trainable model (differentiable composition, refactorises) vs frozen model (cached factor)
for n, kname in ((2048, "ARD"), (16384, "ARD"), (16384, "Sum(Linear, Matern52)")):
    g = torch.Generator(device=dev).manual_seed(1)
    X = torch.rand((n, 16), generator=g, device=dev); Y = torch.randn((n, 1), generator=g, device=dev)
    out = []
    for frozen in (False, True):
        m = cigp(KERNELS[kname](), 1.0).to(dev).requires_grad_(not frozen)
        xq = torch.rand((64, 16), generator=g, device=dev).requires_grad_(True)
        opt = torch.optim.Adam([xq], lr=0.05)

        def step():
            opt.zero_grad()
            mean, var = m(X, Y, xq)
            (-(mean.squeeze() + 2.0 * var.diagonal().clamp_min(1e-12).sqrt()).sum()).backward()
            opt.step()
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            step()
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) / 5 * 1e3)
    print("N=%d %s acquisition step (64 query points, UCB, Adam): trainable model %.2f ms, frozen model on the cached factor %.2f ms" % (n, kname, out[0], out[1]))
