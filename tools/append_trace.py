"""Posterior.append of 56 points at N=16384 (for a rocprofv3 --kernel-trace timeline)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fidelityfusion_amd import kernel
from fidelityfusion_amd.cigp_v10 import cigp
torch.set_default_dtype(torch.float64)
dev = "cuda:0"
n = 16384
g = torch.Generator(device=dev).manual_seed(0)
X = torch.rand((n, 16), generator=g, device=dev); Y = torch.randn((n, 1), generator=g, device=dev)
Xs = torch.rand((256, 16), generator=g, device=dev)
m = cigp(kernel.ARDKernel(16), 1.0).to(dev)
with torch.no_grad():
    m(X, Y, Xs)
    post = m._post
    Xn = torch.rand((200, 16), generator=g, device=dev); Yn = torch.randn((200, 1), generator=g, device=dev)
    post.append(Xn[:8], Yn[:8]); post.predict(Xs); torch.cuda.synchronize()
    for i in range(3):
        t0 = time.perf_counter(); post.append(Xn[8 + 56 * i:64 + 56 * i], Yn[8 + 56 * i:64 + 56 * i]); torch.cuda.synchronize(); t1 = time.perf_counter()
        post.predict(Xs); torch.cuda.synchronize(); t2 = time.perf_counter()
        print("append %.2f ms, next query %.2f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3))
