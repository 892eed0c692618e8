import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time, torch, numpy as np
from fidelityfusion_amd import kernel
from fidelityfusion_amd.cigp_v10 import cigp
torch.set_default_dtype(torch.float64)
dev="cuda:0"
for n in (8192, 16384):
    g = torch.Generator(device=dev).manual_seed(0)
    X = torch.rand((n, 16), generator=g, device=dev); Y = torch.randn((n, 1), generator=g, device=dev)
    Xs = torch.rand((256, 16), generator=g, device=dev)
    m = cigp(kernel.ARDKernel(16), 1.0).to(dev)
    with torch.no_grad():
        m(X, Y, Xs); torch.cuda.synchronize()
        m._post = None
        t0 = time.perf_counter(); m(X, Y, Xs); torch.cuda.synchronize(); t1 = time.perf_counter()
        ts = []
        for _ in range(5):
            t2 = time.perf_counter(); m(X, Y, Xs); torch.cuda.synchronize(); ts.append(time.perf_counter() - t2)
        post = m._post[2]
        Xn = torch.rand((64, 16), generator=g, device=dev); Yn = torch.randn((64, 1), generator=g, device=dev)
        post.append(Xn[:8], Yn[:8]); torch.cuda.synchronize()
        t3 = time.perf_counter(); post.append(Xn[8:], Yn[8:]); torch.cuda.synchronize(); t4 = time.perf_counter()
    print("N=%d: first query (factor) %.1f ms, repeated query (nt=256) %.2f ms, append 56 points %.2f ms" % (n, (t1 - t0) * 1e3, min(ts) * 1e3, (t4 - t3) * 1e3))
