import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from fidelityfusion_amd import _lib, kernel
from fidelityfusion_amd.cigp_v10 import cigp
torch.set_default_dtype(torch.float64)
dev = torch.device("cuda", 0)
for n in (48, 64, 80, 96, 112, 128):
    X = torch.rand(n, 2, device=dev); Y = torch.sin(X.sum(1, keepdim=True))
    m = cigp(kernel.ARDKernel(2), 1.0).to(dev)
    def step():
        for p in m.parameters(): p.grad = None
        (-m.negative_log_likelihood(X, Y)).backward()
    out = []
    for fin in (0, 1):
        _lib.check(_lib.lib.ffgp_set_option(_lib.handle(0), b"small_finish", float(fin)), "opt")
        for _ in range(10): step()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(500): step()
        torch.cuda.synchronize(); out.append((time.perf_counter() - t0) / 500 * 1e3)
    print("n=%d step blocked %.3f ms, finishing kernel %.3f ms" % (n, out[0], out[1]))
