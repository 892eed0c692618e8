import os, sys, time
sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _devlib  # noqa: F401,E402  (development build: these switches are not in the shipped library)
import torch
from fidelityfusion_amd import _lib, kernel
from fidelityfusion_amd.cigp_v10 import cigp
torch.set_default_dtype(torch.float64)
dev = torch.device("cuda", 0)
for n in (64, 128, 256, 512):
    X = torch.rand(n, 2, device=dev); Y = torch.sin(X.sum(1, keepdim=True))
    m = cigp(kernel.ARDKernel(2), 1.0).to(dev)
    def step():
        for p in m.parameters(): p.grad = None
        (-m.negative_log_likelihood(X, Y)).backward()
    out = []
    for gmax in (0, 1024):
        _lib.check(_lib.lib.ffgp_set_option(_lib.handle(0), b"raw_graph_max_n", float(gmax)), "opt")
        for _ in range(10): step()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(500): step()
        torch.cuda.synchronize(); out.append((time.perf_counter() - t0) / 500 * 1e3)
        g = [p.grad.clone() for p in m.parameters()]
        out.append(g)
    print("n=%d step plain %.3f ms, graph %.3f ms, grad diff %.1e" % (n, out[0], out[2], max(float((a - b).abs().max()) for a, b in zip(out[1], out[3]))))
