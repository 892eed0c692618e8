"""GP_basic.log_likelihood (the V2 form: potrf + potrs) forward / forward+backward, super-block sweeps on and off."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fidelityfusion_amd import _lib, kernel
from fidelityfusion_amd.gp_basic import GP_basic
torch.set_default_dtype(torch.float64)
dev = "cuda:0"
for n, D in ((4096, 8), (16384, 16)):
    g = torch.Generator(device=dev).manual_seed(0)
    X = torch.rand((n, D), generator=g, device=dev); Y = torch.randn((n, 1), generator=g, device=dev)
    m = GP_basic(kernel.ARDKernel(D), 0.6).to(dev)
    for sb in (0, 1024):
        _lib.set_option("super_block", sb, 0)
        out = []
        for grad in (False, True):
            def step():
                if grad:
                    for p in m.parameters():
                        p.grad = None
                    (-m.log_likelihood(X, Y)).sum().backward()
                else:
                    with torch.no_grad():
                        m.log_likelihood(X, Y)
            for _ in range(2):
                step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                step()
            torch.cuda.synchronize()
            out.append((time.perf_counter() - t0) / 5 * 1e3)
        print("N=%d GP_basic.log_likelihood, super_block=%d: forward %.2f ms, forward+backward %.2f ms" % (n, sb, out[0], out[1]))
_lib.set_option("super_block", 1024, 0)
