"""One HOGP block (GAR's per-fidelity model) at config-5 size on the device: log_likelihood forward / +backward, forward."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from fidelityfusion_amd import kernel
from fidelityfusion_amd.hogp_simple import HOGP_simple

torch.set_default_dtype(torch.float64)
dev = "cuda:0"
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
d1 = d2 = int(sys.argv[2]) if len(sys.argv) > 2 else 64
g = torch.Generator(device=dev).manual_seed(0)
X = torch.rand((n, 8), generator=g, device=dev)
Y = torch.randn((n, d1, d2), generator=g, device=dev)
Xt = torch.rand((64, 8), generator=g, device=dev)
m = HOGP_simple(kernel.ARDKernel(8), 1.0, [d1, d2]).double().to(dev)


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best * 1e3


def fwd():
    with torch.no_grad():
        m.log_likelihood(X, Y)


def fwdbwd():
    for p in m.parameters():
        p.grad = None
    m.log_likelihood(X, Y).backward()


def eig():
    from fidelityfusion_amd import eigh as E
    E.eigh(m.K[0].detach())


def eig_roc():
    torch.linalg.eigh(m.K[0].detach(), UPLO="U")


def pred():
    with torch.no_grad():
        m.forward(X, Xt)


t_f, t_fb, t_e, t_r, t_p = timed(fwd), timed(fwdbwd), timed(eig), timed(eig_roc), timed(pred)
print("HOGP block N=%d d=%dx%d: log_likelihood fwd %.1f ms (of which eigh(K_x) %.1f ms on ffgp_syevd; rocSOLVER would take %.1f ms), fwd+bwd %.1f ms, "
      "forward(64 pts) %.1f ms" % (n, d1, d2, t_f, t_e, t_r, t_fb, t_p))
