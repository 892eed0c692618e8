"""Composed kernels: the two-descriptor fused path against the part-by-part composition, same model, same box.
python tools/pair_bench.py [N] [D]   -- cigp over SumKernel(LinearKernel, MaternKernel) (the reference demos' kernel,
cigp_v10.py:81) and ProductKernel(ARDKernel, RationalQuadraticKernel): NLL forward and forward + backward."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from fidelityfusion_amd import kernel
from fidelityfusion_amd.cigp_v10 import cigp

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
D = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev = torch.device("cuda", 0)
rng = np.random.default_rng(0)
X = torch.tensor(rng.uniform(0, 1, (n, D)), device=dev)
Y = torch.tensor(np.sin(X.cpu().numpy().sum(1, keepdims=True)) + 0.05 * rng.standard_normal((n, 1)), device=dev)


def timed(fn, reps=5):
    fn(); fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


S, P = kernel.SumKernel, kernel.ProductKernel
for name, k in (("Sum(Linear, Matern52)", S(kernel.LinearKernel(D), kernel.MaternKernel(D))),
                ("Product(ARD, RQ)", P(kernel.ARDKernel(D), kernel.RationalQuadraticKernel())),
                ("Sum(Prod(ARD,RQ), Linear)", S(P(kernel.ARDKernel(D), kernel.RationalQuadraticKernel()), kernel.LinearKernel(D))),
                ("Sum(Prod(ARD,M32), Prod(Lin,SE))", S(P(kernel.ARDKernel(D), kernel.MaternKernel(D, nu=1.5)),
                                                       P(kernel.LinearKernel(D), kernel.SquaredExponentialKernel())))):
    m = cigp(k, 2.0).to(dev)

    def fwd():
        with torch.no_grad():
            return m.negative_log_likelihood(X, Y)

    def fwdbwd():
        for p in m.parameters():
            p.grad = None
        m.negative_log_likelihood(X, Y).backward()

    out = {}
    for fuse in (True, False):
        kernel.FUSE_PAIRS = fuse
        out[fuse] = (timed(fwd), timed(fwdbwd), float(fwd()))
    kernel.FUSE_PAIRS = True
    print("%-32s N=%d D=%d  forward %.2f ms fused / %.2f composed   fwd+bwd %.2f / %.2f   (values %.12g / %.12g)"
          % (name, n, D, out[True][0], out[False][0], out[True][1], out[False][1], out[True][2], out[False][2]))

# input gradients of the demo kernel (what an acquisition optimiser differentiates): K(x_train, x_query) w.r.t. 256 query points
k = S(kernel.LinearKernel(D), kernel.MaternKernel(D)).to(dev)
for p in k.parameters():
    p.requires_grad_(False)
Xq = torch.tensor(rng.uniform(0, 1, (256, D)), device=dev)
R = torch.tensor(rng.standard_normal((n, 256)), device=dev)


def xgrad():
    xq = Xq.clone().requires_grad_(True)
    (k(X, xq) * R).sum().backward()
    return xq.grad


res = {}
for fuse in (True, False):
    kernel.FUSE_PAIRS = fuse
    res[fuse] = (timed(xgrad, 20), xgrad())
kernel.FUSE_PAIRS = True
print("d K(X, Xq) / d Xq, Sum(Linear, Matern52), %d x 256: %.3f ms fused / %.3f composed (max diff %.1e)"
      % (n, res[True][0], res[False][0], float((res[True][1] - res[False][1]).abs().max())))
