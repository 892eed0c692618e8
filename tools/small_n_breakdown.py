"""Where one small-N training step goes: raw C call (sync / async), the autograd wrapper, the module's parameter plumbing."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from fidelityfusion_amd import _lib, kernel
from fidelityfusion_amd import functional as F
from fidelityfusion_amd.cigp_v10 import cigp

torch.set_default_dtype(torch.float64)
dev = torch.device("cuda", 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
X = torch.rand(n, 2, device=dev)
Y = torch.sin(X.sum(1, keepdim=True)) + 0.05 * torch.rand(n, 1, device=dev)
m = cigp(kernel.ARDKernel(2), 1.0).to(dev)


def timed(fn):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def step():
    for p in m.parameters():
        p.grad = None
    (-m.negative_log_likelihood(X, Y)).backward()


w = torch.ones(2, device=dev, requires_grad=True)
amp = torch.ones(1, device=dev, requires_grad=True)
dadd = torch.tensor([0.37], device=dev, requires_grad=True)


def wrapper_only():
    F.nlml(X, Y, w, amp, diag_add=dadd, clamp=1e-30).backward()


def effective_only():
    with torch.no_grad():
        m.kernel.effective()
        m.log_beta.exp().pow(-1) + 1e-6


h = _lib.handle(0)
_lib.bind_stream(h, 0)
keep = []
p, _ = F._problem(dev, X, Y, w.detach(), amp.detach(), dadd.detach(), None, None, 0.0, 0.0, 1e-30, 1, 3.1415, keep, (0, 1.0))
out = torch.empty((), device=dev)
g = _lib.Grads()
gw, ga, gd = torch.empty(2, device=dev), torch.empty(1, device=dev), torch.empty(1, device=dev)
g.g_w_dev, g.g_amp_dev, g.g_diag_add_dev = gw.data_ptr(), ga.data_ptr(), gd.data_ptr()


def raw_sync():
    _lib.lib.ffgp_nlml_fused(h, C.byref(p), out.data_ptr(), C.byref(g))


def raw_async():
    _lib.lib.ffgp_nlml_fused_async(h, C.byref(p), out.data_ptr(), C.byref(g))


def raw_fwd_sync():
    _lib.lib.ffgp_nlml_fused(h, C.byref(p), out.data_ptr(), None)


print("n=%d  module step %.3f ms | F.nlml+backward %.3f | parameter plumbing %.3f | C call fwd+grad sync %.3f, async (pipelined) %.3f | C fwd only sync %.3f"
      % (n, timed(step), timed(wrapper_only), timed(effective_only), timed(raw_sync), timed(raw_async), timed(raw_fwd_sync)))


def fwd_nograd():
    with torch.no_grad():
        m.negative_log_likelihood(X, Y)


def fwd_grad():
    m.negative_log_likelihood(X, Y)


val = m.negative_log_likelihood(X, Y)


def bwd_only():
    for p_ in m.parameters():
        p_.grad = None
    val.backward(retain_graph=True)


def zero_only():
    for p_ in m.parameters():
        p_.grad = None


def empty3():
    torch.empty((), device=dev); torch.empty((5,), device=dev); torch.empty((n, 1), device=dev)


print("      module (raw-parameter path): forward no_grad %.3f | forward recording %.3f | backward only %.3f | zeroing grads %.4f | 3 x torch.empty %.4f"
      % (timed(fwd_nograd), timed(fwd_grad), timed(bwd_only), timed(zero_only), timed(empty3)))
