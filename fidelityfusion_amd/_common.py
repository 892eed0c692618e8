"""Shared plumbing of the torch-facing operators: device selection, fp64 device copies, the size checks every raw device pointer
is preceded by, and the LinAlgError the reference's `torch.linalg.cholesky` raises.  (Split out of functional.py in round 4.)
"""
import ctypes as C

import torch

from . import _lib


NEG_INF = float("-inf")


def _device_of(*tensors):
    for t in tensors:
        if isinstance(t, torch.Tensor) and t.is_cuda:
            return t.device
    if not torch.cuda.is_available():
        raise _lib.FFGPError("fidelityfusion_amd needs an MI355X (gfx950) GPU; there is no CPU path")
    return torch.device("cuda", torch.cuda.current_device())


def _dev(t, dev):
    """fp64 contiguous copy/view of t on the compute device (detached)."""
    return t.detach().to(device=dev, dtype=torch.float64).contiguous()


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _check_xy(X, Y=None, what="x_train"):
    """The library reads raw device pointers: every size it derives them from is checked here first (the reference fails
    with a broadcast / solve error on the same mistakes; an unchecked mismatch would be an out-of-bounds device read)."""
    if X.dim() != 2:
        raise ValueError("%s must be 2-D [N, D], got shape %s" % (what, tuple(X.shape)))
    if Y is not None:
        if Y.dim() != 2:
            raise ValueError("y_train must be 2-D [N, d], got shape %s" % (tuple(Y.shape),))
        if Y.shape[0] != X.shape[0]:
            raise ValueError("y_train has %d rows for %d training inputs" % (Y.shape[0], X.shape[0]))


def _check_same_D(a, b, what="x_test"):
    if b.dim() != 2 or b.shape[1] != a.shape[1]:
        raise ValueError("%s must be [*, %d] like the training inputs, got shape %s" % (what, a.shape[1], tuple(b.shape)))


def _weights(w, D, dev):
    """[D] inverse length scales on the device: one value is broadcast over the input dimensions, D values are taken as
    they are, anything else (e.g. ARDKernel(input_dim=3) on 5-D inputs) is the caller's mistake."""
    wd = _dev(w.reshape(-1), dev)
    if wd.numel() == 1 and D > 1:
        wd = wd.expand(D).contiguous()
    if wd.numel() != D:
        raise ValueError("the kernel has %d length scales but the inputs have %d dimensions" % (wd.numel(), D))
    return wd


def _raise_not_pd(rc, what):
    raise torch.linalg.LinAlgError(
        "%s: The factorization could not be completed because the input is not positive-definite "
        "(the leading minor of order %d is not positive-definite)." % (what, rc))


def _split_kfun(kfun):
    """(id, float | tensor) -> ((id, float), tensor | None): a tensor parameter is differentiated (g_kparam)."""
    if isinstance(kfun[1], torch.Tensor):
        return (int(kfun[0]), float(kfun[1].detach())), kfun[1]
    return (int(kfun[0]), float(kfun[1])), None
