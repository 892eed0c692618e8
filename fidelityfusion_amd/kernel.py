"""Kernel modules with the reference's names, parameters and call signature
(`GaussianProcess/kernel.py`: ARDKernel :65-105, SquaredExponentialKernel :239-272); the covariance itself is
assembled by the HIP library.

Each module owns the same raw nn.Parameters as the reference (names show up in state_dict logs,
`FidelityFusion_Models/log/ResGP/train.log:2`) and exposes `effective()` -> (w, amp, clamp): the inverse length
scales, amplitude and squared-distance clamp of the generic form libffgp evaluates,
    K_ij = amp * exp(-1/2 * max(sum_k ((x_ik - x_jk) w_k)^2, clamp)).
The raw -> effective maps are plain torch ops, so autograd carries the closed-form gradients the library
returns back to the raw parameters (abs/exp chain rule included).
"""
import torch
import torch.nn as nn

from . import functional as F

EPS = 1e-9


class _StationaryKernel(nn.Module):
    def effective(self):  # pragma: no cover - interface
        raise NotImplementedError

    def kfun(self):
        """(radial profile id, profile parameter) -- include/ffgp.h FFGP_KFUN_*; squared exponential by default."""
        return (0, 1.0)

    def forward(self, x1, x2):
        """Covariance matrix [n1, n2]; differentiable w.r.t. the kernel parameters (the fused likelihood in cigp /
        gp_computation_pack does not go through this call -- it assembles Sigma and its gradient in one pass)."""
        w, amp, clamp = self.effective()
        return F.kernel_matrix(x1, x2, w, amp, clamp, self.kfun())


class ARDKernel(_StationaryKernel):
    """K = |signal_variance| * exp(-1/2 * cdist(x1/l, x2/l)^2), l = |length_scales| + eps (kernel.py:100-105).
    torch.cdist clamps the squared distance at 1e-30 before its sqrt; that clamp is kept."""

    def __init__(self, input_dim, initial_length_scale=1.0, initial_signal_variance=1.0, eps=EPS):
        super().__init__()
        self.length_scales = nn.Parameter(torch.ones(input_dim) * initial_length_scale)
        self.signal_variance = nn.Parameter(torch.tensor([initial_signal_variance]))
        self.eps = eps

    def effective(self):
        return 1.0 / (self.length_scales.abs() + self.eps), self.signal_variance.abs(), 1e-30


class SquaredExponentialKernel(_StationaryKernel):
    """K = exp(signal_variance)^2 * exp(-1/2 * sqdist / exp(length_scale)^2), scalar length scale, both raw
    parameters are logs (kernel.py:253-272).  No clamp on the distance."""

    def __init__(self, length_scale=1.0, signal_variance=1.0):
        super().__init__()
        self.length_scale = nn.Parameter(torch.tensor([length_scale]))
        self.signal_variance = nn.Parameter(torch.tensor([signal_variance]))

    def effective(self):
        return torch.exp(-self.length_scale), self.signal_variance.exp().pow(2), F.NEG_INF


class MaternKernel(_StationaryKernel):
    """Matern kernel with independent length scales, nu in {0.5, 1.5, 2.5} and the reference's extra `rho`
    (GaussianProcess/kernel.py:109-169): K = |signal_variance| * phi_nu(cdist(x1/l, x2/l)^2 ; rho)."""

    _KFUN = {0.5: 1, 1.5: 2, 2.5: 3}

    def __init__(self, input_dim, initial_length_scale=1.0, initial_signal_variance=1.0, nu=2.5, rho=1, eps=EPS):
        super().__init__()
        self.length_scales = nn.Parameter(torch.ones(input_dim) * initial_length_scale)
        self.signal_variance = nn.Parameter(torch.tensor([initial_signal_variance]))
        self.eps = eps
        self.nu = nu
        self.rho = rho

    def effective(self):
        return 1.0 / (self.length_scales.abs() + self.eps), self.signal_variance.abs(), 1e-30

    def kfun(self):
        if self.nu not in self._KFUN:   # the reference returns None for any other nu (kernel.py:161-166)
            raise ValueError("MaternKernel: nu must be 0.5, 1.5 or 2.5")
        return (self._KFUN[self.nu], float(self.rho))
