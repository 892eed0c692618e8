"""Kernel modules with the reference's names, parameters and call signature
(`GaussianProcess/kernel.py`: LinearKernel :23-63, ARDKernel :65-105, MaternKernel :109-169, SumKernel :172-203,
ProductKernel :205-236, SquaredExponentialKernel :239-272, RationalQuadraticKernel :275-310); the covariance itself
is assembled by the HIP library.  MaternKernel_scalarLengthScale (:312-347) is provided in the reference's own arithmetic
(norm-expansion distance, matrix product on the fp64 GEMM), NaN behaviour included -- see its docstring.

Each module owns the same raw nn.Parameters as the reference (names show up in state_dict logs,
`FidelityFusion_Models/log/ResGP/train.log:2`) and exposes `effective()` -> (w, amp, clamp): the inverse length
scales, amplitude and squared-distance clamp of the generic form libffgp evaluates,
    K_ij = amp * exp(-1/2 * max(sum_k ((x_ik - x_jk) w_k)^2, clamp)).
The raw -> effective maps are plain torch ops, so autograd carries the closed-form gradients the library
returns back to the raw parameters (abs/exp chain rule included).
"""
import torch
import torch.nn as nn

from . import _lib
from . import functional as F

EPS = 1e-9


class _StationaryKernel(nn.Module):
    _ffgp_device_aware = True

    def effective(self):  # pragma: no cover - interface
        raise NotImplementedError

    def kfun(self):
        """(radial profile id, profile parameter) -- include/ffgp.h FFGP_KFUN_*; squared exponential by default."""
        return (0, 1.0)

    def links(self):
        """the raw parameters and the elementwise maps to (w, amp) as the library's link ids (functional.nlml_raw), or None"""
        return None

    def descriptor(self):
        """this kernel as one part of a composed kernel (include/ffgp.h ffgp_kdesc)"""
        w, amp, clamp = self.effective()
        kf = self.kfun()
        return {"kfun": kf[0], "w": w, "amp": amp, "clamp": clamp, "kparam": kf[1], "center": None}

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

    def links(self):
        return {"w": self.length_scales, "w_link": _lib.LINK_INV_ABS_EPS, "w_c": float(self.eps), "amp": self.signal_variance,
                "amp_link": _lib.LINK_ABS, "clamp": 1e-30, "kfun": 0}


class SquaredExponentialKernel(_StationaryKernel):
    """K = exp(signal_variance)^2 * exp(-1/2 * sqdist / exp(length_scale)^2), scalar length scale, both raw
    parameters are logs (kernel.py:253-272).  No clamp on the distance."""

    def __init__(self, length_scale=1.0, signal_variance=1.0):
        super().__init__()
        self.length_scale = nn.Parameter(torch.tensor([length_scale]))
        self.signal_variance = nn.Parameter(torch.tensor([signal_variance]))

    def effective(self):
        return torch.exp(-self.length_scale), self.signal_variance.exp().pow(2), F.NEG_INF

    def links(self):
        return {"w": self.length_scale, "w_link": _lib.LINK_EXP_NEG, "w_c": 0.0, "amp": self.signal_variance,
                "amp_link": _lib.LINK_EXP_SQ, "clamp": F.NEG_INF, "kfun": 0}


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

    def links(self):
        if self.nu not in self._KFUN:
            return None
        return {"w": self.length_scales, "w_link": _lib.LINK_INV_ABS_EPS, "w_c": float(self.eps), "amp": self.signal_variance,
                "amp_link": _lib.LINK_ABS, "clamp": 1e-30, "kfun": self._KFUN[self.nu], "kparam": float(self.rho)}

    def kfun(self):
        if self.nu not in self._KFUN:   # the reference returns None for any other nu (kernel.py:161-166)
            raise ValueError("MaternKernel: nu must be 0.5, 1.5 or 2.5")
        return (self._KFUN[self.nu], float(self.rho))


class RationalQuadraticKernel(_StationaryKernel):
    """K = signal_variance^2 * (1 + sqdist / (2 alpha length_scale^2))^-alpha, scalar length scale, all three raw
    parameters learnable, no clamp on the distance (kernel.py:275-310).  alpha's gradient comes back through
    ffgp_grads.g_kparam_dev."""

    def __init__(self, length_scale=1., signal_variance=1., alpha=1.):
        super().__init__()
        self.length_scale = nn.Parameter(torch.tensor([length_scale]))
        self.signal_variance = nn.Parameter(torch.tensor([signal_variance]))
        self.alpha = nn.Parameter(torch.tensor([alpha]))

    def effective(self):
        return 1.0 / self.length_scale, self.signal_variance.pow(2), F.NEG_INF

    def kfun(self):
        return (4, self.alpha)


class LinearKernel(nn.Module):
    """K = |signal_variance| * ((x1 - center) / length_scales) ((x2 - center) / length_scales)^T  (kernel.py:23-63);
    the product runs on the fp64 matrix-core GEMM, forward and backward."""
    _ffgp_device_aware = True

    def __init__(self, input_dim, initial_length_scale=1.0, initial_signal_variance=1.0):
        super().__init__()
        self.length_scales = nn.Parameter(torch.ones(input_dim) * initial_length_scale)
        self.signal_variance = nn.Parameter(torch.tensor([initial_signal_variance]))
        self.center = nn.Parameter(torch.zeros(input_dim))

    def descriptor(self):
        """K = amp * sum_k w_k^2 (x_k - c_k)(x'_k - c_k) with w = 1 / length_scales (raw, as the reference divides), amp = |sv|"""
        return {"kfun": F.FFGP_KFUN_LINEAR, "w": 1.0 / self.length_scales, "amp": self.signal_variance.abs(), "clamp": F.NEG_INF,
                "kparam": 1.0, "center": self.center}

    def forward(self, x1, x2):
        c, ls = self.center.to(x1.device), self.length_scales.to(x1.device)
        z1, z2 = (x1 - c) / ls, (x2 - c) / ls
        K = F.matmul_nt(z1, z2)
        odt = x1.dtype if x1.dtype.is_floating_point else torch.float64
        K = K.to(device=x1.device, dtype=odt)
        return K * self.signal_variance.abs().to(K.device)


class _Pair(nn.Module):
    def __init__(self, kernel1, kernel2):
        super().__init__()
        self.kernel1 = kernel1
        self.kernel2 = kernel2

    @property
    def _ffgp_device_aware(self):
        return all(getattr(k, "_ffgp_device_aware", False) for k in (self.kernel1, self.kernel2))

    def pair(self):
        """(leaf descriptors, operator spec) when every leaf of this composition is a kernel the library evaluates itself (the
        stationary profiles and LinearKernel) and there are at most four of them -- then kernel, Sigma extras and all gradients
        are ONE tile pass each (ffgp_assemble_tree, ffgp_problem.tree).  Two leaves: spec = FFGP_KOP_*; nested Sum / Product
        objects (the reference composes arbitrary modules, kernel.py:172-236): spec = (shape, ops) of the canonical forms in
        include/ffgp.h, reached by swapping the operands of commutative nodes.  None for anything else (more leaves, user
        modules): those are evaluated part by part on the device and composed there."""
        if not FUSE_PAIRS:
            return None
        flat = _flatten(self)
        if flat is None:
            return None
        leaves, form = flat
        descs = [k.descriptor() for k in leaves]
        if len(descs) == 2:
            return descs, form[1][0]
        return descs, form

    def fusable(self):
        """`pair()` would return descriptors (checked on the module structure alone: no tensor is touched)"""
        return FUSE_PAIRS and _flatten(self) is not None

    def forward(self, x1, x2):
        pr = self.pair()
        if pr is not None and x1.dim() == 2 and x2.dim() == 2:
            return F.kernel_pair(x1, x2, pr[0], pr[1])   # (input gradients: ffgp_kernel_input_weights_tree in its backward)
        return self._compose(self.kernel1(x1, x2), self.kernel2(x1, x2))


def _flatten(k):
    """a composition as (leaf modules in canonical order, (shape, ops)) -- include/ffgp.h ffgp_ktree -- or None"""
    if not isinstance(k, _Pair):
        return ([k], None) if hasattr(k, "descriptor") else None
    L, R = _flatten(k.kernel1), _flatten(k.kernel2)
    if L is None or R is None:
        return None
    (ll, lf), (rl, rf) = L, R
    if len(ll) < len(rl):      # a + b == b + a and a * b == b * a bit for bit: put the deeper operand first
        (ll, lf), (rl, rf) = (rl, rf), (ll, lf)
    n1, n2, op = len(ll), len(rl), k._OP
    if (n1, n2) == (1, 1):
        return ll + rl, (F.FFGP_TREE_CHAIN, (op,))
    if (n1, n2) == (2, 1):
        return ll + rl, (F.FFGP_TREE_CHAIN, (lf[1][0], op))
    if (n1, n2) == (2, 2):
        return ll + rl, (F.FFGP_TREE_BALANCED, (lf[1][0], rf[1][0], op))
    if (n1, n2) == (3, 1):
        return ll + rl, (F.FFGP_TREE_CHAIN, (lf[1][0], lf[1][1], op))
    return None


FUSE_PAIRS = True   # False: always compose part by part (tests compare the two paths)


class SumKernel(_Pair):
    """kernel1(x1, x2) + kernel2(x1, x2)  (kernel.py:172-203)."""
    _OP = F.FFGP_KOP_SUM

    @staticmethod
    def _compose(a, b):
        return a + b


class ProductKernel(_Pair):
    """kernel1(x1, x2) * kernel2(x1, x2)  (kernel.py:205-236)."""
    _OP = F.FFGP_KOP_PRODUCT

    @staticmethod
    def _compose(a, b):
        return a * b


class MaternKernel_scalarLengthScale(nn.Module):
    """K = signal_variance^2 * (1 + sqrt(3 sqdist) / length_scale^2)^(-nu), sqdist = |x1|^2 + |x2|^2 - 2 x1 x2^T, all three
    raw parameters learnable (`nu` too) -- GaussianProcess/kernel.py:312-347, formula at :346-347 (despite its name it is
    not a Matern covariance).

    The distance is the UNCLAMPED norm expansion: wherever rounding leaves it negative -- on the diagonal of
    kernel(x, x), where the three terms cancel to +-1e-16 -- sqrt returns NaN, in the reference and here alike (which
    entries go negative depends on the summation order of the matrix product, so the NaN PATTERN is not reproducible
    between any two BLAS builds; the reference's own demos never use this kernel on coincident points).  Kept as the
    reference wrote it rather than repaired: cross-covariances between distinct points are finite and pinned by a
    fixture.  The matrix product runs on the fp64 matrix-core GEMM (forward and backward), the elementwise tail in torch
    on the device, so autograd reaches length_scale, signal_variance, nu and the inputs as in the reference."""
    _ffgp_device_aware = True

    def __init__(self, length_scale=1.0, signal_variance=1.0, nu=2.5):
        super().__init__()
        self.length_scale = nn.Parameter(torch.tensor([length_scale]))
        self.signal_variance = nn.Parameter(torch.tensor([signal_variance]))
        self.nu = nn.Parameter(torch.tensor([nu]))

    def forward(self, x1, x2):
        dev = F._device_of(x1, x2)
        a, b = x1.to(device=dev, dtype=torch.float64), x2.to(device=dev, dtype=torch.float64)
        sqdist = (a * a).sum(1).reshape(-1, 1) + (b * b).sum(1) - 2.0 * F.matmul_nt(a, b)
        ls, sv, nu = (p.to(device=dev, dtype=torch.float64) for p in (self.length_scale, self.signal_variance, self.nu))
        K = sv.pow(2) * torch.pow(1.0 + torch.sqrt(3.0 * sqdist) / ls.pow(2), -nu)
        odt = x1.dtype if x1.dtype.is_floating_point else torch.float64
        return K.to(device=x1.device, dtype=odt)
