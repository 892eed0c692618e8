"""The "GP computation pack" (reference: GaussianProcess/gp_computation_pack.py) on libffgp.

    Gaussian_log_likelihood(y, cov, Kinv_method='cholesky3')        :34-91
    conditional_Gaussian(y, Sigma, K_s, K_ss, Kinv_method='cholesky3')  :93-118
    negative_log_likelihood(kernel, log_beta, x_train, y_train)     :120-136  (returns +LL)
    Tensor_linear(l_shape, h_shape)                                 :138-159  (CIGAR's learnable fidelity map)

Only the 'cholesky3' method -- the one every model in the reference uses -- is implemented; the other
`Kinv_method`s of the reference are alternative formulas for the same quantities (explicit inverses, d=1 only)
and raise NotImplementedError here, an unknown name raises ValueError as in the reference (:91,116).
"""
import math

import torch

from . import functional as F


def _kfun(k):
    return k.kfun() if hasattr(k, "kfun") else (0, 1.0)

EPS = 1e-9
JITTER = 1e-6
PI = 3.1415

_REFERENCE_METHODS = ("cholesky1", "cholesky2", "cholesky3", "direct", "torch_distribution_MN1", "torch_distribution_MN2")


def _check_method(name, allowed):
    if name == "cholesky3":
        return
    if name in allowed:
        raise NotImplementedError("Kinv_method=%r: only 'cholesky3' is built on the HIP path" % name)
    raise ValueError("Kinv_method should be either direct or cholesky")


def Gaussian_log_likelihood(y, cov, Kinv_method="cholesky3"):
    """LL of N(0, cov) with the reference's Sigma^-2 quadratic form (gamma = cholesky_solve(y, L), :76-80).
    d == 1 returns shape [1, 1] as the reference does (:80), d > 1 a 0-dim tensor (:77)."""
    assert len(y.shape) == 2 and len(cov.shape) == 2, "y, mean, cov should be 2D tensors"
    _check_method(Kinv_method, _REFERENCE_METHODS)
    ll = -F.gaussian_ll_v2(y, cov)   # differentiable w.r.t. y and cov (closed-form V2 gradients)
    return ll.reshape(1, 1) if y.shape[1] == 1 else ll


def conditional_Gaussian(y, Sigma, K_s, K_ss, Kinv_method="cholesky3"):
    """mu = K_s^T Sigma^-1 y ; cov = K_ss - (L^-1 K_s)^T (L^-1 K_s)   (:103-110).
    Both solves ride as passenger rows of one factorisation; differentiable w.r.t. all four arguments (closed-form
    backward on the saved factor), which is what acquisition optimisers differentiate through
    (Bayesian_optimization/cigp.py:52-70, acq.py:10-80)."""
    _check_method(Kinv_method, ("cholesky1", "cholesky3", "direct"))
    return F.conditional_gaussian(y, Sigma, K_s, K_ss)


def negative_log_likelihood(kernel, log_beta, x_train, y_train):
    """Sigma = K + exp(-log_beta) I + 1e-6 mean(K) I ; returns +LL with pi = 3.1415 (:120-136)."""
    if not hasattr(kernel, "effective"):   # composed kernel: Sigma is built on the device, then the fused factorisation
        K = F.kernel_on_device(kernel, x_train, x_train)
        Sigma = F.add_diagonal(K, log_beta.exp().pow(-1), JITTER * K.mean())
        return -F.gaussian_nll_from_cov(y_train, Sigma, F.FFGP_LL_V1, PI)
    w, amp, clamp = kernel.effective()
    nll = F.nlml(x_train, y_train, w, amp, diag_add=log_beta.exp().pow(-1), mean_jitter=JITTER, clamp=clamp,
                 variant=F.FFGP_LL_V1, pi_const=PI, **F._slot_args(), kfun=_kfun(kernel))
    return -nll


class Tensor_linear(torch.nn.Module):
    """CIGAR's learnable map from the low-fidelity output shape to the high-fidelity one (reference :138-159;
    `FidelityFusion_Models/CIGAR.py:33-36,75,122`): one matrix per output mode, initialised to the identity, or to
    its bilinear interpolation when the high fidelity is finer.  The product -- N x d_l x d_h, 17 GFLOP per call at
    config 4's d = 1024, inside every training step together with its two backward products -- runs on the fp64
    matrix-core GEMM.  Kept quirk: every mode product is applied to the INPUT, not to the running result, so only
    the last mode's matrix acts (:156-158)."""

    def __init__(self, l_shape, h_shape):
        super().__init__()
        self.l_shape = l_shape
        self.h_shape = h_shape
        vectors = []
        for i in range(len(self.l_shape)):
            if self.l_shape[i] < self.h_shape[i]:
                init = torch.eye(self.l_shape[i])
                init = torch.nn.functional.interpolate(init.reshape(1, 1, *init.shape), (self.l_shape[i], self.h_shape[i]),
                                                       mode="bilinear")
                init = init.squeeze().T
            elif self.l_shape[i] == self.h_shape[i]:
                init = torch.eye(self.l_shape[i])
            else:   # the reference leaves init_tensor unbound here (NameError / stale value)
                raise ValueError("Tensor_linear: the high-fidelity shape must not be coarser than the low-fidelity one")
            vectors.append(torch.nn.Parameter(init))
        self.vectors = torch.nn.ParameterList(vectors)

    def forward(self, x):
        i = len(self.l_shape) - 1
        V = self.vectors[i]                       # [h_i, l_i]
        xm = x.movedim(i + 1, -1)
        y = F.matmul_nt(xm.reshape(-1, xm.shape[-1]), V)
        odt = x.dtype if x.dtype.is_floating_point else torch.float64
        y = y.to(device=x.device, dtype=odt)
        return y.reshape(*xm.shape[:-1], V.shape[0]).movedim(-1, i + 1)
