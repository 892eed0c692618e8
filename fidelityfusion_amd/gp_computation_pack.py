"""The "GP computation pack" (reference: GaussianProcess/gp_computation_pack.py) on libffgp.

    Gaussian_log_likelihood(y, cov, Kinv_method='cholesky3')        :34-91
    conditional_Gaussian(y, Sigma, K_s, K_ss, Kinv_method='cholesky3')  :93-118
    negative_log_likelihood(kernel, log_beta, x_train, y_train)     :120-136  (returns +LL)
    Tensor_linear(l_shape, h_shape)                                 :138-159  (CIGAR's learnable fidelity map)

'cholesky3' -- the method every model in the reference uses -- is the fused path; 'cholesky1', 'cholesky2' and
'direct' (alternative formulas with their own quirks, used by no model) are composed from the same device pieces;
the two `torch_distribution_MN*` branches (:85-88) read y's last axis as the event axis, so they only run when d == N and
then return N copies of the normalising constant -- built from the same device log-determinant; an unknown name raises
ValueError as in the reference (:91,116).
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


def _check_method(name, allowed, built):
    if name in built:
        return
    if name in allowed:
        raise NotImplementedError("Kinv_method=%r is not built on the HIP path (in the reference these two branches only "
                                  "run when N equals the output dimension)" % name)
    raise ValueError("Kinv_method should be either direct or cholesky")


def _logdet(cov):
    """log|cov| through the fused factorisation (differentiable): the V1 value of a zero column is sum(log L_ii) + const"""
    n = cov.shape[0]
    z = torch.zeros((n, 1), dtype=cov.dtype, device=cov.device)
    return 2.0 * (F.gaussian_nll_from_cov(z, cov, F.FFGP_LL_V1, math.pi) - 0.5 * n * math.log(2.0 * math.pi))


def Gaussian_log_likelihood(y, cov, Kinv_method="cholesky3"):
    """LL of N(0, cov).  'cholesky3' (what every model uses): the reference's Sigma^-2 quadratic form
    (gamma = cholesky_solve(y, L), :76-80); d == 1 returns shape [1, 1] as the reference does (:80), d > 1 a 0-dim
    tensor (:77).  The alternative formulas keep their own quirks: 'cholesky1' / 'direct' (:55-59,82-84) use the true
    y^T Sigma^-1 y but count the log-determinant twice and return a [d, d] matrix; 'cholesky2' (:60-63) the Sigma^-2
    form with the double log-determinant.  All are differentiable w.r.t. y and cov."""
    assert len(y.shape) == 2 and len(cov.shape) == 2, "y, mean, cov should be 2D tensors"
    _check_method(Kinv_method, _REFERENCE_METHODS, _REFERENCE_METHODS)
    if Kinv_method.startswith("torch_distribution_MN"):
        return _mvn_at_its_mean(y, cov)
    if Kinv_method == "cholesky3":
        ll = -F.gaussian_ll_v2(y, cov)   # differentiable w.r.t. y and cov (closed-form V2 gradients)
        return ll.reshape(1, 1) if y.shape[1] == 1 else ll
    quad, const = _alt_terms(y, cov, Kinv_method)
    return -0.5 * (quad + const)


def _mvn_at_its_mean(y, cov):
    """`MultivariateNormal(y, scale_tril=L | covariance_matrix=cov).log_prob(y)` (:85-88; gp_basic.py:147-151): the rows of
    y are N location vectors whose length must equal cov's size, and each is evaluated at itself -- N copies of
    -N/2 log 2 pi - 1/2 log|cov|, differentiable w.r.t. cov through the fused factorisation"""
    n = cov.shape[0]
    if y.shape[1] != n:   # torch.distributions refuses the broadcast the same way
        raise ValueError("torch_distribution_MN*: y's last dimension (%d) must equal the covariance size (%d)" % (y.shape[1], n))
    val = -0.5 * n * math.log(2.0 * math.pi) - 0.5 * _logdet(cov).to(y.device)
    return val.reshape(1).expand(y.shape[0]).to(y.dtype if y.dtype.is_floating_point else torch.float64)


def _alt_terms(y, cov, Kinv_method):
    """(quadratic form [d, d], 2 log|cov| + N log 2 pi) of the alternative formulas"""
    n, d = y.shape
    const = 2.0 * _logdet(cov).to(y.device) + n * math.log(2.0 * math.pi)
    if Kinv_method in ("cholesky1", "direct"):
        quad, _ = F.conditional_gaussian(y, cov, y, torch.zeros((d, d), dtype=y.dtype, device=y.device))   # y^T Sigma^-1 y
    else:
        eye = torch.eye(n, dtype=y.dtype, device=y.device)
        alpha, _ = F.conditional_gaussian(y, cov, eye, torch.zeros((n, n), dtype=y.dtype, device=y.device))  # Sigma^-1 y
        quad = F.matmul_nt(alpha.T.contiguous(), alpha.T.contiguous()).to(device=y.device, dtype=y.dtype)
    return quad, const


def conditional_Gaussian(y, Sigma, K_s, K_ss, Kinv_method="cholesky3"):
    """mu = K_s^T Sigma^-1 y ; cov = K_ss - (L^-1 K_s)^T (L^-1 K_s)   (:103-110).
    Both solves ride as passenger rows of one factorisation; differentiable w.r.t. all four arguments (closed-form
    backward on the saved factor), which is what acquisition optimisers differentiate through
    (Bayesian_optimization/cigp.py:52-70, acq.py:10-80)."""
    _check_method(Kinv_method, (), ("cholesky1", "cholesky3", "direct"))   # three spellings of the same quantities
    return F.conditional_gaussian(y, Sigma, K_s, K_ss)


def negative_log_likelihood(kernel, log_beta, x_train, y_train):
    """Sigma = K + exp(-log_beta) I + 1e-6 mean(K) I ; returns +LL with pi = 3.1415 (:120-136)."""
    lk = F.raw_path(kernel, x_train, y_train, log_beta)
    if lk is not None:   # everything already on the GPU in fp64: ONE library call on the raw parameters (exp(-log_beta) inside)
        return F.nlml_raw(x_train, y_train, lk, log_beta, F._lib.LINK_EXP_NEG, 0.0, mean_jitter=JITTER, variant=F.FFGP_LL_V1,
                          pi_const=PI, sign=-1.0)
    pr = kernel.pair() if hasattr(kernel, "pair") else None
    if pr is not None and F.pair_inputs_plain(x_train):   # Sum / Product of two library kernels: two descriptors, one assembly pass, one gradient pass
        return -F.nlml_pair(x_train, y_train, pr[0], pr[1], diag_add=log_beta.exp().pow(-1), mean_jitter=JITTER,
                            variant=F.FFGP_LL_V1, pi_const=PI, **F._slot_args())
    if not hasattr(kernel, "effective"):   # any other composition: Sigma is built on the device, then the fused factorisation
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
