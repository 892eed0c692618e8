"""`GP_basic` (reference: GaussianProcess/gp_basic.py:15-153) -- the noise_variance^2 convention used by CAR.

Sigma = K + noise_variance^2 I [+ the FULL y_var matrix when y_train = [y, y_var]] with no jitter (:63-65,117-119);
the 'cholesky3' branch is the fused path (the other spellings are composed from the same device pieces): forward = conditional Gaussian (:78-84), log_likelihood = the
Sigma^-2 form shared with gp_computation_pack.Gaussian_log_likelihood (:130-143).
"""
import math

import torch
import torch.nn as nn

from . import functional as F


def _kfun(k):
    return k.kfun() if hasattr(k, "kfun") else (0, 1.0)
from .gp_computation_pack import _check_method

_METHODS_FWD = ("cholesky1", "cholesky3", "direct")
_METHODS_LL = ("cholesky1", "cholesky2", "cholesky3", "direct", "torch_distribution_MN1", "torch_distribution_MN2")


def _split(y_train):
    if isinstance(y_train, list):
        return y_train[0], y_train[1]
    return y_train, None


class GP_basic(F.PosteriorCacheMixin, nn.Module):
    def __init__(self, kernel, noise_variance):
        super().__init__()
        self.kernel = kernel
        self.noise_variance = nn.Parameter(torch.tensor([noise_variance]))
        self._pcache = F.PosteriorCache()   # the factor of (x_train, y_train, parameters) between predictions

    def forward(self, x_train, y_train, x_test, Kinv_method="cholesky3"):
        _check_method(Kinv_method, (), _METHODS_FWD)   # three spellings of the same posterior (:66-88)
        y_train, y_var = _split(y_train)
        tree = hasattr(self.kernel, "fusable") and self.kernel.fusable()   # Sum / Product over library kernels: descriptor tree
        if not (hasattr(self.kernel, "effective") or (tree and y_var is None)) or torch.is_grad_enabled():
            return self._forward_composed(x_train, y_train, y_var, x_test)      # autograd on: differentiable composition
        if tree:
            def build():
                return F.Posterior(x_train, y_train, None, None, self.noise_variance.pow(2), first_query=x_test, tree=self.kernel.pair())
            post, fresh = self._pcache.get([x_train, y_train] + list(self.parameters()), build)
            mu, var = post.first if fresh else post.predict(x_test, full_cov=True)
            odt = y_train.dtype if y_train.dtype.is_floating_point else torch.float64
            return mu.to(device=y_train.device, dtype=odt).squeeze(), var.to(device=y_train.device, dtype=odt)
        w, amp, clamp = self.kernel.effective()
        if y_var is not None:
            mu, var = F.predict(x_train, y_train, x_test, w, amp, diag_add=self.noise_variance.pow(2), add_mat=y_var,
                                clamp=clamp, full_cov=True, var_add_all=0.0, kfun=_kfun(self.kernel))
            return mu.squeeze(), var
        # the reference refactorises on every call (:78-84); the factor is kept while the same tensors come back unchanged

        def build():
            return F.Posterior(x_train, y_train, w, amp, self.noise_variance.pow(2), clamp=clamp, kfun=_kfun(self.kernel),
                               first_query=x_test)
        post, fresh = self._pcache.get([x_train, y_train] + list(self.parameters()), build)
        mu, var = post.first if fresh else post.predict(x_test, full_cov=True)
        odt = y_train.dtype if y_train.dtype.is_floating_point else torch.float64
        return mu.to(device=y_train.device, dtype=odt).squeeze(), var.to(device=y_train.device, dtype=odt)

    # composed kernels (gp_basic.py:170-173 tries Linear / Sum kernels): Sigma is built on the device from the
    # differentiable kernel call and enters the fused factorisation as ffgp_problem.cov_dev
    def _sigma_composed(self, x_train, y_var):
        Sigma = F.add_diagonal(F.kernel_on_device(self.kernel, x_train, x_train), self.noise_variance.pow(2))
        if y_var is not None:
            Sigma = Sigma + y_var.to(device=Sigma.device, dtype=Sigma.dtype)
        return Sigma

    def _forward_composed(self, x_train, y_train, y_var, x_test):
        K_s = F.kernel_on_device(self.kernel, x_train, x_test)
        K_ss = F.kernel_on_device(self.kernel, x_test, x_test)
        post = None
        tree = hasattr(self.kernel, "fusable") and self.kernel.fusable()
        if y_var is None and (hasattr(self.kernel, "effective") or tree):
            # asked again with unchanged parameters (an acquisition loop): the factor of Sigma is the cached one and only the
            # closed-form backward runs; gradients reach the parameters and y as before
            def build():
                with torch.no_grad():
                    if tree:
                        return F.Posterior(x_train, y_train, None, None, self.noise_variance.pow(2), tree=self.kernel.pair())
                    w, amp, clamp = self.kernel.effective()
                    return F.Posterior(x_train, y_train, w, amp, self.noise_variance.pow(2), clamp=clamp, kfun=_kfun(self.kernel))
            post, _ = self._pcache.get([x_train, y_train] + list(self.parameters()), build)
        mu, var = F.conditional_gaussian(y_train, self._sigma_composed(x_train, y_var), K_s, K_ss, factor=post)
        odt = y_train.dtype if y_train.dtype.is_floating_point else torch.float64
        return mu.to(device=y_train.device, dtype=odt).squeeze(), var.to(device=y_train.device, dtype=odt)

    def log_likelihood(self, x_train, y_train, Kinv_method="cholesky3"):
        _check_method(Kinv_method, _METHODS_LL, _METHODS_LL)
        y_train, y_var = _split(y_train)
        if Kinv_method.startswith("torch_distribution_MN"):   # (:147-151) N copies of the normalising constant, d == N only
            from .gp_computation_pack import _mvn_at_its_mean
            return _mvn_at_its_mean(y_train, self._sigma_composed(x_train, y_var).to(y_train.device))
        if Kinv_method != "cholesky3":   # the alternative formulas (:120-129,141-143), composed from the same device pieces
            from .gp_computation_pack import _alt_terms
            quad, const = _alt_terms(y_train, self._sigma_composed(x_train, y_var).to(y_train.device), Kinv_method)
            return -0.5 * ((quad.sum() if Kinv_method == "cholesky2" else quad) + const)
        lk = F.raw_path(self.kernel, x_train, y_train, self.noise_variance, y_var)
        if lk is not None:   # everything already on the GPU in fp64: ONE library call on the raw parameters (noise_variance ** 2 inside)
            ll = F.nlml_raw(x_train, y_train, lk, self.noise_variance, F._lib.LINK_SQUARE, 0.0, add_mat=y_var, variant=F.FFGP_LL_V2,
                            pi_const=math.pi, sign=-1.0)
            return ll.reshape(1, 1) if y_train.shape[1] == 1 else ll
        pr = self.kernel.pair() if hasattr(self.kernel, "pair") else None
        if pr is not None and F.pair_inputs_plain(x_train, y_var):   # Sum / Product of two library kernels (:170-173): two descriptors, fused like a single kernel
            ll = -F.nlml_pair(x_train, y_train, pr[0], pr[1], diag_add=self.noise_variance.pow(2), add_mat=y_var,
                              variant=F.FFGP_LL_V2, pi_const=math.pi, **F._slot_args())
            return ll.reshape(1, 1) if y_train.shape[1] == 1 else ll
        if not hasattr(self.kernel, "effective") or not F.pair_inputs_plain(x_train, y_var):
            # (also: learnable inputs or a gradient-carrying y_var -- the fused call has no gradients for them)
            ll = -F.gaussian_nll_from_cov(y_train, self._sigma_composed(x_train, y_var), F.FFGP_LL_V2, math.pi)
            return ll.reshape(1, 1) if y_train.shape[1] == 1 else ll
        w, amp, clamp = self.kernel.effective()
        nll = F.nlml(x_train, y_train, w, amp, diag_add=self.noise_variance.pow(2), add_mat=y_var, clamp=clamp,
                     variant=F.FFGP_LL_V2, pi_const=math.pi, **F._slot_args(), kfun=_kfun(self.kernel))
        ll = -nll
        return ll.reshape(1, 1) if y_train.shape[1] == 1 else ll
