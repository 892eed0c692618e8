"""`cigp` -- the single-fidelity GP that AR / ResGP / NAR / CIGAR import `as GPR`
(reference: GaussianProcess/cigp_v10.py:17-69), on the fused HIP path.

Same constructor, parameter (`log_beta[1]`), call signatures, return shapes and quirks:
  * `negative_log_likelihood` returns +LL (= -nll) although it is called "negative" (:69);
  * the constant uses pi = 3.1415 (:15,68);
  * `y_train` may be `[y, y_var]` with y_var an N x N matrix of which only the diagonal enters Sigma (:59-60);
  * `forward` ignores y_var and adds the noise scalar to EVERY entry of the predictive covariance (:31-32,44).
"""
import torch
import torch.nn as nn

from . import functional as F


def _kfun(k):
    return k.kfun() if hasattr(k, "kfun") else (0, 1.0)

JITTER = 1e-6
EPS = 1e-10
PI = 3.1415


def _split(y_train):
    if isinstance(y_train, list):
        return y_train[0], y_train[1]
    return y_train, None


class cigp(nn.Module):
    def __init__(self, kernel, log_beta):
        super().__init__()
        self.kernel = kernel
        self.log_beta = nn.Parameter(torch.tensor([log_beta]))

    def forward(self, x_train, y_train, x_test):
        y_train, _ = _split(y_train)
        w, amp, clamp = self.kernel.effective()
        noise = self.log_beta.exp().pow(-1)
        mean, var = F.predict(x_train, y_train, x_test, w, amp, diag_add=noise + JITTER, clamp=clamp, full_cov=True,
                              var_add_all=float(noise), kfun=_kfun(self.kernel))
        return mean, var

    def negative_log_likelihood(self, x_train, y_train):
        y_train, y_var = _split(y_train)
        w, amp, clamp = self.kernel.effective()
        diag_add = self.log_beta.exp().pow(-1) + JITTER
        nll = F.nlml(x_train, y_train, w, amp, diag_add=diag_add, diag_vec=y_var, clamp=clamp, variant=F.FFGP_LL_V1,
                     pi_const=PI, **F._slot_args(), kfun=_kfun(self.kernel))
        return -nll
