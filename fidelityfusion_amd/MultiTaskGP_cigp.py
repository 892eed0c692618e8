"""`CIGP` of GaussianProcess/MultiTaskGP_cigp.py:14-50 -- the single-kernel multi-output GP written out by hand there --
on the device path.

`forward` (:20-34): the conditional Gaussian with Sigma = K + noise_variance^2 I, the covariance reduced to its diagonal
and expanded to the mean's shape, the mean squeezed.  `log_likelihood` (:36-50) keeps the file's own formula:
-1/2 (||Sigma^-1 y||_F^2 + log|Sigma| + N log 2 pi) -- the Sigma^-2 quadratic form of `cholesky_solve`, and the
log-determinant / constant counted ONCE whatever the number of outputs.
"""
import math

import torch
import torch.nn as nn

from . import functional as F
from . import gp_computation_pack as gp_pack


class CIGP(nn.Module):
    def __init__(self, kernel, noise_variance):
        super().__init__()
        self.kernel = kernel
        self.noise_variance = nn.Parameter(torch.tensor([noise_variance]))

    def _sigma(self, x_train):
        return F.add_diagonal(F.kernel_on_device(self.kernel, x_train, x_train), self.noise_variance.pow(2))

    def forward(self, x_train, y_train, x_test):
        K_s = F.kernel_on_device(self.kernel, x_train, x_test)
        K_ss = F.kernel_on_device(self.kernel, x_test, x_test)
        mu, cov = gp_pack.conditional_Gaussian(y_train, self._sigma(x_train), K_s, K_ss)
        cov = cov.diag().view(-1, 1).expand_as(mu)
        return mu.squeeze(), cov

    def log_likelihood(self, x_train, y_train):
        Sigma = self._sigma(x_train)
        d = y_train.shape[1]
        ll = gp_pack.Gaussian_log_likelihood(y_train, Sigma.to(y_train.device)).reshape(())   # -1/2 (||a||^2 + d logdet + N d log 2 pi)
        if d > 1:    # the file counts log|Sigma| + N log 2 pi once (:50)
            ll = ll + 0.5 * (d - 1) * (gp_pack._logdet(Sigma).to(ll.device) + len(x_train) * math.log(2.0 * math.pi))
        return ll
