"""`CIGP_withMean` -- a GP with a learnable mean function (reference: GaussianProcess/cigp_withMean.py:29-64; the same
shape as Bayesian_optimization/cigp.py:32-87) on the device path.

Same constructor (`input_dim, output_dim, kernel, noise_variance`), the same small MLP mean (`mean_func`: Linear(input_dim,
5) -> LeakyReLU -> Linear(5, output_dim), so reference state_dicts load), `forward(x_train, y_train, x_test)` =
conditional Gaussian of the residual y - m(x) plus m(x_test) (:44-57), `log_likelihood` = the Sigma^-2 form of
`Gaussian_log_likelihood` on the residual (:59-62).  Sigma = K + noise_variance^2 I is composed on the device (no N x N
identity) and factorised by the fused path; everything is differentiable, the mean's parameters included.
"""
import torch
import torch.nn as nn

from . import functional as F
from . import gp_computation_pack as gp_pack


def zeroMean(x):
    return torch.zeros(x.shape[0], 3)


class constMean(nn.Module):
    def __init__(self, output_dim):
        super().__init__()
        self.mean = nn.Parameter(torch.zeros(output_dim))

    def forward(self, x):
        return self.mean.expand(x.shape[0], -1)


class CIGP_withMean(nn.Module):
    def __init__(self, input_dim, output_dim, kernel, noise_variance):
        super().__init__()
        self.kernel = kernel
        self.noise_variance = nn.Parameter(torch.tensor([noise_variance]))
        self.mean_func = nn.Sequential(nn.Linear(input_dim, 5), nn.LeakyReLU(), nn.Linear(5, output_dim))

    def _sigma(self, x_train):
        return F.add_diagonal(F.kernel_on_device(self.kernel, x_train, x_train), self.noise_variance.pow(2))

    def forward(self, x_train, y_train, x_test):
        K_s = F.kernel_on_device(self.kernel, x_train, x_test)
        K_ss = F.kernel_on_device(self.kernel, x_test, x_test)
        mean_part_train = self.mean_func(x_train)
        mean_part_test = self.mean_func(x_test)
        mu, cov = gp_pack.conditional_Gaussian(y_train - mean_part_train, self._sigma(x_train), K_s, K_ss)
        return mu + mean_part_test.to(mu.device), cov

    def log_likelihood(self, x_train, y_train):
        return gp_pack.Gaussian_log_likelihood(y_train - self.mean_func(x_train), self._sigma(x_train).to(y_train.device))
