"""`HOGP_simple` -- the high-order GP block of GAR (reference: FidelityFusion_Models/two_fidelity_models/
hogp_simple.py:15-126, used by GAR.py:27-31,60-66,94,119) on the device.

Same constructor, parameters (`noise_variance`, the shared kernel, `grid`, `mapping_vector`), cached attributes
(`K`, `K_eigen`, `A`, `g`) and return values: `log_likelihood` returns +NLL / (N * prod(d)) with the true pi,
`forward(x_train, x_test)` the posterior mean and the reference's variance expression.

What runs where: the covariance matrices come from the library's assembly (differentiable `kernel_matrix`), every
mode product -- the O(N^2 prod(d)) part: 550 GFLOP each at N = 8192, d = 64 x 64 -- runs on the fp64 matrix-core
GEMM (`functional.matmul_nt`, forward and backward).  Symmetric eigendecompositions: matrices up to 64 x 64 -- the
per-mode kernels -- run on the hand-written LDS Jacobi solver (`ffgp_syevj_small`); the N x N input kernel goes to
rocSOLVER (`torch.linalg.eigh` on the device: 0.66 s at N = 8192): a vendor-library call, not a hand-written kernel --
the one place on this path where that is so.  (A blocked one-sided Jacobi built from the batched Gram GEMM and the LDS
solver was written and measured: kernel matrices have condition numbers >= 1e9 and the Gram step squares them, so the
small eigenvalues never converge -- dropped; a QR-based block Jacobi is what "next" means here.)  Autograd chains the
pieces.

Kept quirks: ONE kernel module is shared by the input space and every output mode (:27-29); the per-mode grids are
0..d-1 as float columns; `forward` needs `log_likelihood` to have been called (it reads the cached `K`, `K_eigen`,
`A`, `g`); the "variance" is diag(K) + (A-weighted squared eigenvector products) (:60-75).  One deliberate difference:
`K_x.inverse() @ U_x` (:68) is evaluated as `U_x / lambda_x` -- the same matrix, without inverting a numerically
singular K_x.
"""
import math

import torch
import torch.nn as nn

from . import functional as F


class eigen_pairs:
    """matrices up to 64 x 64 (the per-mode kernels; tiny input sets) go to the hand-written LDS Jacobi solver
    (`ffgp_syevj_small`, ~80 us where rocSOLVER's syevd takes 1.7 ms); larger ones to rocSOLVER"""

    def __init__(self, matrix):
        if matrix.shape[0] <= 64 and matrix.is_cuda:
            self.value, self.vector = F.eigh_small(matrix)
        else:
            self.value, self.vector = torch.linalg.eigh(matrix, UPLO="U")


def mode_dot(t, M, mode):
    """mode-n product (contracts t's axis `mode` with M's axis 1) on the fp64 GEMM"""
    tm = t.movedim(mode, -1)
    r = F.matmul_nt(tm.reshape(-1, tm.shape[-1]), M)
    return r.reshape(*tm.shape[:-1], M.shape[0]).movedim(-1, mode)


def multi_mode_dot(t, Ms):
    for i, M in enumerate(Ms):
        t = mode_dot(t, M, i)
    return t


def _outer(vs):
    """tucker_to_tensor((1, [v_i as columns])): the outer product of the vectors"""
    out = vs[0].reshape(-1)
    for v in vs[1:]:
        out = out.unsqueeze(-1) * v.reshape(*([1] * out.dim()), -1)
    return out


class HOGP_simple(nn.Module):
    def __init__(self, kernel, noise_variance, output_shape, learnable_grid=False, learnable_map=False):
        super().__init__()
        self.noise_variance = nn.Parameter(torch.tensor([noise_variance]))
        self.K = []
        self.K_eigen = []
        self.kernel_list = nn.ModuleList([kernel for _ in range(len(output_shape) + 1)])   # the same module, shared
        self.grid = nn.ParameterList([nn.Parameter(torch.tensor(range(v)).reshape(-1, 1).float()) for v in output_shape])
        if learnable_grid is False:
            for p in self.grid:
                p.requires_grad = False
        self.mapping_vector = nn.ParameterList([nn.Parameter(torch.eye(v)) for v in output_shape])
        if learnable_map is False:
            for p in self.mapping_vector:
                p.requires_grad = False

    def _dev(self):
        return F._device_of(*list(self.parameters()))

    def _kernel(self, i, a, b):
        """kernel_list[i](a, b) on the device in fp64; a 1-column grid meets a D-dimensional ARD kernel through the
        same broadcast the reference's `x / length_scales` performs"""
        k = self.kernel_list[i]
        ls = getattr(k, "length_scales", None)
        if ls is not None and a.shape[1] == 1 and ls.numel() > 1:
            a, b = a.expand(-1, ls.numel()), b.expand(-1, ls.numel())
        return F.kernel_on_device(k, a, b)

    def log_likelihood(self, x_train, y_train):
        if isinstance(y_train, list):
            y_train = y_train[0]          # the variance part is not used (reference :83-86,108-109)
        dev = self._dev()
        y = y_train.to(device=dev, dtype=torch.float64)
        self.K.clear()
        self.K_eigen.clear()
        self.K.append(self._kernel(0, x_train, x_train))
        self.K_eigen.append(eigen_pairs(self.K[-1]))
        for i in range(len(self.kernel_list) - 1):
            _in = mode_dot(self.grid[i].to(device=dev, dtype=torch.float64), self.mapping_vector[i].to(device=dev, dtype=torch.float64), 0)
            self.K.append(self._kernel(i + 1, _in, _in))
            self.K_eigen.append(eigen_pairs(self.K[-1]))
        A = _outer([e.value for e in self.K_eigen])
        A = A + self.noise_variance.to(dev).pow(-1)
        T_1 = multi_mode_dot(y, [e.vector.T.contiguous() for e in self.K_eigen])
        T_3 = multi_mode_dot(T_1 * A.pow(-1 / 2), [e.vector for e in self.K_eigen])
        b = T_3.reshape(-1)
        g = multi_mode_dot(T_1 * A.pow(-1), [e.vector for e in self.K_eigen])
        self.A = A
        self.g = g
        nd = A.numel()
        loss = -0.5 * nd * math.log(2 * math.pi) - 0.5 * torch.log(A).sum() - 0.5 * (b * b).sum()
        loss = -loss / nd
        odt = y_train.dtype if y_train.dtype.is_floating_point else torch.float64
        return loss.to(device=y_train.device, dtype=odt)

    def forward(self, x_train, x_test):
        dev = self._dev()
        K_star = self._kernel(0, x_test, x_train)
        predict_u = multi_mode_dot(self.g, [K_star] + self.K[1:])
        n_dim = len(self.K_eigen) - 1
        diag_K_dims = _outer([K.diag() for K in self.K[1:]]).unsqueeze(0)
        diag_K_x = self._kernel(0, x_test, x_test).diag()
        for _ in range(n_dim):
            diag_K_x = diag_K_x.unsqueeze(-1)
        diag_K = diag_K_x * diag_K_dims
        S_2 = (self.A * self.A.pow(-1 / 2)).pow(2)
        e0 = self.K_eigen[0]
        ev_x = mode_dot(K_star, (e0.vector / e0.value.unsqueeze(0)).T.contiguous(), 1).pow(2)   # (K* K_x^-1 U_x)^2 = (K* U_x / lambda)^2
        evs = [ev_x] + [self.K_eigen[i + 1].vector.pow(2) for i in range(n_dim)]
        var_diag = diag_K + multi_mode_dot(S_2, evs)
        odt = x_test.dtype if x_test.dtype.is_floating_point else torch.float64
        return predict_u.to(device=x_test.device, dtype=odt), var_diag.to(device=x_test.device, dtype=odt)
