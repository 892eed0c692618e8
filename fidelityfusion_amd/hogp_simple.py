"""`HOGP_simple` -- the high-order GP block of GAR (reference: FidelityFusion_Models/two_fidelity_models/
hogp_simple.py:15-126, used by GAR.py:27-31,60-66,94,119) on the device.

Same constructor, parameters (`noise_variance`, the shared kernel, `grid`, `mapping_vector`), cached attributes
(`K`, `K_eigen`, `A`, `g`) and return values: `log_likelihood` returns +NLL / (N * prod(d)) with the true pi,
`forward(x_train, x_test)` the posterior mean and the reference's variance expression.

What runs where: the covariance matrices come from the library's assembly (differentiable `kernel_matrix`), every
mode product -- the O(N^2 prod(d)) part: 550 GFLOP each at N = 8192, d = 64 x 64 -- runs on the fp64 matrix-core
GEMM (`functional.matmul_nt`, forward and backward).  Symmetric eigendecompositions: matrices up to 64 x 64 -- the
per-mode kernels -- run on the hand-written LDS Jacobi solver (`ffgp_syevj_small`); the N x N input kernel runs on the
library's two-stage solver (`eigh.eigh` -> `ffgp_syevd`: band reduction, bulge chasing, divide & conquer, two
back-transformations; 0.26 s at N = 8192 where rocSOLVER's syevd takes 0.66 s).  No vendor library is called anywhere in
this module (the GPU tests substitute their own `eigen_pairs` built on torch.linalg.eigh as the comparator); "jacobi" is the
round-2 block Jacobi (slow, independent cross-check).  The likelihood's backward is closed-form (`_KronNLL`): GEMMs only,
no differentiation through `eigh`.

Kept quirks: ONE kernel module is shared by the input space and every output mode (:27-29); the per-mode grids are
0..d-1 as float columns; `forward` needs `log_likelihood` to have been called (it reads the cached `K`, `K_eigen`,
`A`, `g`); the "variance" is diag(K) + (A-weighted squared eigenvector products) (:60-75).

`variance_mode` (constructor keyword, attribute): "explicit_inverse" (default; "reference" is accepted as its old name) evaluates
`K_star @ K_x.inverse() @ U_x` in the reference's order of operations (:68) with an EXPLICIT inverse of the jitter-free kernel
matrix followed by two GEMMs -- but the inverse is NOT the reference's LU inverse: it is formed from the eigenpairs the likelihood
call already produced, K_x^-1 = (U / lambda) U^T on the fp64 GEMM (the reference's `.inverse()` is LAPACK's LU; on these
numerically singular matrices -- cond 5e6 ... 1e13 on the fixtures -- any two inverses agree to ~cond * eps, which is also what
the fixtures of the reference's own LU hold this to; tiny or negative computed eigenvalues are amplified by 1 / lambda exactly as
they are by any other inverse of such a matrix).  "eigen"
evaluates the same matrix as `K_star @ (U_x / lambda_x)` without forming the inverse.  Which one is "right" is moot (the
expression is not a variance); the default keeps the reference's arithmetic shape.
"""
import math

import torch
import torch.nn as nn

from . import functional as F


EIGENSOLVER = "ffgp"    # n > 64: "ffgp" = the library's two-stage solver (default); "jacobi" = its block Jacobi (slow cross-check)


class eigen_pairs:
    """matrices up to 64 x 64 (the per-mode kernels; tiny input sets) go to the hand-written LDS Jacobi solver
    (`ffgp_syevj_small`), larger ones -- the N x N input kernel -- to the library's two-stage solver (`ffgp_syevd`).
    Reference: `eigen_pairs`, two_fidelity_models/hogp_simple.py:15-19.

    There is no vendor or CPU route: the matrix must live on the MI355X.  The pairs of an n > 64 matrix are computed on a
    DETACHED copy (`value`, `vector` carry no autograd history): the likelihood's gradient is the closed form of `_KronNLL`, which
    never differentiates through eigenvectors, and `forward()`'s variance expression treats the cached pairs as constants, as the
    reference's cached `K_eigen` of the last likelihood call are used there (:60-75)."""

    def __init__(self, matrix):
        if not matrix.is_cuda:
            raise F._lib.FFGPError("eigen_pairs: the matrix must be on the GPU (fidelityfusion_amd has no CPU path)")
        if matrix.shape[0] <= 64:
            self.value, self.vector = F.eigh_small(matrix)
        elif EIGENSOLVER in ("ffgp", "jacobi"):
            from . import eigh as _eigh
            with torch.no_grad():
                fn = _eigh.eigh if EIGENSOLVER == "ffgp" else _eigh.jacobi_eigh
                self.value, self.vector = fn(matrix.detach().to(torch.float64))
        else:
            raise ValueError("unknown EIGENSOLVER %r" % (EIGENSOLVER,))


class _InvFromEigen(torch.autograd.Function):
    """K^-1 = (U / lambda) U^T from the eigenpairs of the symmetric K (explicit inverse on the fp64 GEMM); backward
    d K = -K^-1 (d K^-1) K^-1"""

    @staticmethod
    def forward(ctx, K, lam, U):
        Kinv = F.matmul_nt(U / lam.unsqueeze(0), U)
        ctx.save_for_backward(Kinv)
        return Kinv

    @staticmethod
    def backward(ctx, dKi):
        (Kinv,) = ctx.saved_tensors
        t = F.matmul_nt(Kinv, dKi.T.contiguous())          # K^-1 dKi   (K^-1 is symmetric)
        return -F.matmul_nt(t, Kinv), None, None


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


class _KronNLL(torch.autograd.Function):
    """+NLL / numel of vec(Y) ~ N(0, K_0 (x) K_1 (x) ... + diag(tau)) through the per-mode eigendecompositions
    (hogp_simple.py:92-117 of the reference), with a closed-form backward instead of autograd through `eigh`:

        S = (x)K_m + diag(tau),  A = (x)lambda_m + tau,  g = S^-1 y  (the tensor the reference caches as `self.g`)
        dNLL/dY    = g
        dNLL/dA    = 1/2 (1/A - (T_1/A)^2)   elementwise in the eigenbasis, where tau is added (summed for a scalar tau)
        dNLL/dK_m  = 1/2 U_m diag(c_m) U_m^T  -  1/2 g_(m) [g x_{m' != m} K_m']_(m)^T,
                     c_m[i] = sum_{others} (prod_{m' != m} lambda_m') / A

    No eigenvector derivatives (the 1 / (lambda_i - lambda_j) terms of eigh's backward, ill-defined for the clustered
    spectra kernel matrices have) appear; every O(N^2 prod d) / O(N^3) product runs on the fp64 matrix-core GEMM.
    That form holds for the reference's scalar tau = 1/noise (+ scalar y_var).  An elementwise y_var is added to A in
    the eigenbasis (MFGP_ver2023May/base_gp/hogp.py:118), which makes S depend on the eigenvectors themselves; that
    case takes the general eigh pullback, written out in `backward`.
    """

    @staticmethod
    def forward(ctx, cache, y, tau, *Ks):
        es = [eigen_pairs(K) for K in Ks]
        A = _outer([e.value for e in es]) + tau
        Ainv = A.reciprocal()
        T_1 = multi_mode_dot(y, [e.vector.T.contiguous() for e in es])
        W = T_1 * Ainv
        g = multi_mode_dot(W, [e.vector for e in es])
        nd = A.numel()
        nll = 0.5 * nd * math.log(2 * math.pi) + 0.5 * torch.log(A).sum() + 0.5 * (T_1 * W).sum()
        cache.update(eigen=es, A=A, g=g)
        ctx.pack = (es, Ainv, W, g, Ks, tau.shape, nd)
        return nll / nd

    @staticmethod
    def backward(ctx, dl):
        es, Ainv, W, g, Ks, tau_shape, nd = ctx.pack
        s = dl / nd
        need = ctx.needs_input_grad
        dY = g * s if need[1] else None
        dtau = None
        if need[2]:
            # tau sits on A (the eigenbasis diagonal): d/dA [1/2 log A + 1/2 T_1^2 / A] = 1/2 (1/A - W^2)
            dtau = (0.5 * s * (Ainv - W * W)).sum_to_size(tau_shape)
        n = len(Ks)
        dKs = [None] * n
        uniform = math.prod(tau_shape) == 1
        if any(need[3:]):
            P0 = mode_dot(g, Ks[0], 0) if uniform and any(need[4:]) else None   # the one N^2 prod(d) product the modes m >= 1 share
            for m in range(n):
                if not need[3 + m]:
                    continue
                t = Ainv if uniform else Ainv - W * W
                for mp in range(n):
                    if mp != m:
                        shape = [1] * n
                        shape[mp] = -1
                        t = t * es[mp].value.reshape(shape)
                c = t.sum(dim=[a for a in range(n) if a != m])
                U = es[m].vector
                if uniform:
                    term1 = F.matmul_nt(U * c.unsqueeze(0), U)
                    P = g if m == 0 else P0
                    for mp in range(1, n):
                        if mp != m:
                            P = mode_dot(P, Ks[mp], mp)
                    gm = g.movedim(m, 0).reshape(g.shape[m], -1)
                    Pm = P.movedim(m, 0).reshape(g.shape[m], -1)
                    dKs[m] = (0.5 * s) * (term1 - F.matmul_nt(gm, Pm))
                else:
                    # an elementwise tau lives in the eigenbasis, so S depends on the eigenvectors themselves: the
                    # general eigh pullback U (diag(dlambda) + (U^T dU) / (lambda_j - lambda_i)) U^T with U^T dU = T_1(m) W(m)^T
                    lam = es[m].value
                    T1m = (W / Ainv).movedim(m, 0).reshape(g.shape[m], -1)
                    Wm = W.movedim(m, 0).reshape(g.shape[m], -1)
                    E = lam.unsqueeze(0) - lam.unsqueeze(1)
                    E.diagonal().fill_(float("inf"))
                    X = F.matmul_nt(T1m, Wm) / E
                    X.diagonal().add_(0.5 * c)
                    dKs[m] = s * F.matmul_nt(F.matmul_nt(U, X.T.contiguous()), U)
        return (None, dY, dtau) + tuple(dKs)


def kron_nll(y, tau, Ks):
    """(loss, cache): loss = +NLL / numel; cache holds the (detached) `eigen` pairs, `A` and `g` of this evaluation"""
    cache = {}
    loss = _KronNLL.apply(cache, y, tau, *Ks)
    return loss, cache


class HOGP_simple(nn.Module):
    def __init__(self, kernel, noise_variance, output_shape, learnable_grid=False, learnable_map=False,
                 variance_mode="explicit_inverse"):
        super().__init__()
        if variance_mode == "reference":          # (the mode's name until round 4: it never was the reference's LU inverse)
            variance_mode = "explicit_inverse"
        if variance_mode not in ("explicit_inverse", "eigen"):
            raise ValueError("variance_mode must be 'explicit_inverse' or 'eigen', got %r" % (variance_mode,))
        self.variance_mode = variance_mode
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
        self._K_inv = None            # (reference variance mode: the explicit inverse is formed once per likelihood call)
        self.K.append(self._kernel(0, x_train, x_train))
        for i in range(len(self.kernel_list) - 1):
            _in = mode_dot(self.grid[i].to(device=dev, dtype=torch.float64), self.mapping_vector[i].to(device=dev, dtype=torch.float64), 0)
            self.K.append(self._kernel(i + 1, _in, _in))
        loss, cache = kron_nll(y, self.noise_variance.to(dev).pow(-1), self.K)
        self.K_eigen[:] = cache["eigen"]
        self.A = cache["A"]
        self.g = cache["g"]
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
        if self.variance_mode in ("explicit_inverse", "reference"):   # K_star @ K_x^-1 @ U_x in the reference's order of operations (:68)
            if torch.is_grad_enabled() and self.K[0].requires_grad:
                K_inv = _InvFromEigen.apply(self.K[0], e0.value.detach(), e0.vector.detach())
            else:
                if getattr(self, "_K_inv", None) is None:
                    with torch.no_grad():
                        self._K_inv = _InvFromEigen.apply(self.K[0].detach(), e0.value.detach(), e0.vector.detach())
                K_inv = self._K_inv
            ev_x = F.matmul_nt(F.matmul_nt(K_star, K_inv.T.contiguous()), e0.vector.T.contiguous()).pow(2)
        else:                                   # the same matrix from the cached eigenpairs: K_x^-1 U_x = U_x / lambda
            ev_x = mode_dot(K_star, (e0.vector / e0.value.unsqueeze(0)).T.contiguous(), 1).pow(2)
        evs = [ev_x] + [self.K_eigen[i + 1].vector.pow(2) for i in range(n_dim)]
        var_diag = diag_K + multi_mode_dot(S_2, evs)
        odt = x_test.dtype if x_test.dtype.is_floating_point else torch.float64
        return predict_u.to(device=x_test.device, dtype=odt), var_diag.to(device=x_test.device, dtype=odt)
