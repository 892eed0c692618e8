"""Differentiable building blocks on the library's kernels: covariance matrices (single kernels and composed trees), the Cholesky
factor with passenger rows, the conditional Gaussian and the likelihood of a caller-built covariance (the reference's
gp_computation_pack.py:59-110 formulas), fp64 matrix-core GEMMs, subset matching and the small eigensolver.
"""
import ctypes as C
import math

import torch

from . import _lib
from ._common import NEG_INF, _check_same_D, _check_xy, _dev, _device_of, _ptr, _raise_not_pd, _split_kfun, _weights
from ._lib import FFGP_LL_V1, FFGP_LL_V2, FFGP_VAR_DIAG, FFGP_VAR_FULL, PI_TRUNC, Grads, KDesc, KDescGrads, Problem, check, lib
from .kdesc import FFGP_KFUN_LINEAR, FFGP_KOP_PRODUCT, FFGP_KOP_SUM, FFGP_TREE_BALANCED, FFGP_TREE_CHAIN, _PAIR_KEYS, _pair_descs, _pair_grad_buffers, _pair_grads_out, _pair_split, _tree_spec


class _KernelMatrix(torch.autograd.Function):
    """K(x1, x2) [n1, n2] (no Sigma extras); backward gives d/d{w, amp} for a dense upstream dK (ffgp_kernel_grad)."""

    @staticmethod
    def forward(ctx, x1, x2, w, amp, clamp, kfun, kparam=None):
        dev = _device_of(x1, x2, w, amp)
        if kparam is not None:
            kfun = (kfun[0], float(kparam.detach()))
        h = _lib.handle(dev.index)
        _lib.bind_stream(h, dev.index)
        a, b = _dev(x1, dev), _dev(x2, dev)
        if a.dim() > 2:  # SE_kernel.py:29-32 flattens >2-D inputs
            a, b = a.reshape(a.shape[0], -1), b.reshape(b.shape[0], -1)
        _check_xy(a, what="x1")
        _check_same_D(a, b, "x2")
        D = a.shape[1]
        wd = _weights(w, D, dev)
        ad = _dev(amp.reshape(-1)[:1], dev)
        K = torch.empty((a.shape[0], b.shape[0]), dtype=torch.float64, device=dev)
        check(lib.ffgp_assemble(h, _ptr(a), a.shape[0], _ptr(b), b.shape[0], D, _ptr(wd), _ptr(ad), clamp, None, None, 0, None,
                                0, 0.0, 0.0, _ptr(K), b.shape[0], 0, int(kfun[0]), float(kfun[1])), "ffgp_assemble")
        ctx.saved = (a, b, wd, ad, clamp, kfun, dev)
        ctx.meta = [(t.shape, t.dtype, t.device) for t in (w, amp)]
        ctx.xmeta = [(t.shape, t.dtype, t.device) for t in (x1, x2)]
        ctx.kp_meta = (kparam.shape, kparam.dtype, kparam.device) if kparam is not None and kparam.requires_grad else None
        odt = x1.dtype if x1.dtype.is_floating_point else torch.float64
        ctx.out = (x1.device, odt)
        return K.to(device=x1.device, dtype=odt)

    @staticmethod
    def backward(ctx, dK):
        a, b, wd, ad, clamp, kfun, dev = ctx.saved
        h = _lib.handle(dev.index)
        _lib.bind_stream(h, dev.index)
        dKd = _dev(dK, dev)
        D = a.shape[1]
        g_w = torch.empty((D,), dtype=torch.float64, device=dev)
        g_amp = torch.empty((1,), dtype=torch.float64, device=dev)
        g_kp = torch.empty((1,), dtype=torch.float64, device=dev) if ctx.kp_meta is not None else None
        check(lib.ffgp_kernel_grad(h, _ptr(a), a.shape[0], _ptr(b), b.shape[0], D, _ptr(wd), _ptr(ad), clamp, int(kfun[0]),
                                   float(kfun[1]), _ptr(dKd), dKd.shape[1], _ptr(g_w), _ptr(g_amp), _ptr(g_kp)),
              "ffgp_kernel_grad")
        (ws, wdt, wdev), (as_, adt, adev) = ctx.meta
        if math.prod(ws) == 1 and D > 1:
            g_w = g_w.sum().reshape(1)
        if g_kp is not None:
            ks, kdt, kdev = ctx.kp_meta
            g_kp = g_kp.reshape(ks).to(device=kdev, dtype=kdt)
        gx1 = gx2 = None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            # input gradients (acquisition functions differentiate the posterior w.r.t. the test points):
            # Wt = dK o amp o (-2 phi') from the library, then two thin products with a ones column riding along
            n1, n2 = a.shape[0], b.shape[0]
            Wt = torch.empty((n1, n2), dtype=torch.float64, device=dev)
            check(lib.ffgp_kernel_input_weights(h, _ptr(a), n1, _ptr(b), n2, D, _ptr(wd), _ptr(ad), clamp, int(kfun[0]),
                                                float(kfun[1]), _ptr(dKd), dKd.shape[1], _ptr(Wt), n2),
                  "ffgp_kernel_input_weights")
            w2 = (wd * wd).reshape(1, D)
            one = lambda t: torch.cat([t, torch.ones((t.shape[0], 1), dtype=torch.float64, device=dev)], 1)
            if ctx.needs_input_grad[0]:
                P = _gemm(dev, 0, 1, Wt, one(b), n1, D + 1, n2, 1.0)          # [Wt X2 | rowsum(Wt)]
                gx1 = -w2 * (P[:, D:] * a - P[:, :D])
                shp, dt, dv = ctx.xmeta[0]
                gx1 = gx1.reshape(shp).to(device=dv, dtype=dt)
            if ctx.needs_input_grad[1]:
                P = _gemm(dev, 1, 1, Wt, one(a), n2, D + 1, n1, 1.0)          # [Wt^T X1 | colsum(Wt)]
                gx2 = w2 * (P[:, :D] - P[:, D:] * b)
                shp, dt, dv = ctx.xmeta[1]
                gx2 = gx2.reshape(shp).to(device=dv, dtype=dt)
        return (gx1, gx2, g_w.reshape(ws).to(device=wdev, dtype=wdt), g_amp.reshape(as_).to(device=adev, dtype=adt),
                None, None, g_kp)


def kernel_matrix(x1, x2, w, amp, clamp=NEG_INF, kfun=(0, 1.0)):
    """K(x1, x2) [n1, n2] on the device (no Sigma extras); differentiable w.r.t. the inputs x1 / x2, w, amp and a
    tensor profile parameter."""
    kfun, kparam = _split_kfun(kfun)
    return _KernelMatrix.apply(x1, x2, w, amp, clamp, kfun, kparam)


class _KernelPair(torch.autograd.Function):
    """K = the composed kernel of x1, x2 [n1, n2]; backward: every leaf's w / amp / kparam / center from one read of dK, and --
    when x1 / x2 carry gradients -- every leaf's input-weight matrix from a second pass (ffgp_kernel_input_weights_tree) followed
    by two thin matrix-core products per leaf."""

    @staticmethod
    def forward(ctx, x1, x2, op, meta, *tensors):
        dev = _device_of(x1, x2, tensors[0])
        h = _lib.handle(dev.index)
        _lib.bind_stream(h, dev.index)
        a, b = _dev(x1, dev), _dev(x2, dev)
        _check_xy(a, what="x1")
        _check_same_D(a, b, "x2")
        D = a.shape[1]
        keep = []
        tree = _pair_descs(dev, D, meta, tensors, keep, op)
        K = torch.empty((a.shape[0], b.shape[0]), dtype=torch.float64, device=dev)
        check(lib.ffgp_assemble_tree(h, _ptr(a), a.shape[0], _ptr(b), b.shape[0], D, C.byref(tree), None, None, 0, None, 0, 0.0, 0.0,
                                     _ptr(K), b.shape[0], 0), "ffgp_assemble_tree")
        ctx.saved = (a, b, tree, keep, dev, meta)
        ctx.staged = keep[0]
        ctx.metas = [(t.shape, t.dtype, t.device) if isinstance(t, torch.Tensor) else None for t in tensors]
        ctx.xmeta = [(t.shape, t.dtype, t.device) for t in (x1, x2)]
        odt = x1.dtype if x1.dtype.is_floating_point else torch.float64
        return K.to(device=x1.device, dtype=odt)

    @staticmethod
    def backward(ctx, dK):
        a, b, tree, keep, dev, meta = ctx.saved
        staged = ctx.staged
        h = _lib.handle(dev.index)
        _lib.bind_stream(h, dev.index)
        dKd = _dev(dK, dev)
        D = a.shape[1]
        nl = len(meta)
        n1, n2 = a.shape[0], b.shape[0]
        needs = [bool(f) for f in ctx.needs_input_grad[4:4 + 4 * nl]]
        garr, bufs = _pair_grad_buffers(dev, D, needs)
        if garr is not None:
            check(lib.ffgp_kernel_grad_tree(h, _ptr(a), n1, _ptr(b), n2, D, C.byref(tree), _ptr(dKd), dKd.shape[1], garr),
                  "ffgp_kernel_grad_tree")
        gx1 = gx2 = None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            Wt = torch.empty((nl, n1, n2), dtype=torch.float64, device=dev)
            check(lib.ffgp_kernel_input_weights_tree(h, _ptr(a), n1, _ptr(b), n2, D, C.byref(tree), _ptr(dKd), dKd.shape[1], _ptr(Wt),
                                                     n2, n1 * n2), "ffgp_kernel_input_weights_tree")
            one = lambda t: torch.cat([t, torch.ones((t.shape[0], 1), dtype=torch.float64, device=dev)], 1)
            for e in range(nl):
                wd, _, cen = staged[e]
                w2 = (wd * wd).reshape(1, D)
                if meta[e][0] == FFGP_KFUN_LINEAR:
                    cen = cen.reshape(1, D) if cen is not None else None
                    if ctx.needs_input_grad[0]:      # dK/dp = amp w^2 (q - c)
                        t = w2 * _gemm(dev, 0, 1, Wt[e], (b - cen) if cen is not None else b, n1, D, n2, 1.0)
                        gx1 = t if gx1 is None else gx1 + t
                    if ctx.needs_input_grad[1]:
                        t = w2 * _gemm(dev, 1, 1, Wt[e], (a - cen) if cen is not None else a, n2, D, n1, 1.0)
                        gx2 = t if gx2 is None else gx2 + t
                else:
                    if ctx.needs_input_grad[0]:
                        P = _gemm(dev, 0, 1, Wt[e], one(b), n1, D + 1, n2, 1.0)          # [Wt X2 | rowsum(Wt)]
                        t = -w2 * (P[:, D:] * a - P[:, :D])
                        gx1 = t if gx1 is None else gx1 + t
                    if ctx.needs_input_grad[1]:
                        P = _gemm(dev, 1, 1, Wt[e], one(a), n2, D + 1, n1, 1.0)          # [Wt^T X1 | colsum(Wt)]
                        t = w2 * (P[:, :D] - P[:, D:] * b)
                        gx2 = t if gx2 is None else gx2 + t
            if gx1 is not None:
                shp, dt, dv = ctx.xmeta[0]
                gx1 = gx1.reshape(shp).to(device=dv, dtype=dt)
            if gx2 is not None:
                shp, dt, dv = ctx.xmeta[1]
                gx2 = gx2.reshape(shp).to(device=dv, dtype=dt)
        return (gx1, gx2, None, None) + tuple(_pair_grads_out(bufs, D, needs, ctx.metas))


def kernel_pair(x1, x2, descs, op):
    """The composed kernel on the device from descriptor dicts {kfun, w, amp, clamp, kparam, center} in one pass.
    op: FFGP_KOP_* for two descriptors, or (shape, ops) for a nested composition of three / four (see `_tree_spec`)."""
    meta, tensors = _pair_split(descs)
    return _KernelPair.apply(x1, x2, op, meta, *tensors)


def kernel_on_device(kernel, x1, x2):
    """kernel(x1, x2) as a differentiable fp64 tensor resident on the compute device -- the entry of the composed
    path for kernels without a fused (w, amp, profile) descriptor (SumKernel, ProductKernel, LinearKernel, user
    modules).  Kernels of this package are evaluated on device-resident inputs; anything else is called as the
    caller wrote it and its result moved."""
    dev = _device_of(x1, x2)
    if getattr(kernel, "_ffgp_device_aware", False):
        x1 = x1.to(device=dev, dtype=torch.float64)
        x2 = x2.to(device=dev, dtype=torch.float64)
    return kernel(x1, x2).to(device=dev, dtype=torch.float64)


def add_diagonal(K, *terms):
    """K + sum(terms) * I without an N x N identity (differentiable; terms are scalars / [1] tensors / [N] vectors)."""
    S = K.clone()
    dg = S.diagonal()
    for t in terms:
        if t is None:
            continue
        dg.add_(t.to(device=K.device, dtype=K.dtype).reshape(-1) if isinstance(t, torch.Tensor) else t)
    return S


def _pad_ld(n):
    return (n + 1) // 2 * 2


@torch.no_grad()
def cholesky_with_rows(Sigma, rows=None):
    """Lower factor of Sigma [n, n]; if `rows` [m, n] is given also returns rows @ L^-T (= (L^-1 rows^T)^T),
    computed inside the factorisation (ffgp_potrf_rows)."""
    dev = _device_of(Sigma, rows)
    h = _lib.handle(dev.index)
    _lib.bind_stream(h, dev.index)
    n = Sigma.shape[0]
    m = 0 if rows is None else rows.shape[0]
    ld = _pad_ld(n)
    W = torch.zeros((n + m, ld), dtype=torch.float64, device=dev)
    W[:n, :n] = _dev(Sigma, dev)
    if m:
        W[n:, :n] = _dev(rows, dev)
    rc = check(lib.ffgp_potrf_rows(h, _ptr(W), n, n + m, ld), "ffgp_potrf_rows")
    if rc > 0:
        _raise_not_pd(rc, "linalg.cholesky")
    L = torch.tril(W[:n, :n])
    return (L, W[n:, :n]) if m else (L, None)


def cholesky(Sigma):
    """Drop-in for torch.linalg.cholesky on the GP path (lower factor, raises LinAlgError if not PD)."""
    L, _ = cholesky_with_rows(Sigma)
    return L.to(device=Sigma.device, dtype=Sigma.dtype)


class _CondGauss(torch.autograd.Function):
    """mu = K_s^T Sigma^-1 y, cov = K_ss - K_s^T Sigma^-1 K_s (gp_computation_pack.py:103-110); y^T and K_s^T ride as
    passenger rows of ONE factorisation.  Backward (closed form; B = Sigma^-1 K_s, alpha = Sigma^-1 y come from one
    L^T solve on the saved factor, everything else is GEMMs):
        dK_s = alpha Gmu^T - B (Gc + Gc^T)      dK_ss = Gc      dy = B Gmu
        dSigma = -1/2 (dy alpha^T + alpha dy^T) + 1/2 B (Gc + Gc^T) B^T          (symmetric, as torch's cholesky backward)"""

    @staticmethod
    def forward(ctx, y, Sigma, K_s, K_ss, factor=None):
        dev = _device_of(y, Sigma, K_s, K_ss)
        yd, Ksd = _dev(y, dev), _dev(K_s, dev)
        d = yd.shape[1]
        if factor is None:
            L, R = cholesky_with_rows(Sigma, torch.cat([yd.T, Ksd.T], 0))
            Gt, Vt = R[:d].contiguous(), R[d:].contiguous()      # Gamma^T [d, n], V^T [nt, n]
        else:
            # `factor`: a Posterior that already holds chol(Sigma) and Gamma = L^-1 y for exactly this (y, Sigma) -- the caller
            # vouches for that (cigp's cache is keyed on the tensors and their versions).  Sigma's VALUES are not read; it
            # stays an input so that its gradient reaches the hyper-parameters.  One TRSM sweep instead of N^3 / 3.
            n = factor.n
            L = factor.W[:n]
            V = Ksd.clone()
            check(lib.ffgp_trsm_lower(factor._h(), _ptr(factor.W), n, factor.ld, _ptr(V), V.shape[1], V.shape[1]), "ffgp_trsm_lower")
            Vt = V.T.contiguous()
            Gt = factor.Gamma.T.contiguous()
        mu = _gemm(dev, 0, 0, Vt, Gt, Vt.shape[0], d, Vt.shape[1], 1.0)
        cov = _dev(K_ss, dev) - _gemm(dev, 0, 0, Vt, Vt, Vt.shape[0], Vt.shape[0], Vt.shape[1], 1.0)
        ctx.saved = (L, Gt, Vt, dev)
        ctx.meta = [(t.shape, t.dtype, t.device) for t in (y, Sigma, K_s, K_ss)]
        odt = y.dtype if y.dtype.is_floating_point else torch.float64
        return mu.to(device=y.device, dtype=odt), cov.to(device=K_ss.device, dtype=K_ss.dtype)

    @staticmethod
    def backward(ctx, Gmu, Gc):
        L, Gt, Vt, dev = ctx.saved
        n, d, nt = L.shape[0], Gt.shape[0], Vt.shape[0]
        h = _lib.handle(dev.index)
        _lib.bind_stream(h, dev.index)
        X = torch.cat([Gt, Vt], 0).T.contiguous()            # [n, d + nt]  ->  [alpha | B] = L^-T [Gamma | V]
        check(lib.ffgp_trsm_lower_t(h, _ptr(L), n, L.stride(0), _ptr(X), d + nt, X.stride(0)), "ffgp_trsm_lower_t")
        alpha, B = X[:, :d].contiguous(), X[:, d:].contiguous()
        Gmu = torch.zeros((nt, d), dtype=torch.float64, device=dev) if Gmu is None else _dev(Gmu, dev)
        Gc = torch.zeros((nt, nt), dtype=torch.float64, device=dev) if Gc is None else _dev(Gc, dev)
        Gs = (Gc + Gc.T).contiguous()
        BGs = _gemm(dev, 0, 0, B, Gs, n, nt, nt, 1.0)         # B Gs   (Gs symmetric: NT form is fine)
        out = [None, None, None, None]
        dy = _gemm(dev, 0, 1, B, Gmu, n, d, nt, 1.0)          # B Gmu
        if ctx.needs_input_grad[0]:
            out[0] = dy
        if ctx.needs_input_grad[1]:
            T1 = _gemm(dev, 0, 0, dy, alpha, n, n, d, 1.0)    # dy alpha^T
            out[1] = -0.5 * (T1 + T1.T) + _gemm(dev, 0, 0, BGs, B, n, n, nt, 0.5)
        if ctx.needs_input_grad[2]:
            out[2] = _gemm(dev, 0, 0, alpha, Gmu, n, nt, d, 1.0) - BGs
        if ctx.needs_input_grad[3]:
            out[3] = Gc
        return tuple(None if t is None else t.reshape(m[0]).to(device=m[2], dtype=m[1]) for t, m in zip(out, ctx.meta)) + (None,)


def conditional_gaussian(y, Sigma, K_s, K_ss, factor=None):
    return _CondGauss.apply(y, Sigma, K_s, K_ss, factor)


class _GaussNLLFromCov(torch.autograd.Function):
    """value(Y, cov) for a caller-built covariance (V1: +nll, V2: -LL of the Sigma^-2 form); backward returns
    d/dY and the symmetric d/d(cov) -- what torch's cholesky backward gives the reference."""

    @staticmethod
    def forward(ctx, Y, cov, variant, pi_const, rec=True):
        dev = _device_of(Y, cov)
        h = _lib.handle(dev.index)
        _lib.bind_stream(h, dev.index)
        Yd, Cd = _dev(Y, dev), _dev(cov, dev)
        n, d = Yd.shape
        p = Problem()
        p.n, p.D, p.d = n, 0, d
        p.Y_dev, p.cov_dev, p.ld_cov = _ptr(Yd), _ptr(Cd), Cd.shape[1]
        p.ll_variant, p.pi_const = variant, pi_const
        out = torch.empty((), dtype=torch.float64, device=dev)
        g = None
        ctx.grads = {}
        if rec and (ctx.needs_input_grad[0] or ctx.needs_input_grad[1]):   # rec: the caller's grad mode (off inside forward)
            g = Grads()
            ctx.grads["Y"] = torch.empty((n, d), dtype=torch.float64, device=dev)
            ctx.grads["cov"] = torch.empty((n, n), dtype=torch.float64, device=dev)
            g.g_Y_dev, g.g_cov_dev, g.ld_gcov = _ptr(ctx.grads["Y"]), _ptr(ctx.grads["cov"]), n
        rc = check(lib.ffgp_nlml_fused(h, C.byref(p), _ptr(out), C.byref(g) if g is not None else None), "ffgp_nlml_fused")
        if rc > 0:
            _raise_not_pd(rc, "linalg.cholesky")
        ctx.meta = [(t.shape, t.dtype, t.device) for t in (Y, cov)]
        return out.to(device=Y.device, dtype=Y.dtype if Y.dtype.is_floating_point else torch.float64)

    @staticmethod
    def backward(ctx, gout):
        outs = []
        for i, (key, (shape, dtype, device)) in enumerate(zip(("Y", "cov"), ctx.meta)):
            if not ctx.needs_input_grad[i]:
                outs.append(None)
                continue
            t = ctx.grads[key] * gout.to(device=ctx.grads[key].device, dtype=torch.float64)
            outs.append(t.reshape(shape).to(device=device, dtype=dtype))
        return outs[0], outs[1], None, None, None


def gaussian_nll_from_cov(Y, cov, variant=FFGP_LL_V2, pi_const=math.pi):
    return _GaussNLLFromCov.apply(Y, cov, variant, pi_const, torch.is_grad_enabled())


def gaussian_ll_v2(Y, cov):
    """-LL of the reference's 'cholesky3' Gaussian_log_likelihood (Sigma^-2 quadratic form), from a given cov;
    differentiable w.r.t. Y and cov."""
    return gaussian_nll_from_cov(Y, cov, FFGP_LL_V2, math.pi)


def _gemm(dev, opa, opb, A, B, m, n, k, alpha):
    h = _lib.handle(dev.index)
    _lib.bind_stream(h, dev.index)
    out = torch.empty((m, n), dtype=torch.float64, device=dev)
    if m and n:
        check(lib.ffgp_gemm(h, opa, opb, 0, 0, _ptr(A), A.stride(0), _ptr(B), B.stride(0), _ptr(out), n, m, n, k, float(alpha),
                            0.0), "ffgp_gemm")
    return out


class _MatmulNT(torch.autograd.Function):
    """alpha * A @ B^T on the fp64 matrix-core GEMM; the two backward products run on the same kernel."""

    @staticmethod
    def forward(ctx, A, B, alpha):
        dev = _device_of(A, B)
        a, b = _dev(A, dev), _dev(B, dev)
        ctx.saved = (a, b, alpha, dev)
        ctx.meta = [(t.dtype, t.device) for t in (A, B)]
        return _gemm(dev, 0, 0, a, b, a.shape[0], b.shape[0], a.shape[1], alpha)

    @staticmethod
    def backward(ctx, dC):
        a, b, alpha, dev = ctx.saved
        dc = _dev(dC, dev)
        (adt, adev), (bdt, bdev) = ctx.meta
        dA = dB = None
        if ctx.needs_input_grad[0]:   # dA = alpha dC B        [m, n] x [n, k]
            dA = _gemm(dev, 0, 1, dc, b, a.shape[0], a.shape[1], b.shape[0], alpha).to(device=adev, dtype=adt)
        if ctx.needs_input_grad[1]:   # dB = alpha dC^T A      [n, m] x [m, k]
            dB = _gemm(dev, 1, 1, dc, a, b.shape[0], b.shape[1], a.shape[0], alpha).to(device=bdev, dtype=bdt)
        return dA, dB, None


def matmul_nt(A, B, alpha=1.0):
    """alpha * A @ B^T for A [m, k], B [n, k] on the fp64 matrix-core GEMM (ffgp_gemm); fp64 result on the device."""
    return _MatmulNT.apply(A, B, alpha)


@torch.no_grad()
def rows_in(x1, x2):
    """Boolean mask [n1]: row i of x1 equals some row of x2 (exact IEEE ==, as the reference's broadcast comparison in
    MF_data.py:196-199) -- a device hash join (ffgp_rows_in)."""
    dev = _device_of(x1, x2)
    h = _lib.handle(dev.index)
    _lib.bind_stream(h, dev.index)
    a, b = _dev(x1, dev), _dev(x2, dev)
    D = int(math.prod(a.shape[1:])) if a.dim() > 1 else 1
    a, b = a.reshape(a.shape[0], D), b.reshape(b.shape[0], int(math.prod(b.shape[1:])) if b.dim() > 1 else 1)
    found = torch.zeros((a.shape[0],), dtype=torch.uint8, device=dev)
    if a.shape[0] and D > 0 and a.shape[1] == b.shape[1]:
        check(lib.ffgp_rows_in(h, _ptr(a), a.shape[0], _ptr(b), b.shape[0], a.shape[1], _ptr(found)), "ffgp_rows_in")
    return found.bool().to(x1.device)


@torch.no_grad()
def _syevj_small(M, descending=False):
    """batched hand-written Jacobi eigensolver for [B, n, n] (n <= 64) device tensors: (evals [B, n], Q [B, n, n])"""
    dev = M.device
    h = _lib.handle(dev.index)
    _lib.bind_stream(h, dev.index)
    B, n = M.shape[0], M.shape[-1]
    M = M.contiguous()
    Q = torch.empty((B, n, n), dtype=torch.float64, device=dev)
    ev = torch.empty((B, n), dtype=torch.float64, device=dev)
    check(lib.ffgp_syevj_small(h, _ptr(M), n, n, B, n * n, _ptr(Q), n, n * n, _ptr(ev), n, 1 if descending else 0),
          "ffgp_syevj_small")
    return ev, Q


class _EighSmall(torch.autograd.Function):
    """torch.linalg.eigh for one symmetric matrix with n <= 64 on the hand-written LDS Jacobi kernel (ffgp_syevj_small),
    with the standard backward  gK = sym( U (diag(g_lambda) + (U^T g_U) o E) U^T ),  E_ij = 1 / (lambda_j - lambda_i)."""

    @staticmethod
    def forward(ctx, K):
        dev = _device_of(K)
        ev, Q = _syevj_small(_dev(K, dev)[None])
        ctx.save_for_backward(ev[0], Q[0])
        ctx.meta = (K.dtype, K.device)
        return ev[0].to(device=K.device, dtype=K.dtype), Q[0].to(device=K.device, dtype=K.dtype)

    @staticmethod
    def backward(ctx, g_ev, g_Q):
        ev, U = ctx.saved_tensors
        dev = ev.device
        n = ev.shape[0]
        inner = torch.zeros((n, n), dtype=torch.float64, device=dev)
        if g_Q is not None:
            S = _gemm(dev, 1, 1, U, _dev(g_Q, dev), n, n, n, 1.0)                # U^T g_U
            diff = ev.unsqueeze(0) - ev.unsqueeze(1)                               # lambda_j - lambda_i
            E = torch.where(diff != 0, 1.0 / diff, torch.zeros_like(diff))
            inner = S * E
        if g_ev is not None:
            inner = inner + torch.diag(_dev(g_ev, dev))
        gK = _gemm(dev, 0, 0, _gemm(dev, 0, 1, U, inner.contiguous(), n, n, n, 1.0), U, n, n, n, 1.0)   # U inner U^T
        gK = 0.5 * (gK + gK.T)
        return gK.to(device=ctx.meta[1], dtype=ctx.meta[0])


def eigh_small(K):
    """(eigenvalues ascending [n], eigenvectors [n, n]) of a symmetric K with n <= 64, differentiable"""
    return _EighSmall.apply(K)
