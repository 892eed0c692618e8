"""The kept factor of a trained model: `Posterior` (factor once, query / differentiate / append afterwards) and the modules' cache of
it.  Reference: `cigp.forward` re-factorises on every call (GaussianProcess/cigp_v10.py:24-48); the acquisition loops of
Bayesian_optimization/acq.py:10-80 query a frozen model again and again.
"""
import ctypes as C
import math
import weakref

import torch

from . import _lib
from ._common import NEG_INF, _check_same_D, _check_xy, _dev, _device_of, _ptr, _raise_not_pd, _split_kfun, _weights
from ._lib import FFGP_LL_V1, FFGP_LL_V2, FFGP_VAR_DIAG, FFGP_VAR_FULL, PI_TRUNC, Grads, KDesc, KDescGrads, Problem, check, lib
from .kdesc import FFGP_KFUN_LINEAR, FFGP_KOP_PRODUCT, FFGP_KOP_SUM, FFGP_TREE_BALANCED, FFGP_TREE_CHAIN, _PAIR_KEYS, _pair_descs, _pair_grad_buffers, _pair_grads_out, _pair_split, _tree_spec
from .linalg import _gemm, _pad_ld, kernel_matrix, kernel_pair


class _PosteriorQuery(torch.autograd.Function):
    """mean = K_s^T alpha, var = K_ss - V^T V (V = L^-1 K_s) on a CACHED factor, differentiable w.r.t. K_s and K_ss only
    (the factor, alpha and the hyper-parameters are constants of a `Posterior`): what an acquisition optimiser needs to
    move its query points (Bayesian_optimization/acq.py:50-62) -- one TRSM sweep forward, one backward, no
    refactorisation.   dK_s = alpha Gm^T - Sigma^-1 K_s (Gv + Gv^T)   [diag mode: - 2 Sigma^-1 K_s diag(gv)],  dK_ss = Gv."""

    @staticmethod
    def forward(ctx, post, Ks, Kss, full_cov):
        dev, n = post.dev, post.n
        nt = Ks.shape[1]
        if post.alpha is None:
            post._solve_alpha()
        Ksd = _dev(Ks, dev)
        mean = _gemm(dev, 1, 1, Ksd, post.alpha, nt, post.d, n, 1.0)
        V = Ksd.clone()
        check(lib.ffgp_trsm_lower(post._h(), _ptr(post.W), n, post.ld, _ptr(V), nt, nt), "ffgp_trsm_lower")
        if full_cov:
            var = _dev(Kss, dev) - _gemm(dev, 1, 1, V, V, nt, nt, n, 1.0)
        else:
            var = _dev(Kss, dev) - (V * V).sum(0)
        ctx.pack = (post, V, n, full_cov, post.alpha)
        return mean, var

    @staticmethod
    def backward(ctx, Gm, Gv):
        post, V, n, full_cov, alpha = ctx.pack
        if post.n != n:
            raise RuntimeError("Posterior.append() was called between a differentiable query and its backward()")
        dev = post.dev
        nt = V.shape[1]
        dKs = torch.zeros_like(V)
        if Gm is not None:
            dKs = _gemm(dev, 0, 0, alpha, _dev(Gm, dev), n, nt, post.d, 1.0)          # alpha Gm^T
        dKss = None
        if Gv is not None:
            B = V.clone()
            check(lib.ffgp_trsm_lower_t(post._h(), _ptr(post.W), n, post.ld, _ptr(B), nt, nt), "ffgp_trsm_lower_t")   # Sigma^-1 K_s
            g = _dev(Gv, dev)
            if full_cov:
                dKs = dKs - _gemm(dev, 0, 0, B, (g + g.T).contiguous(), n, nt, nt, 1.0)
            else:
                dKs = dKs - 2.0 * B * g.unsqueeze(0)
            dKss = g
        return None, dKs, dKss, None


class Posterior:
    """A factored GP block kept on the device: factor once, query many times, append points without refactorising
    (the reference's `cigp.forward` re-runs `torch.linalg.cholesky` on every call,
    cigp_v10.py:31-35 -- inside an acquisition loop or when serving predictions that is N^3/3 per query for a factor
    that has not changed).

        predict(Xs)      assembly of K_s, one TRSM sweep on the cached factor (N^2 nt), two thin GEMMs
        append(X, Y)     L21 = (L^-1 K_nk)^T, L22 = chol(S_kk - L21 L21^T): O(N^2 k) instead of O(N^3 / 3)

    Parameters are the library's effective ones (w, amp, diag_add, clamp, kfun), frozen at construction -- or, for a composed
    kernel (SumKernel / ProductKernel over library kernels, `kernel._Pair.pair()`), `tree = (descriptors, operator spec)`."""

    def __init__(self, X, Y, w, amp, diag_add, clamp=NEG_INF, kfun=(0, 1.0), capacity=None, first_query=None,
                 var_add_all=0.0, tree=None):
        """first_query (optional [nt, D]): its K_s^T rides, with Y^T, as passenger rows of the factorisation itself, so
        the first answer (`self.first` = (mean, covariance)) costs what the fused one-shot posterior costs; the rows
        below the factor are scratch afterwards (later appends overwrite them)."""
        dev = _device_of(X, Y, w if tree is None else tree[0][0]["w"])
        self.dev = dev
        self.kfun, _ = _split_kfun(kfun)
        self.clamp = clamp
        Xd, Yd = _dev(X, dev), _dev(Y, dev)
        _check_xy(Xd, Yd)
        n, D = Xd.shape
        d = Yd.shape[1]
        self.tree = None
        if tree is not None:
            # frozen copies of the leaves' effective quantities on the device; the ctypes tree lives as long as this object
            descs = [{k: (_dev(v.detach(), dev).clone() if isinstance(v, torch.Tensor) else v) for k, v in dsc.items()} for dsc in tree[0]]
            meta, tensors = _pair_split(descs)
            self._tree_keep = []
            self.tree = (descs, tree[1], _pair_descs(dev, D, meta, tensors, self._tree_keep, tree[1]))
            self.w = self.amp = None
        else:
            self.w = _weights(w, D, dev)
            self.amp = _dev(amp.reshape(-1)[:1], dev)
        self.dadd = _dev(diag_add.reshape(-1)[:1], dev)
        Xq = _dev(first_query, dev) if first_query is not None else None
        if Xq is not None:
            _check_same_D(Xd, Xq)
        nt = Xq.shape[0] if Xq is not None else 0
        self.cap = max(int(capacity or 0), n)
        self.ld = _pad_ld(self.cap)
        rows = max(self.cap, n + d + nt)                      # room for the passenger rows of the first factorisation
        self.W = torch.zeros((rows, self.ld), dtype=torch.float64, device=dev)
        self.X = torch.empty((self.cap, D), dtype=torch.float64, device=dev)
        self.X[:n] = Xd
        self.n, self.D, self.d = n, D, d
        h = self._h()
        self._assemble(Xd, Xd, self.W, self.ld, lower=1, diag=True)
        self.W[n:n + d, :n] = Yd.T
        if nt:
            self._assemble(Xq, Xd, self.W[n + d:], self.ld, lower=0, diag=False)          # K_s^T [nt, n]
        rc = check(lib.ffgp_potrf_rows(h, _ptr(self.W), n, n + d + nt, self.ld), "ffgp_potrf_rows")
        if rc > 0:
            _raise_not_pd(rc, "linalg.cholesky")
        Gt = self.W[n:n + d, :n].contiguous()                 # Gamma^T
        self.Gamma = Gt.T.contiguous()
        self.alpha = None                                     # Sigma^-1 Y: solved when a later query needs it
        self.first = None
        if nt:
            Vt = self.W[n + d:n + d + nt, :n].contiguous()    # V^T = (L^-1 K_s)^T
            mean = _gemm(dev, 0, 0, Vt, Gt, nt, d, n, 1.0)
            var = torch.empty((nt, nt), dtype=torch.float64, device=dev)
            self._assemble(Xq, Xq, var, nt, lower=0, diag=False)
            self.first = (mean, var - _gemm(dev, 0, 0, Vt, Vt, nt, nt, n, 1.0) + var_add_all)

    def _h(self):
        h = _lib.handle(self.dev.index)
        _lib.bind_stream(h, self.dev.index)
        return h

    def _assemble(self, A, B, out, ld, lower, diag):
        if self.tree is not None:
            check(lib.ffgp_assemble_tree(self._h(), _ptr(A), A.shape[0], _ptr(B), B.shape[0], self.D, C.byref(self.tree[2]),
                                         _ptr(self.dadd) if diag else None, None, 0, None, 0, 0.0, 0.0, _ptr(out), ld, lower),
                  "ffgp_assemble_tree")
            return
        check(lib.ffgp_assemble(self._h(), _ptr(A), A.shape[0], _ptr(B), B.shape[0], self.D, _ptr(self.w), _ptr(self.amp),
                                self.clamp, _ptr(self.dadd) if diag else None, None, 0, None, 0, 0.0, 0.0, _ptr(out), ld, lower,
                                int(self.kfun[0]), float(self.kfun[1])), "ffgp_assemble")

    def _solve_alpha(self):
        self.alpha = self.Gamma.clone()
        check(lib.ffgp_trsm_lower_t(self._h(), _ptr(self.W), self.n, self.ld, _ptr(self.alpha), self.d, self.d),
              "ffgp_trsm_lower_t")

    @torch.no_grad()
    def predict(self, Xs, full_cov=True, var_add_all=0.0):
        """mean [nt, d], covariance [nt, nt] (or variance [nt]) at Xs; the noise convention is the caller's
        (`var_add_all` lands on every entry, cigp_v10.py:44)."""
        dev, n = self.dev, self.n
        Xsd = _dev(Xs, dev)
        _check_same_D(self.X, Xsd)
        nt = Xsd.shape[0]
        if self.alpha is None:
            self._solve_alpha()
        Ks = torch.empty((n, nt), dtype=torch.float64, device=dev)
        self._assemble(self.X[:n], Xsd, Ks, nt, lower=0, diag=False)
        mean = _gemm(dev, 1, 1, Ks, self.alpha, nt, self.d, n, 1.0)                 # K_s^T alpha
        check(lib.ffgp_trsm_lower(self._h(), _ptr(self.W), n, self.ld, _ptr(Ks), nt, nt), "ffgp_trsm_lower")   # V = L^-1 K_s
        if full_cov:
            var = torch.empty((nt, nt), dtype=torch.float64, device=dev)
            self._assemble(Xsd, Xsd, var, nt, lower=0, diag=False)
            var = var - _gemm(dev, 1, 1, Ks, Ks, nt, nt, n, 1.0) + var_add_all
        elif self.tree is not None:
            var = self._kernel(Xsd, Xsd).diagonal() - (Ks * Ks).sum(0) + var_add_all
        else:
            var = float(self.amp) - (Ks * Ks).sum(0) + var_add_all      # phi(0) = 1 for every radial profile
        return mean, var

    def _kernel(self, a, b):
        """the frozen kernel as a differentiable call (w.r.t. a, b)"""
        if self.tree is not None:
            return kernel_pair(a, b, self.tree[0], self.tree[1])
        return kernel_matrix(a, b, self.w, self.amp, self.clamp, kfun=self.kfun)

    def predict_diff(self, Xs, full_cov=True, var_add_all=0.0):
        """`predict` with autograd w.r.t. the query points: K_s and K_ss come from the differentiable kernel call, the
        solves run on the cached factor (`_PosteriorQuery`).  The hyper-parameters, X and Y are constants here -- use
        the model's own forward under autograd when their gradients are wanted as well."""
        dev, n = self.dev, self.n
        Xsd = Xs.to(device=dev, dtype=torch.float64)
        _check_same_D(self.X, Xsd)
        Ks = self._kernel(self.X[:n], Xsd)
        if full_cov:
            Kss = self._kernel(Xsd, Xsd)
        elif self.tree is not None:
            Kss = self._kernel(Xsd, Xsd).diagonal()
        else:
            Kss = self.amp.expand(Xsd.shape[0])                  # phi(0) = 1 for every radial profile
        mean, var = _PosteriorQuery.apply(self, Ks, Kss, full_cov)
        return mean, var + var_add_all

    @torch.no_grad()
    def append(self, X_new, Y_new):
        """Extend the factor by k points: the new block row of L is a TRSM on the cached factor, the new diagonal
        block a k x k Cholesky of the Schur complement."""
        dev, n, h = self.dev, self.n, self._h()
        Xn, Yn = _dev(X_new, dev), _dev(Y_new, dev)
        _check_same_D(self.X, Xn, "X_new")
        if Yn.dim() != 2 or Yn.shape != (Xn.shape[0], self.d):
            raise ValueError("Y_new must be [%d, %d], got shape %s" % (Xn.shape[0], self.d, tuple(Yn.shape)))
        k = Xn.shape[0]
        if n + k > self.cap or n + k > self.W.shape[0]:       # grow geometrically; the factor is copied once
            cap = max(n + k, 2 * self.cap)
            ld = _pad_ld(cap)
            W = torch.zeros((cap, ld), dtype=torch.float64, device=dev)
            W[:n, :n] = self.W[:n, :n]
            Xb = torch.empty((cap, self.D), dtype=torch.float64, device=dev)
            Xb[:n] = self.X[:n]
            self.W, self.X, self.cap, self.ld = W, Xb, cap, ld
        B = torch.empty((n, k), dtype=torch.float64, device=dev)
        self._assemble(self.X[:n], Xn, B, k, lower=0, diag=False)
        check(lib.ffgp_trsm_lower(h, _ptr(self.W), n, self.ld, _ptr(B), k, k), "ffgp_trsm_lower")        # L^-1 K_nk = L21^T
        ks = _pad_ld(k)
        S = torch.zeros((k, ks), dtype=torch.float64, device=dev)
        self._assemble(Xn, Xn, S, ks, lower=0, diag=True)
        S[:, :k] -= _gemm(dev, 1, 1, B, B, k, k, n, 1.0)                                                 # Schur complement
        # the small factor goes through a second handle: this handle's store of inverted diagonal blocks stays
        # attached to the big factor and is only extended by the new blocks
        h2 = _lib.handle(dev.index, 1)
        _lib.bind_stream(h2, dev.index)
        rc = check(lib.ffgp_potrf(h2, _ptr(S), k, ks), "ffgp_potrf")
        if rc > 0:
            _raise_not_pd(n + rc, "linalg.cholesky")
        G_new = Yn - _gemm(dev, 1, 1, B, self.Gamma, k, self.d, n, 1.0)                                  # y_new - L21 Gamma
        check(lib.ffgp_trsm_lower(h2, _ptr(S), k, ks, _ptr(G_new), self.d, self.d), "ffgp_trsm_lower")
        self.W[n:n + k, :n] = B.T
        self.W[n:n + k, n:n + k] = torch.tril(S[:, :k])
        self.X[n:n + k] = Xn
        self.Gamma = torch.cat([self.Gamma, G_new], 0)
        self.n = n + k
        self.alpha = None


class PosteriorCache:
    """Keeps the `Posterior` of a model while the SAME tensor objects (training inputs, targets, every parameter) come
    back with unchanged in-place version counters: in-place updates bump `_version`, `p.data = ...` moves the pointer,
    and weak references make sure a recycled address can never alias.  Not part of a model's state (pickles empty).

    Invalidation rule: edits that bypass the version counter -- `p.data.copy_(...)`, `.data.clamp_()`, writes through a
    numpy array that shares the tensor's memory (`torch.from_numpy`) -- are NOT seen; call the model's
    `clear_posterior_cache()` after such an edit (the reference refactorises on every call and needs no such rule).  The
    cache pins one N x N fp64 factor per model (2 GB at N = 16384); `clear_posterior_cache()` releases it, and
    `model.cache_posterior = False` turns the cache off for that model (every call refactorises, as the reference)."""

    def __init__(self):
        self._c = None
        self.enabled = True

    def __getstate__(self):
        return {"_c": None, "enabled": self.enabled}

    def get(self, objs, build):
        """(posterior, fresh): the cached one if `objs` are unchanged, else `build()` (which is then cached)"""
        vers = tuple((t._version, t.data_ptr()) for t in objs)
        c = self._c
        if c is not None and len(c[0]) == len(objs) and all(r() is t for r, t in zip(c[0], objs)) and c[1] == vers:
            return c[2], False
        post = build()
        self._c = ([weakref.ref(t) for t in objs], vers, post) if self.enabled else None
        return post, True

    @property
    def posterior(self):
        return self._c[2] if self._c is not None else None

    def clear(self):
        self._c = None


class PosteriorCacheMixin:
    """`clear_posterior_cache()` / `cache_posterior` for the GP modules that keep a `_pcache` (see PosteriorCache)."""

    def clear_posterior_cache(self):
        self._pcache.clear()

    @property
    def cache_posterior(self):
        return self._pcache.enabled

    @cache_posterior.setter
    def cache_posterior(self, on):
        self._pcache.enabled = bool(on)
        if not on:
            self._pcache.clear()
