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
from .train import train_many  # noqa: F401  (K Adam steps of independent models per library call: the reference's train loops)


def _kfun(k):
    return k.kfun() if hasattr(k, "kfun") else (0, 1.0)

JITTER = 1e-6
EPS = 1e-10
PI = 3.1415


def _split(y_train):
    if isinstance(y_train, list):
        return y_train[0], y_train[1]
    return y_train, None


def negative_log_likelihood_many(models, xs, ys):
    """[m.negative_log_likelihood(x, y) for m, x, y in zip(models, xs, ys)] as one tensor [F] -- the per-fidelity / per-seed loops of
    the reference's experiments (Experiments/GAR_Aligned/exp_aligned.py:58-126, FidelityFusion_Models/ResGP.py:78-112) train
    independent models one after the other, each step a separate call; independent models can take their steps together.
    With every model on the GPU in fp64 the batch is served by at most two library calls:
      * the small models (N <= 128, D <= 16, d <= 16): one workgroup per model (`ffgp_nlml_fused_small_batch`);
      * the larger ones, when there are at least two: ONE factorisation chain (`ffgp_nlml_fused_batch`) -- eight N = 4096 blocks then
        cost what three cost one after the other.  Since round 5 the members may have DIFFERENT sizes (the reference's fidelities do:
        300 / 300 / 250 points in FidelityFusion_Models/ResGP.py:121-136, 100 low against 4..32 high in
        Experiments/GAR_Aligned/exp_aligned.py:66-74): a member drops out of the chain when its columns are used up;
    anything else falls back to the individual calls.  Values and gradients are those of the individual calls, bit for bit."""
    items = []
    for m, x, y in zip(models, xs, ys):
        y, y_var = _split(y)
        lk = F.raw_many_ok(m.kernel, x, y, m.log_beta) if (y_var is None or F.raw_ok(y_var)) else None
        if lk is None or (len(items) and x.device != items[0]["X"].device):
            items = None
            break
        items.append({"X": x, "Y": y, "lk": lk, "rdadd": m.log_beta, "dadd_link": F._lib.LINK_EXP_NEG, "dadd_c": JITTER, "diag_vec": y_var,
                      "variant": F.FFGP_LL_V1, "pi_const": PI, "sign": -1.0})
    single = lambda i: models[i].negative_log_likelihood(xs[i], ys[i]).reshape(())
    if items is None:
        return torch.stack([single(i) for i in range(len(models))])
    shapes = [(it["X"].shape[0], it["Y"].shape[1]) for it in items]
    if F.many_batchable(shapes):
        return F.nlml_raw_many(items)
    # a mix of small and larger models: one call per kind, results back in the caller's order
    out = [None] * len(items)
    small = [i for i, (n, _) in enumerate(shapes) if n <= F.SMALL_BATCH_MAX_N]
    large = [i for i, (n, _) in enumerate(shapes) if n > F.SMALL_BATCH_MAX_N]
    for idx in (small, large):
        if len(idx) >= 2 and F.many_batchable([shapes[i] for i in idx]):
            vals = F.nlml_raw_many([items[i] for i in idx])
            for k, i in enumerate(idx):
                out[i] = vals[k]
        else:
            for i in idx:
                out[i] = single(i)
    return torch.stack(out)


class cigp(F.PosteriorCacheMixin, nn.Module):
    def __init__(self, kernel, log_beta):
        super().__init__()
        self.kernel = kernel
        self.log_beta = nn.Parameter(torch.tensor([log_beta]))
        self._pcache = F.PosteriorCache()   # the factor of the last (x_train, y_train, parameters)

    @property
    def _post(self):   # (kept for tools/tests) the cached F.Posterior, if any
        return self._pcache.posterior

    def _cached_posterior(self, x_train, y_train, first_query=None, var_add_all=0.0):
        """(F.Posterior, fresh): the factor of (x_train, y_train, parameters), kept while the SAME tensor objects are
        passed with unchanged in-place version counters (in-place updates bump the version; `p.data = ...` moves the
        pointer).  The reference refactorises Sigma on every call (:31-35); repeated queries of a trained model
        (acquisition loops, serving) cost one TRSM sweep here instead of N^3/3."""
        def build():
            with torch.no_grad():
                noise = self.log_beta.exp().pow(-1)
                if not hasattr(self.kernel, "effective"):      # SumKernel / ProductKernel over library kernels: descriptor tree
                    return F.Posterior(x_train, y_train, None, None, noise.double() + JITTER, first_query=first_query,
                                       var_add_all=var_add_all, tree=self.kernel.pair())
                w, amp, clamp = self.kernel.effective()
                return F.Posterior(x_train, y_train, w, amp, noise.double() + JITTER, clamp=clamp, kfun=_kfun(self.kernel),
                                   first_query=first_query, var_add_all=var_add_all)
        return self._pcache.get([x_train, y_train] + list(self.parameters()), build)

    def forward(self, x_train, y_train, x_test):
        y_train, _ = _split(y_train)
        odt = y_train.dtype if y_train.dtype.is_floating_point else torch.float64
        # one (w, amp, profile) kernel, or a composition of library kernels the tile pass evaluates (the demos' SumKernel(Linear, Matern))
        fused = hasattr(self.kernel, "effective") or (hasattr(self.kernel, "fusable") and self.kernel.fusable())
        if fused and torch.is_grad_enabled() and not (x_train.requires_grad or y_train.requires_grad
                                                      or any(p.requires_grad for p in self.parameters())):
            # autograd on, model frozen (`model.requires_grad_(False)`): only the query points can want gradients -- the
            # acquisition optimisers of Bayesian_optimization/acq.py:50-62 -- so the factor is cached and the query is
            # differentiated on it
            post, _ = self._cached_posterior(x_train, y_train)
            mean, var = post.predict_diff(x_test, full_cov=True, var_add_all=float(self.log_beta.exp().pow(-1)))
            return mean.to(device=y_train.device, dtype=odt), var.to(device=y_train.device, dtype=odt)
        # the fused posterior is a no_grad path (every prediction call of the reference's models sits under
        # torch.no_grad()); with autograd on and a trainable model the same quantities are composed from differentiable
        # pieces, gradients flowing to x_test, the parameters and y_train as in the reference
        if not fused or torch.is_grad_enabled():
            return self._forward_composed(x_train, y_train, x_test)
        noise = float(self.log_beta.exp().pow(-1))
        # first query: K_s^T rides in the factorisation (the cost of the fused one-shot posterior), factor kept
        post, fresh = self._cached_posterior(x_train, y_train, first_query=x_test, var_add_all=noise)
        mean, var = post.first if fresh else post.predict(x_test, full_cov=True, var_add_all=noise)
        return mean.to(device=y_train.device, dtype=odt), var.to(device=y_train.device, dtype=odt)

    # composed kernels (SumKernel(LinearKernel, MaternKernel) of the reference's own demos, cigp_v10.py:81,111,147):
    # the parts are evaluated on the device, Sigma is composed there and enters the fused factorisation as cov_dev
    def _forward_composed(self, x_train, y_train, x_test):
        noise = self.log_beta.exp().pow(-1)
        Sigma = F.add_diagonal(F.kernel_on_device(self.kernel, x_train, x_train), noise, JITTER)
        K_s = F.kernel_on_device(self.kernel, x_train, x_test)
        K_ss = F.kernel_on_device(self.kernel, x_test, x_test)
        # a trainable model queried again and again with unchanged parameters (an acquisition loop that never froze it):
        # the factor of Sigma is the cached one, only the backward formulas run -- gradients still reach the parameters and y
        fused = hasattr(self.kernel, "effective") or (hasattr(self.kernel, "fusable") and self.kernel.fusable())
        post = self._cached_posterior(x_train, y_train)[0] if fused else None
        mean, var = F.conditional_gaussian(y_train, Sigma, K_s, K_ss, factor=post)
        var = var + noise.to(var.device)
        odt = y_train.dtype if y_train.dtype.is_floating_point else torch.float64
        return mean.to(device=y_train.device, dtype=odt), var.to(device=y_train.device, dtype=odt)

    def _nll_composed(self, x_train, y_train, y_var):
        K = F.kernel_on_device(self.kernel, x_train, x_train)
        Sigma = F.add_diagonal(K, self.log_beta.exp().pow(-1), JITTER, y_var.diag() if y_var is not None else None)
        return F.gaussian_nll_from_cov(y_train, Sigma, F.FFGP_LL_V1, PI)

    def negative_log_likelihood(self, x_train, y_train):
        y_train, y_var = _split(y_train)
        lk = F.raw_path(self.kernel, x_train, y_train, self.log_beta) if (y_var is None or F.raw_ok(y_var)) else None
        if lk is not None:
            # everything already on the GPU in fp64: ONE library call on the raw parameters (abs / reciprocal / exp maps inside)
            return F.nlml_raw(x_train, y_train, lk, self.log_beta, F._lib.LINK_EXP_NEG, JITTER, diag_vec=y_var, variant=F.FFGP_LL_V1,
                              pi_const=PI, sign=-1.0)
        pr = self.kernel.pair() if hasattr(self.kernel, "pair") else None
        if pr is not None and F.pair_inputs_plain(x_train, y_var):   # SumKernel(LinearKernel, MaternKernel) of the demos (:81,111,147): two descriptors, fused like a single kernel
            return -F.nlml_pair(x_train, y_train, pr[0], pr[1], diag_add=self.log_beta.exp().pow(-1).double() + JITTER, diag_vec=y_var,
                                variant=F.FFGP_LL_V1, pi_const=PI, **F._slot_args())
        if not hasattr(self.kernel, "effective"):
            return -self._nll_composed(x_train, y_train, y_var)
        w, amp, clamp = self.kernel.effective()
        # (.double() first: with an fp32 log_beta -- the reference's default dtype -- the sum would round the jitter away in fp32;
        # the reference adds the two terms to the fp64 kernel matrix one after the other, :57-58)
        diag_add = self.log_beta.exp().pow(-1).double() + JITTER
        nll = F.nlml(x_train, y_train, w, amp, diag_add=diag_add, diag_vec=y_var, clamp=clamp, variant=F.FFGP_LL_V1,
                     pi_const=PI, **F._slot_args(), kfun=_kfun(self.kernel))
        return -nll
