"""The fused likelihood calls as autograd Functions: effective parameters (`nlml`), the modules' raw parameters (`nlml_raw`), batches
of independent problems (`nlml_raw_many` / `nlml_many`: one workgroup each, or one shared factorisation chain), composed kernels
(`nlml_pair`) and the one-shot fused posterior (`predict`).  Reference call sites: GaussianProcess/cigp_v10.py:50-69,
gp_computation_pack.py:112-136, MFGP_ver2023May/base_gp/cigp.py:118-137; the backward replaces `loss.backward()` at
FidelityFusion_Models/ResGP.py:84-88.
"""
import ctypes as C
import math
import threading

import torch

from . import _lib
from ._common import NEG_INF, _check_same_D, _check_xy, _dev, _device_of, _ptr, _raise_not_pd, _split_kfun, _weights
from ._lib import FFGP_LL_V1, FFGP_LL_V2, FFGP_VAR_DIAG, FFGP_VAR_FULL, PI_TRUNC, Grads, KDesc, KDescGrads, Problem, check, lib
from .blocks import _pending, concurrent_blocks
from .kdesc import FFGP_KFUN_LINEAR, FFGP_KOP_PRODUCT, FFGP_KOP_SUM, FFGP_TREE_BALANCED, FFGP_TREE_CHAIN, _PAIR_KEYS, _pair_descs, _pair_grad_buffers, _pair_grads_out, _pair_split, _tree_spec


def _problem(dev, X, Y, w, amp, diag_add, diag_vec, add_mat, add_all, mean_jitter, clamp, variant, pi_const, keep,
             kfun=(0, 1.0)):
    Xd, Yd = _dev(X, dev), _dev(Y, dev)
    _check_xy(Xd, Yd)
    n, D = Xd.shape
    d = Yd.shape[1]
    if w is not None:   # (a composed kernel carries its parts in Problem.pair instead)
        wd = _weights(w, D, dev)
        ad = _dev(amp.reshape(-1)[:1], dev)
        if ad.numel() != 1:
            raise ValueError("the kernel amplitude must hold one value, got shape %s" % (tuple(amp.shape),))
    if diag_vec is not None and tuple(diag_vec.shape) not in ((n,), (n, n)):
        raise ValueError("y_var / diag_vec must be [%d] or [%d, %d] for %d training points, got %s"
                         % (n, n, n, n, tuple(diag_vec.shape)))
    if add_mat is not None and tuple(add_mat.shape) != (n, n):
        raise ValueError("y_var / add_mat must be [%d, %d], got %s" % (n, n, tuple(add_mat.shape)))
    p = Problem()
    p.n, p.D, p.d = n, D, d
    p.X_dev, p.Y_dev = _ptr(Xd), _ptr(Yd)
    keep += [Xd, Yd]
    if w is not None:
        p.w_dev, p.amp_dev = _ptr(wd), _ptr(ad)
        keep += [wd, ad]
    p.clamp_min = clamp
    if diag_add is not None:
        dd = _dev(diag_add.reshape(-1)[:1], dev)
        p.diag_add_dev = _ptr(dd)
        keep.append(dd)
    if diag_vec is not None:
        dv = _dev(diag_vec, dev)
        if dv.dim() == 2:  # an N x N matrix whose diagonal is wanted (cigp_v10.py:59-60): read in place, stride N+1
            p.diag_stride = dv.shape[1] + 1
        else:
            p.diag_stride = 1
        p.diag_vec_dev = _ptr(dv)
        keep.append(dv)
    if add_mat is not None:
        am = _dev(add_mat, dev)
        p.add_mat_dev = _ptr(am)
        p.ld_add = am.shape[1]
        keep.append(am)
    p.add_all = float(add_all)
    p.mean_jitter = float(mean_jitter)
    p.ll_variant = variant
    p.pi_const = pi_const
    p.kfun, p.kparam = int(kfun[0]), float(kfun[1])   # a learnable profile parameter (RQ's alpha) arrives as a tensor
    return p, (n, D, d)


class _NLML(torch.autograd.Function):
    """nll(X, Y; w, amp, diag_add, diag_vec, add_mat) -> 0-dim tensor.  V1: +nll; V2: -LL.
    Gradients (closed form, computed by the same fused call): Y, w, amp, diag_add, diag_vec."""

    @staticmethod
    def forward(ctx, X, Y, w, amp, diag_add, diag_vec, add_mat, add_all, mean_jitter, clamp, variant, pi_const, slot=0,
                defer=False, kfun=(0, 1.0), kparam=None, rec=True):
        dev = _device_of(X, Y, w, amp)
        if kparam is not None:
            kfun = (kfun[0], float(kparam.detach()))
        h = _lib.handle(dev.index, slot)
        _lib.bind_stream(h, dev.index)
        keep = []
        p, (n, D, d) = _problem(dev, X, Y, w, amp, diag_add, diag_vec, add_mat, add_all, mean_jitter, clamp, variant,
                                pi_const, keep, kfun)
        # under torch.no_grad() nothing will ever call backward, although leaf inputs (nn.Parameters, a y with
        # requires_grad) still report requires_grad = True: the gradient pipeline (TRTRI, LAUUM, A A^T, gradient tiles: 2x
        # the forward's flops and 2-3 more N x ld workspaces) only runs when autograd is recording.  `rec` is the caller's
        # grad mode, captured by the wrapper: inside forward() autograd is always off
        needs = [rec and bool(ctx.needs_input_grad[i]) for i in (1, 2, 3, 4, 5, 15)]    # Y, w, amp, diag_add, diag_vec, kparam
        out = torch.empty((), dtype=torch.float64, device=dev)
        g = None
        grads = {}
        if any(needs):
            g = Grads()
            if needs[0]:
                grads["Y"] = torch.empty((n, d), dtype=torch.float64, device=dev)
                g.g_Y_dev = _ptr(grads["Y"])
            # the scalar-sized gradients share one buffer [w (D) | amp | diag_add | kparam]: one scaling launch in backward
            small = torch.empty((D + 3,), dtype=torch.float64, device=dev)
            grads["_small"] = small
            base, step = small.data_ptr(), small.element_size()
            if needs[1]:
                grads["w"] = small[:D]
                g.g_w_dev = C.c_void_p(base)
            if needs[2]:
                grads["amp"] = small[D:D + 1]
                g.g_amp_dev = C.c_void_p(base + D * step)
            if needs[3]:
                grads["diag_add"] = small[D + 1:D + 2]
                g.g_diag_add_dev = C.c_void_p(base + (D + 1) * step)
            if needs[4]:
                grads["diag_vec"] = torch.empty((n,), dtype=torch.float64, device=dev)
                g.g_diag_vec_dev = _ptr(grads["diag_vec"])
            if needs[5]:
                grads["kparam"] = small[D + 2:D + 3]
                g.g_kparam_dev = C.c_void_p(base + (D + 2) * step)
        gref = C.byref(g) if g is not None else None
        if defer:   # enqueue only: the caller collects the status with wait(slot) after launching its other blocks
            check(lib.ffgp_nlml_fused_async(h, C.byref(p), _ptr(out), gref), "ffgp_nlml_fused_async")
            _pending.setdefault((dev.index, slot), []).append(keep)
        else:
            rc = check(lib.ffgp_nlml_fused(h, C.byref(p), _ptr(out), gref), "ffgp_nlml_fused")
            if rc > 0:
                _raise_not_pd(rc, "linalg.cholesky")
        ctx.grads = grads
        ctx.meta = [(t.shape, t.dtype, t.device) if isinstance(t, torch.Tensor) else None
                    for t in (Y, w, amp, diag_add, diag_vec, kparam)]
        return out.to(device=Y.device, dtype=Y.dtype if Y.dtype.is_floating_point else torch.float64)

    @staticmethod
    def backward(ctx, gout):
        gr = dict(ctx.grads)
        small = gr.pop("_small", None)
        if small is not None:   # scale the scalar-sized gradients in one launch, then hand out views
            D = small.numel() - 3
            scaled = small * gout.to(device=small.device, dtype=torch.float64)
            views = {"w": scaled[:D], "amp": scaled[D:D + 1], "diag_add": scaled[D + 1:D + 2], "kparam": scaled[D + 2:D + 3]}

        def fin(key, idx):
            if key not in gr or ctx.meta[idx] is None:
                return None
            shape, dtype, device = ctx.meta[idx]
            if key in ("w", "amp", "diag_add", "kparam"):
                t = views[key]
            else:
                t = gr[key] * gout.to(device=gr[key].device, dtype=torch.float64)
            if key == "w" and math.prod(shape) == 1 and t.numel() > 1:
                t = t.sum().reshape(1)  # a scalar length scale was broadcast over the D input dimensions
            if key == "diag_vec" and len(shape) == 2:
                t = torch.diag_embed(t)
            return t.reshape(shape).to(device=device, dtype=dtype)

        return (None, fin("Y", 0), fin("w", 1), fin("amp", 2), fin("diag_add", 3), fin("diag_vec", 4), None, None, None,
                None, None, None, None, None, None, fin("kparam", 5), None)


DEFER_RAW_ERRORS = False


"""Opt-in.  False (default): a Sigma that is not positive definite raises torch.linalg.LinAlgError from the likelihood call
itself, as `torch.linalg.cholesky` does inside the reference's `negative_log_likelihood` (GaussianProcess/cigp_v10.py:61).
True: a training step through the raw-parameter path (`nlml_raw`: everything GPU-resident, gradients requested) is ENQUEUED and
its status collected in backward(), so the host builds the backward pass while the GPU factors (0.33 -> 0.25 ms per step at
N = 128).  The error then surfaces from `loss.backward()` -- or from the next likelihood call on that device if backward() is
never reached -- and the value is NaN meanwhile; a loss that is evaluated with gradients enabled and never back-propagated only
raises at the next call.  Calls under torch.no_grad(), CPU-resident tensors and every other path always raise at the call."""


_raw_pending = {}   # device index -> handle with an enqueued, not yet collected raw-parameter call (only with DEFER_RAW_ERRORS)


_raw_pending_lock = threading.Lock()


def _settle_raw(dev_index):
    with _raw_pending_lock:
        h = _raw_pending.pop(dev_index, None)
    if h is not None:
        rc = check(lib.ffgp_wait(h), "ffgp_wait")
        if rc > 0:
            _raise_not_pd(rc, "linalg.cholesky")


class _NLMLRaw(torch.autograd.Function):
    """sign * nll on the modules' RAW parameters (ffgp_nlml_fused_raw): the raw -> effective maps and their chain rule run inside the
    library call, so one training step is ONE autograd node and one library call instead of a dozen elementwise torch kernels
    with their autograd nodes -- the whole cost of a step at the sizes the reference's demos run (N = 16 ... 300).
    Everything must already live on one GPU in fp64 (see `raw_ok`).  Gradients: Y, raw w, raw amp, raw diag_add, diag_vec, kparam."""

    @staticmethod
    def forward(ctx, X, Y, rw, ramp, rdadd, diag_vec, add_mat, kparam, links, add_all, mean_jitter, clamp, variant, pi_const, kfun_id,
                sign, rec, kp_const):
        dev = X.device
        if _raw_pending:
            _settle_raw(dev.index)      # an earlier (deferred) step never reached backward(): its status is due now
        h = _lib.handle(dev.index, 0)
        _lib.bind_stream(h, dev.index)
        n, D = X.shape
        d = Y.shape[1]
        p = Problem()
        p.n, p.D, p.d = n, D, d
        p.X_dev, p.Y_dev, p.w_dev, p.amp_dev = X.data_ptr(), Y.data_ptr(), rw.data_ptr(), ramp.data_ptr()
        p.clamp_min = clamp
        if rdadd is not None:
            p.diag_add_dev = rdadd.data_ptr()
        if diag_vec is not None:
            p.diag_stride = diag_vec.shape[1] + 1 if diag_vec.dim() == 2 else 1
            p.diag_vec_dev = diag_vec.data_ptr()
        if add_mat is not None:
            p.add_mat_dev, p.ld_add = add_mat.data_ptr(), add_mat.shape[1]
        p.add_all, p.mean_jitter, p.ll_variant, p.pi_const = add_all, mean_jitter, variant, pi_const
        p.kfun, p.kparam = kfun_id, (float(kparam) if kparam is not None else kp_const)
        need = ctx.needs_input_grad
        nY, nw, na, nd, nv, nk = (rec and bool(need[i]) for i in (1, 2, 3, 4, 5, 7))
        Dw = rw.numel()
        out = torch.empty((), dtype=torch.float64, device=dev)
        g = None
        buf = None
        oY = Dw + 3
        ov = oY + (n * d if nY else 0)
        if nY or nw or na or nd or nv or nk:
            # ONE buffer for every gradient [raw w | raw amp | raw diag_add | kparam | Y (n d) | diag_vec (n)]: one scaling launch in backward
            g = Grads()
            buf = torch.empty((ov + (n if nv else 0),), dtype=torch.float64, device=dev)
            base = buf.data_ptr()
            if nw:
                g.g_w_dev = base
            if na:
                g.g_amp_dev = base + 8 * Dw
            if nd:
                g.g_diag_add_dev = base + 8 * (Dw + 1)
            if nk:
                g.g_kparam_dev = base + 8 * (Dw + 2)
            if nY:
                g.g_Y_dev = base + 8 * oY
            if nv:
                g.g_diag_vec_dev = base + 8 * ov
        links.out_scale = sign          # the sign (+LL for the reference's `negative_log_likelihood`) is applied inside the call
        if g is not None and DEFER_RAW_ERRORS:
            check(lib.ffgp_nlml_fused_raw_async(h, C.byref(p), C.byref(links), out.data_ptr(), C.byref(g)), "ffgp_nlml_fused_raw_async")
            with _raw_pending_lock:
                _raw_pending[dev.index] = h
            ctx.dev_index = dev.index
        else:
            rc = check(lib.ffgp_nlml_fused_raw(h, C.byref(p), C.byref(links), out.data_ptr(), C.byref(g) if g is not None else None),
                       "ffgp_nlml_fused_raw")
            if rc > 0:
                _raise_not_pd(rc, "linalg.cholesky")
            ctx.dev_index = None
        ctx.pack = (buf, Dw, n, d, oY, ov, (nw, na, nd, nk, nY, nv), rw.shape, ramp.shape, None if rdadd is None else rdadd.shape,
                    None if diag_vec is None else diag_vec.shape, None if kparam is None else kparam.shape)
        return out

    @staticmethod
    def backward(ctx, gout):
        buf, Dw, n, d, oY, ov, (nw, na, nd, nk, nY, nv), sw, sa, sd, sv, sk = ctx.pack
        if ctx.dev_index is not None:
            _settle_raw(ctx.dev_index)
        gw = ga = gd = gk = gYo = gvo = None
        if buf is not None:
            sc = buf * gout
            if nw:
                gw = sc[:Dw].view(sw)
            if na:
                ga = sc[Dw:Dw + 1].view(sa)
            if nd:
                gd = sc[Dw + 1:Dw + 2].view(sd)
            if nk:
                gk = sc[Dw + 2:Dw + 3].view(sk)
            if nY:
                gYo = sc[oY:oY + n * d].view(n, d)
            if nv:
                gvo = sc[ov:ov + n]
                if len(sv) == 2:
                    gvo = torch.diag_embed(gvo)
        return (None, gYo, gw, ga, gd, gvo, None, gk) + (None,) * 10


class _NLMLRawMany(torch.autograd.Function):
    """F independent small problems in one library call (ffgp_nlml_fused_small_batch): values [F]; gradients for every problem's
    Y, raw w, raw amp, raw diag_add, diag_vec.  Inputs per problem: X, Y, rw, ramp, rdadd, diag_vec (6 tensors, None allowed for the
    last two); `metas[f]` = (links, mean_jitter, clamp, variant, pi_const, kfun_id, kparam, sign)."""

    @staticmethod
    def forward(ctx, metas, rec, *tensors):
        nF = len(metas)
        dev = tensors[0].device
        if dev.index in _raw_pending:
            _settle_raw(dev.index)
        h = _lib.handle(dev.index, 0)
        _lib.bind_stream(h, dev.index)
        chain_batch = tensors[0].shape[0] > SMALL_BATCH_MAX_N     # (nlml_raw_many only builds homogeneous batches of either kind)
        P = (Problem * nF)()
        L = (_lib.Links * nF)()
        G = (Grads * nF)()
        out = torch.empty((nF,), dtype=torch.float64, device=dev)
        layout, total = [], 0
        any_grad = False
        for f in range(nF):
            X, Y, rw, ramp, rdadd, dvec = tensors[6 * f:6 * f + 6]
            links, mean_jitter, clamp, variant, pi_const, kfun_id, kparam, sign = metas[f]
            n, D = X.shape
            d = Y.shape[1]
            p = P[f]
            p.n, p.D, p.d = n, D, d
            p.X_dev, p.Y_dev, p.w_dev, p.amp_dev = X.data_ptr(), Y.data_ptr(), rw.data_ptr(), ramp.data_ptr()
            p.clamp_min = clamp
            if rdadd is not None:
                p.diag_add_dev = rdadd.data_ptr()
            if dvec is not None:
                p.diag_stride = dvec.shape[1] + 1 if dvec.dim() == 2 else 1
                p.diag_vec_dev = dvec.data_ptr()
            p.mean_jitter, p.ll_variant, p.pi_const, p.kfun, p.kparam = mean_jitter, variant, pi_const, kfun_id, kparam
            L[f] = links
            L[f].out_scale = sign
            need = ctx.needs_input_grad[2 + 6 * f:2 + 6 * f + 6]
            nY, nw, na, nd, nv = (rec and bool(need[i]) for i in (1, 2, 3, 4, 5))
            Dw = rw.numel()
            seg = (total, Dw, n, d, (nw, na, nd, nY, nv), rw.shape, ramp.shape, None if rdadd is None else rdadd.shape,
                   None if dvec is None else dvec.shape)
            layout.append(seg)
            total += Dw + 3 + n * d + n
            any_grad = any_grad or nY or nw or na or nd or nv
        buf = torch.empty((total,), dtype=torch.float64, device=dev) if any_grad else None
        if buf is not None:
            base = buf.data_ptr()
            for f, (off, Dw, n, d, (nw, na, nd, nY, nv), *_r) in enumerate(layout):
                g = G[f]
                b = base + 8 * off
                if nw:
                    g.g_w_dev = b
                if na:
                    g.g_amp_dev = b + 8 * Dw
                if nd:
                    g.g_diag_add_dev = b + 8 * (Dw + 1)
                if nY:
                    g.g_Y_dev = b + 8 * (Dw + 3)
                if nv:
                    g.g_diag_vec_dev = b + 8 * (Dw + 3 + n * d)
        if chain_batch:
            # equal-shape blocks beyond the one-workgroup sizes: ONE factorisation chain for all of them (ffgp_nlml_fused_batch);
            # the status is per block, and the FIRST block that is not positive definite raises -- the reference's loop over models
            # would have stopped there (FidelityFusion_Models/ResGP.py:82-88)
            status = (C.c_int * nF)()
            rc = check(lib.ffgp_nlml_fused_batch(h, nF, P, L, out.data_ptr(), G if buf is not None else None, status),
                       "ffgp_nlml_fused_batch")
            if rc > 0:
                bad = next(f for f in range(nF) if status[f] > 0)
                _raise_not_pd(status[bad], "linalg.cholesky (block %d of the batch)" % bad)
            ctx.dev_index = None
        elif buf is not None and DEFER_RAW_ERRORS:
            check(lib.ffgp_nlml_fused_small_batch_async(h, nF, P, L, out.data_ptr(), G), "ffgp_nlml_fused_small_batch_async")
            with _raw_pending_lock:
                _raw_pending[dev.index] = h
            ctx.dev_index = dev.index
        else:
            rc = check(lib.ffgp_nlml_fused_small_batch(h, nF, P, L, out.data_ptr(), G if buf is not None else None),
                       "ffgp_nlml_fused_small_batch")
            if rc > 0:
                _raise_not_pd(rc, "linalg.cholesky")
            ctx.dev_index = None
        ctx.pack = (buf, layout)
        return out

    @staticmethod
    def backward(ctx, gout):
        buf, layout = ctx.pack
        if ctx.dev_index is not None:
            _settle_raw(ctx.dev_index)
        grads = []
        for f, (off, Dw, n, d, (nw, na, nd, nY, nv), sw, sa, sd, sv) in enumerate(layout):
            gX = gY = gw = ga = gd = gv = None
            if buf is not None:
                sc = buf[off:off + Dw + 3 + n * d + n] * gout[f]
                if nw:
                    gw = sc[:Dw].view(sw)
                if na:
                    ga = sc[Dw:Dw + 1].view(sa)
                if nd:
                    gd = sc[Dw + 1:Dw + 2].view(sd)
                if nY:
                    gY = sc[Dw + 3:Dw + 3 + n * d].view(n, d)
                if nv:
                    gv = sc[Dw + 3 + n * d:]
                    if len(sv) == 2:
                        gv = torch.diag_embed(gv)
            grads += [gX, gY, gw, ga, gd, gv]
        return (None, None) + tuple(grads)


SMALL_BATCH_MAX_N, SMALL_BATCH_MAX_D, SMALL_BATCH_MAX_d = 128, 16, 16


def nlml_raw_many(items):
    """items: list of dicts {X, Y, lk (kernel.links()), rdadd, dadd_link, dadd_c, diag_vec, mean_jitter, variant, pi_const, sign} -- F
    independent small problems (n <= 128, D <= 16, d <= 16) evaluated by ONE library call; returns the tensor [F] of sign * nll."""
    metas, tensors = [], []
    for it in items:
        lk = it["lk"]
        L = _lib.Links()
        L.w_link, L.w_c, L.w_broadcast = lk["w_link"], lk["w_c"], 1 if lk["w"].numel() == 1 and it["X"].shape[1] > 1 else 0
        L.amp_link, L.amp_c = lk["amp_link"], 0.0
        L.dadd_link, L.dadd_c = it["dadd_link"], it["dadd_c"]
        kp = lk.get("kparam")
        if isinstance(kp, torch.Tensor):
            raise ValueError("nlml_raw_many: learnable profile parameters (RationalQuadraticKernel.alpha) take the single-problem call")
        metas.append((L, float(it.get("mean_jitter", 0.0)), lk["clamp"], it.get("variant", FFGP_LL_V1), it.get("pi_const", PI_TRUNC),
                      lk["kfun"], 1.0 if kp is None else float(kp), float(it.get("sign", 1.0))))
        tensors += [it["X"], it["Y"], lk["w"], lk["amp"], it["rdadd"], it.get("diag_vec")]
    metas = tuple(metas)
    rec = torch.is_grad_enabled()
    if items[0]["X"].shape[0] <= SMALL_BATCH_MAX_N:
        return _NLMLRawMany.apply(metas, rec, *tensors)
    return _chain_batches(items, metas, tensors, rec, 0, len(items))


CHAIN_BATCH_MAX_F = 256      # ffgp_nlml_fused_batch's limit on the members of one chain (include/ffgp.h)


def _chain_batches(items, metas, tensors, rec, lo, hi):
    """members [lo, hi) of a shared-chain batch -> tensor [hi - lo].  include/ffgp.h leaves the fallback to the caller: the library
    refuses the batch (FFGP_ERR_ARG) in states it otherwise allows -- option `naive` = 1, `diag_v2` = 0, more than 256 members -- and
    reports FFGP_ERR_ALLOC when the F-fold workspace does not fit.  Here: chunks of at most 256 members; a chunk that does not fit is
    halved until it does; a chunk the library refuses (or a single leftover member) goes through the individual calls, which work in
    all of those states and need one block of memory.  Values and gradients are the individual calls' either way."""
    if hi - lo > CHAIN_BATCH_MAX_F:
        mid = lo + CHAIN_BATCH_MAX_F
        return torch.cat([_chain_batches(items, metas, tensors, rec, lo, mid), _chain_batches(items, metas, tensors, rec, mid, hi)])
    if hi - lo >= 2:
        try:
            return _NLMLRawMany.apply(metas[lo:hi], rec, *tensors[6 * lo:6 * hi])
        except _lib.FFGPError as e:
            if e.code == _lib.FFGP_ERR_ALLOC and hi - lo >= 4:
                mid = (lo + hi) // 2
                return torch.cat([_chain_batches(items, metas, tensors, rec, lo, mid), _chain_batches(items, metas, tensors, rec, mid, hi)])
            if e.code not in (_lib.FFGP_ERR_ARG, _lib.FFGP_ERR_ALLOC):
                raise
    return torch.stack([_single_raw(it) for it in items[lo:hi]])


def _single_raw(it):
    """one member of a batch through the single-problem call (same links, same sign)"""
    return nlml_raw(it["X"], it["Y"], it["lk"], it["rdadd"], it["dadd_link"], it["dadd_c"], diag_vec=it.get("diag_vec"),
                    mean_jitter=float(it.get("mean_jitter", 0.0)), variant=it.get("variant", FFGP_LL_V1),
                    pi_const=it.get("pi_const", PI_TRUNC), sign=float(it.get("sign", 1.0))).reshape(())


def nlml_many(Xs, Ys, ws, amps, diag_adds, clamp=NEG_INF, pi_const=PI_TRUNC):
    """[nlml(X, Y, w, amp, diag_add=dadd, clamp=clamp) for ...] as one tensor [F] through ONE factorisation chain
    (ffgp_nlml_fused_batch): F >= 2 blocks with n > 128 (one shape, or different shapes up to 12288 rows each), everything on one
    GPU in fp64, effective parameters (w [D], amp [1], diag_add [1] per block; squared-exponential profile).  Gradients flow to Y,
    w, amp and diag_add.
    The per-fidelity blocks of one rank in the sharded workloads (bench.py `cigar4`, `gar8`) are such a batch."""
    items = []
    for X, Y, w, amp, dadd in zip(Xs, Ys, ws, amps, diag_adds):
        if not raw_ok(X, Y, w, amp, dadd) or w.numel() != X.shape[1]:
            raise ValueError("nlml_many: every tensor must live on one GPU in fp64, contiguous, with w of length D")
        lk = {"w": w, "w_link": _lib.LINK_ID, "w_c": 0.0, "amp": amp, "amp_link": _lib.LINK_ID, "clamp": clamp, "kfun": 0}
        items.append({"X": X, "Y": Y, "lk": lk, "rdadd": dadd, "dadd_link": _lib.LINK_ID, "dadd_c": 0.0, "pi_const": pi_const})
    if not many_batchable([(it["X"].shape[0], it["Y"].shape[1]) for it in items]):
        raise ValueError("nlml_many: at least two blocks with n > %d each (of one shape, or of different shapes with n <= %d)"
                         % (SMALL_BATCH_MAX_N, RAGGED_CHAIN_MAX_N))
    return nlml_raw_many(items)


def raw_many_ok(kernel, x_train, y_train, *others):
    """`raw_path` + the limits of the batched calls: up to SMALL_BATCH_MAX_N points the one-workgroup batch (any mix of shapes,
    D, d <= 16); beyond that the shared-chain batch (D <= 128; which sets of shapes share a chain: `many_batchable`)"""
    lk = raw_path(kernel, x_train, y_train, *others)
    if lk is None or isinstance(lk.get("kparam"), torch.Tensor):
        return None
    if x_train.shape[0] > SMALL_BATCH_MAX_N:
        return lk if x_train.shape[1] <= 128 else None
    if x_train.shape[1] > SMALL_BATCH_MAX_D or y_train.shape[1] > SMALL_BATCH_MAX_d:
        return None
    return lk


RAGGED_CHAIN_MAX_N = 12288     # members of a ragged shared chain (blocks of different sizes): ffgp_potrf_ragged's limit


def many_batchable(shapes):
    """shapes: [(n, d)] of the members.  One library call serves them when they are all small (n <= SMALL_BATCH_MAX_N: one
    workgroup each), or at least two larger blocks (ffgp_nlml_fused_batch: they share ONE factorisation chain) -- of one shape, or,
    since round 5, of different shapes up to RAGGED_CHAIN_MAX_N rows each (the ragged chain: a member drops out of the chain's
    launches when its columns are used up)."""
    if all(n <= SMALL_BATCH_MAX_N for n, _ in shapes):
        return True
    if len(shapes) < 2 or any(n <= SMALL_BATCH_MAX_N for n, _ in shapes):
        return False
    return len(set(shapes)) == 1 or all(n <= RAGGED_CHAIN_MAX_N for n, _ in shapes)


def raw_ok(*tensors):
    """the raw-parameter fast path needs every tensor resident on ONE GPU in fp64, contiguous, no concurrent-block context, and
    inputs that carry no gradient of their own (the fused call has no input gradients)"""
    if concurrent_blocks.active is not None or _lib.current_slot() != 0:   # (the raw path lives on handle 0 of its GPU)
        return False
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float64 and t.is_contiguous()):
            return False
        if dev is None:
            dev = t.device
        elif t.device != dev:
            return False
    return dev is not None


def nlml_raw(X, Y, lk, rdadd, dadd_link, dadd_c, diag_vec=None, add_mat=None, add_all=0.0, mean_jitter=0.0, variant=FFGP_LL_V1,
             pi_const=PI_TRUNC, sign=1.0):
    """sign * nll through ffgp_nlml_fused_raw.  lk: the kernel's `links()` dict (raw tensors, link ids, clamp, kfun)."""
    if X.dim() != 2 or Y.dim() != 2 or X.shape[0] != Y.shape[0]:
        _check_xy(X, Y)
    if torch.is_grad_enabled() and (X.requires_grad or (add_mat is not None and add_mat.requires_grad)):
        raise ValueError("nlml_raw has no input gradients")
    L = _lib.Links()
    L.w_link, L.w_c, L.w_broadcast = lk["w_link"], lk["w_c"], 1 if lk["w"].numel() == 1 and X.shape[1] > 1 else 0
    L.amp_link, L.amp_c = lk["amp_link"], 0.0
    L.dadd_link, L.dadd_c = dadd_link, dadd_c
    kparam = lk.get("kparam")
    kt = kparam if isinstance(kparam, torch.Tensor) else None
    return _NLMLRaw.apply(X, Y, lk["w"], lk["amp"], rdadd, diag_vec, add_mat, kt, L, float(add_all), float(mean_jitter), lk["clamp"],
                          variant, pi_const, lk["kfun"], float(sign), torch.is_grad_enabled(),
                          1.0 if (kparam is None or kt is not None) else float(kparam))


def raw_path(kernel, x_train, y_train, *others):
    """the kernel's `links()` when the raw-parameter fast path applies to this call, else None"""
    lk = kernel.links() if hasattr(kernel, "links") else None
    if lk is None:
        return None
    kp = lk.get("kparam")
    if not raw_ok(x_train, y_train, lk["w"], lk["amp"], kp if isinstance(kp, torch.Tensor) else None, *others):
        return None
    if x_train.dim() != 2 or y_train.dim() != 2 or x_train.shape[0] != y_train.shape[0] or x_train.shape[1] > 128:
        return None
    if lk["w"].numel() not in (1, x_train.shape[1]) or lk["amp"].numel() != 1:
        return None
    if torch.is_grad_enabled() and any(isinstance(t, torch.Tensor) and t.requires_grad for t in (x_train,) + others[1:]):
        return None   # (others[0] is the noise parameter; inputs and full y_var matrices have no gradient on the fused path)
    return lk


def nlml(X, Y, w, amp, diag_add=None, diag_vec=None, add_mat=None, add_all=0.0, mean_jitter=0.0, clamp=NEG_INF,
         variant=FFGP_LL_V1, pi_const=PI_TRUNC, slot=None, defer=False, kfun=(0, 1.0)):
    """Negative log marginal likelihood of one GP block through the fused HIP path (assemble -> blocked Cholesky
    with Y^T riding as passenger rows -> reductions -> closed-form gradients).

    slot / defer: independent blocks can overlap on one GPU -- issue each under its own torch stream with its own
    `slot` and `defer=True`, then call `wait(slot)` (see `concurrent_blocks`).  slot=None: the calling thread's slot (0, or the
    one a `threaded_blocks` worker runs under) -- resolved here, so that backward, which runs on autograd's thread, uses the same."""
    if slot is None:
        slot = _lib.current_slot()
    kfun, kparam = _split_kfun(kfun)
    return _NLML.apply(X, Y, w, amp, diag_add, diag_vec, add_mat, add_all, mean_jitter, clamp, variant, pi_const, slot,
                       defer, kfun, kparam, torch.is_grad_enabled())


@torch.no_grad()
def predict(X, Y, Xs, w, amp, diag_add=None, diag_vec=None, add_mat=None, add_all=0.0, mean_jitter=0.0, clamp=NEG_INF,
            full_cov=True, var_add_all=0.0, kfun=(0, 1.0)):
    """Posterior mean [Nt, d] and covariance [Nt, Nt] (or variance [Nt]) at Xs."""
    dev = _device_of(X, Y, Xs, w, amp)
    h = _lib.handle(dev.index)
    _lib.bind_stream(h, dev.index)
    keep = []
    kfun, _ = _split_kfun(kfun)
    p, (n, D, d) = _problem(dev, X, Y, w, amp, diag_add, diag_vec, add_mat, add_all, mean_jitter, clamp, FFGP_LL_V1,
                            PI_TRUNC, keep, kfun)
    Xsd = _dev(Xs, dev)
    _check_same_D(keep[0], Xsd)
    nt = Xsd.shape[0]
    mean = torch.empty((nt, d), dtype=torch.float64, device=dev)
    var = torch.empty((nt, nt) if full_cov else (nt,), dtype=torch.float64, device=dev)
    rc = check(lib.ffgp_predict(h, C.byref(p), _ptr(Xsd), nt, FFGP_VAR_FULL if full_cov else FFGP_VAR_DIAG,
                                float(var_add_all), _ptr(mean), _ptr(var), nt), "ffgp_predict")
    if rc > 0:
        _raise_not_pd(rc, "linalg.cholesky")
    odt = Y.dtype if Y.dtype.is_floating_point else torch.float64
    return mean.to(device=Y.device, dtype=odt), var.to(device=Y.device, dtype=odt)


class _NLMLPair(torch.autograd.Function):
    """nlml() for a composed kernel: the composition is assembled straight into the factorisation's buffer and its gradient tile
    reads G once (ffgp_problem.tree / ffgp_grads.g_pair)."""

    @staticmethod
    def forward(ctx, X, Y, op, meta, diag_add, diag_vec, add_mat, add_all, mean_jitter, variant, pi_const, slot, defer, rec,
                *tensors):
        dev = _device_of(X, Y, tensors[0])
        h = _lib.handle(dev.index, slot)
        _lib.bind_stream(h, dev.index)
        keep = []
        p, (n, D, d) = _problem(dev, X, Y, None, None, diag_add, diag_vec, add_mat, add_all, mean_jitter, NEG_INF, variant,
                                pi_const, keep)
        tree = _pair_descs(dev, D, meta, tensors, keep, op)
        p.tree = C.pointer(tree)
        keep.append(tree)
        # positions: Y 1, diag_add 4, diag_vec 5, the leaves' tensors 14 ...
        need_Y, need_da, need_dv = (rec and bool(ctx.needs_input_grad[i]) for i in (1, 4, 5))
        needs = [rec and bool(f) for f in ctx.needs_input_grad[14:14 + 4 * len(meta)]]
        out = torch.empty((), dtype=torch.float64, device=dev)
        g = None
        grads = {}
        if need_Y or need_da or need_dv or any(needs):
            g = Grads()
            if need_Y:
                grads["Y"] = torch.empty((n, d), dtype=torch.float64, device=dev)
                g.g_Y_dev = _ptr(grads["Y"])
            if need_da:
                grads["diag_add"] = torch.empty((1,), dtype=torch.float64, device=dev)
                g.g_diag_add_dev = _ptr(grads["diag_add"])
            if need_dv:
                grads["diag_vec"] = torch.empty((n,), dtype=torch.float64, device=dev)
                g.g_diag_vec_dev = _ptr(grads["diag_vec"])
            garr, bufs = _pair_grad_buffers(dev, D, needs)
            if garr is not None:
                g.g_pair = garr
                grads["_pair"] = bufs
                keep.append(garr)
        gref = C.byref(g) if g is not None else None
        if defer:
            check(lib.ffgp_nlml_fused_async(h, C.byref(p), _ptr(out), gref), "ffgp_nlml_fused_async")
            _pending.setdefault((dev.index, slot), []).append(keep)
        else:
            rc = check(lib.ffgp_nlml_fused(h, C.byref(p), _ptr(out), gref), "ffgp_nlml_fused")
            if rc > 0:
                _raise_not_pd(rc, "linalg.cholesky")
        ctx.grads, ctx.needs, ctx.D = grads, needs, D
        ctx.meta = [(t.shape, t.dtype, t.device) if isinstance(t, torch.Tensor) else None for t in (Y, diag_add, diag_vec)]
        ctx.metas = [(t.shape, t.dtype, t.device) if isinstance(t, torch.Tensor) else None for t in tensors]
        return out.to(device=Y.device, dtype=Y.dtype if Y.dtype.is_floating_point else torch.float64)

    @staticmethod
    def backward(ctx, gout):
        def fin(key, idx):
            if key not in ctx.grads or ctx.meta[idx] is None:
                return None
            shape, dtype, device = ctx.meta[idx]
            t = ctx.grads[key] * gout.to(device=ctx.grads[key].device, dtype=torch.float64)
            if key == "diag_vec" and len(shape) == 2:
                t = torch.diag_embed(t)
            return t.reshape(shape).to(device=device, dtype=dtype)

        pair = _pair_grads_out(ctx.grads.get("_pair"), ctx.D, ctx.needs, ctx.metas, scale=gout)
        return (None, fin("Y", 0), None, None, fin("diag_add", 1), fin("diag_vec", 2)) + (None,) * 8 + tuple(pair)


def pair_inputs_plain(x_train, *extras):
    """True when the fused pair likelihood may be used: `_NLMLPair.backward` returns gradients for Y, diag_add, diag_vec and the
    kernel parameters only, so a caller with learnable / latent inputs (x_train.requires_grad) or a gradient-carrying y_var
    matrix must take the composed path (kernel_on_device -> add_diagonal -> gaussian_nll_from_cov), which differentiates
    through both."""
    if not torch.is_grad_enabled():
        return True
    return not any(isinstance(t, torch.Tensor) and t.requires_grad for t in (x_train,) + extras)


def nlml_pair(X, Y, descs, op, diag_add=None, diag_vec=None, add_mat=None, add_all=0.0, mean_jitter=0.0, variant=FFGP_LL_V1,
              pi_const=PI_TRUNC, slot=None, defer=False):
    """nlml() for a composed kernel given as descriptor dicts and `op` (see kernel._Pair.pair and `kernel_pair`)."""
    if slot is None:
        slot = _lib.current_slot()
    meta, tensors = _pair_split(descs)
    _tree_spec(op, len(meta))
    return _NLMLPair.apply(X, Y, op, meta, diag_add, diag_vec, add_mat, add_all, mean_jitter, variant, pi_const, slot, defer,
                           torch.is_grad_enabled(), *tensors)
