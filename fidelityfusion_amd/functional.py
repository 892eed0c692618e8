"""Torch-facing operators over the libffgp C ABI (device memory, streams and autograd glue only).

Everything numerical happens in the HIP library; this module moves pointers.  Inputs may live on the CPU (the
reference's 2024 API is CPU-only, `torch.eye` without a device at GaussianProcess/cigp_v10.py:31,57): they are
copied to the MI355X, results come back on the input's device and dtype.  Arithmetic is fp64 on the device
whatever the input dtype.
"""
import ctypes as C
import math
import threading
import weakref

import torch

from . import _lib
from ._lib import FFGP_LL_V1, FFGP_LL_V2, FFGP_VAR_DIAG, FFGP_VAR_FULL, PI_TRUNC, Grads, KDesc, KDescGrads, Problem, check, lib

NEG_INF = float("-inf")


def _device_of(*tensors):
    for t in tensors:
        if isinstance(t, torch.Tensor) and t.is_cuda:
            return t.device
    if not torch.cuda.is_available():
        raise _lib.FFGPError("fidelityfusion_amd needs an MI355X (gfx950) GPU; there is no CPU path")
    return torch.device("cuda", torch.cuda.current_device())


def _dev(t, dev):
    """fp64 contiguous copy/view of t on the compute device (detached)."""
    return t.detach().to(device=dev, dtype=torch.float64).contiguous()


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _check_xy(X, Y=None, what="x_train"):
    """The library reads raw device pointers: every size it derives them from is checked here first (the reference fails
    with a broadcast / solve error on the same mistakes; an unchecked mismatch would be an out-of-bounds device read)."""
    if X.dim() != 2:
        raise ValueError("%s must be 2-D [N, D], got shape %s" % (what, tuple(X.shape)))
    if Y is not None:
        if Y.dim() != 2:
            raise ValueError("y_train must be 2-D [N, d], got shape %s" % (tuple(Y.shape),))
        if Y.shape[0] != X.shape[0]:
            raise ValueError("y_train has %d rows for %d training inputs" % (Y.shape[0], X.shape[0]))


def _check_same_D(a, b, what="x_test"):
    if b.dim() != 2 or b.shape[1] != a.shape[1]:
        raise ValueError("%s must be [*, %d] like the training inputs, got shape %s" % (what, a.shape[1], tuple(b.shape)))


def _weights(w, D, dev):
    """[D] inverse length scales on the device: one value is broadcast over the input dimensions, D values are taken as
    they are, anything else (e.g. ARDKernel(input_dim=3) on 5-D inputs) is the caller's mistake."""
    wd = _dev(w.reshape(-1), dev)
    if wd.numel() == 1 and D > 1:
        wd = wd.expand(D).contiguous()
    if wd.numel() != D:
        raise ValueError("the kernel has %d length scales but the inputs have %d dimensions" % (wd.numel(), D))
    return wd


def _raise_not_pd(rc, what):
    raise torch.linalg.LinAlgError(
        "%s: The factorization could not be completed because the input is not positive-definite "
        "(the leading minor of order %d is not positive-definite)." % (what, rc))


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
    return _NLMLRawMany.apply(tuple(metas), torch.is_grad_enabled(), *tensors)


def nlml_many(Xs, Ys, ws, amps, diag_adds, clamp=NEG_INF, pi_const=PI_TRUNC):
    """[nlml(X, Y, w, amp, diag_add=dadd, clamp=clamp) for ...] as one tensor [F] through ONE factorisation chain
    (ffgp_nlml_fused_batch): F >= 2 blocks of one shape (the same n > 128 and d), everything on one GPU in fp64, effective
    parameters (w [D], amp [1], diag_add [1] per block; squared-exponential profile).  Gradients flow to Y, w, amp and diag_add.
    The per-fidelity blocks of one rank in the sharded workloads (bench.py `cigar4`, `gar8`) are such a batch."""
    items = []
    for X, Y, w, amp, dadd in zip(Xs, Ys, ws, amps, diag_adds):
        if not raw_ok(X, Y, w, amp, dadd) or w.numel() != X.shape[1]:
            raise ValueError("nlml_many: every tensor must live on one GPU in fp64, contiguous, with w of length D")
        lk = {"w": w, "w_link": _lib.LINK_ID, "w_c": 0.0, "amp": amp, "amp_link": _lib.LINK_ID, "clamp": clamp, "kfun": 0}
        items.append({"X": X, "Y": Y, "lk": lk, "rdadd": dadd, "dadd_link": _lib.LINK_ID, "dadd_c": 0.0, "pi_const": pi_const})
    if not many_batchable([(it["X"].shape[0], it["Y"].shape[1]) for it in items]):
        raise ValueError("nlml_many: the blocks must share one shape (n, d) with n > %d" % SMALL_BATCH_MAX_N)
    return nlml_raw_many(items)


def raw_many_ok(kernel, x_train, y_train, *others):
    """`raw_path` + the limits of the batched calls: up to SMALL_BATCH_MAX_N points the one-workgroup batch (any mix of shapes,
    D, d <= 16); beyond that the shared-chain batch, which needs every member to have the SAME (n, d) -- checked by the caller
    (`many_batchable`) -- and D <= 128"""
    lk = raw_path(kernel, x_train, y_train, *others)
    if lk is None or isinstance(lk.get("kparam"), torch.Tensor):
        return None
    if x_train.shape[0] > SMALL_BATCH_MAX_N:
        return lk if x_train.shape[1] <= 128 else None
    if x_train.shape[1] > SMALL_BATCH_MAX_D or y_train.shape[1] > SMALL_BATCH_MAX_d:
        return None
    return lk


def many_batchable(shapes):
    """shapes: [(n, d)] of the members.  One library call serves them when they are all small (n <= SMALL_BATCH_MAX_N: one
    workgroup each), or at least two blocks of ONE larger shape (ffgp_nlml_fused_batch: they share one factorisation chain)."""
    if all(n <= SMALL_BATCH_MAX_N for n, _ in shapes):
        return True
    return len(shapes) >= 2 and len(set(shapes)) == 1 and shapes[0][0] > SMALL_BATCH_MAX_N


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


_pending = {}   # (device, slot) -> staging tensors of enqueued-but-not-waited calls (kept alive until wait)


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


def _split_kfun(kfun):
    """(id, float | tensor) -> ((id, float), tensor | None): a tensor parameter is differentiated (g_kparam)."""
    if isinstance(kfun[1], torch.Tensor):
        return (int(kfun[0]), float(kfun[1].detach())), kfun[1]
    return (int(kfun[0]), float(kfun[1])), None


def wait(slot=None, device_index=None):
    """Collect a deferred call: synchronises that slot's stream, raises LinAlgError if its Sigma was not PD."""
    if slot is None:
        slot = _lib.current_slot()
    if device_index is None:
        device_index = torch.cuda.current_device()
    h = _lib.handle(device_index, slot)
    rc = check(lib.ffgp_wait(h), "ffgp_wait")
    _pending.pop((device_index, slot), None)
    if rc > 0:
        _raise_not_pd(rc, "linalg.cholesky")


class concurrent_blocks:
    """Run independent GP blocks concurrently on one GPU:

        with concurrent_blocks(nslots=2) as cb:
            for f, m in enumerate(models):
                with cb.slot(f):                                  # own handle, own stream
                    losses[f] = -m.negative_log_likelihood(x[f], y[f])
        # on exit every slot has been waited for (LinAlgError raised if any block failed)

    The likelihood modules pick the active slot up from this context.

    lookahead=False: the slots' factorisations run WITHOUT their own look-ahead side stream.  Look-ahead hides one block's
    panel chain under its own trailing update; with several blocks in flight the other blocks' updates do that already, and
    the side streams' high-priority kernels only get in each other's way (measured, 4 blocks of N = 8192, d = 1024 on one
    MI355X: 27.9 ms with look-ahead in 3 slots, 23.5 ms without in 2 -- tools/c4_step.py, bench.py --workload cigar4)."""
    active = None

    def __init__(self, nslots=2, device_index=None, lookahead=False):
        self.nslots = nslots
        self.lookahead = bool(lookahead)
        self.device_index = torch.cuda.current_device() if device_index is None else device_index
        self.streams = [torch.cuda.Stream(self.device_index) for _ in range(nslots)]
        self.used = set()
        self.cur = None

    def __enter__(self):
        concurrent_blocks.active = self
        self.origin = torch.cuda.current_stream(self.device_index)
        for s in self.streams:
            s.wait_stream(self.origin)
        return self

    def slot(self, i):
        cb = self

        class _Slot:
            def __enter__(self_inner):
                cb.cur = 1 + (i % cb.nslots)          # slot 0 stays the synchronous default handle
                if cb.cur not in cb.used:   # (restored in concurrent_blocks.__exit__: the slot handles are process-wide)
                    _lib.set_option_handle(_lib.handle(cb.device_index, cb.cur), "lookahead", 1.0 if cb.lookahead else 0.0)
                cb.used.add(cb.cur)
                self_inner.ctx = torch.cuda.stream(cb.streams[cb.cur - 1])
                self_inner.ctx.__enter__()

            def __exit__(self_inner, *exc):
                self_inner.ctx.__exit__(*exc)
                cb.cur = None
        return _Slot()

    def __exit__(self, *exc):
        concurrent_blocks.active = None
        err = None
        for sl in sorted(self.used):
            try:
                with torch.cuda.stream(self.streams[sl - 1]):
                    wait(sl, self.device_index)
            except torch.linalg.LinAlgError as e:   # keep draining the other slots
                err = e
        for s in self.streams:
            self.origin.wait_stream(s)
        for sl in sorted(self.used):   # the library default (look-ahead on) for whoever uses that slot's handle next
            _lib.set_option_handle(_lib.handle(self.device_index, sl), "lookahead", 1.0)
        if err is not None and exc[0] is None:
            raise err
        return False


reserve_block_streams = _lib.reserve_block_streams   # (explicitly via _lib.configure_queues(), or by the first threaded_blocks of a GPU)
configure_queues = _lib.configure_queues


def threaded_blocks(fns, nslots=2, device_index=None):
    """Run independent blocks -- callables without arguments -- concurrently on one GPU from `nslots` host threads and return
    their results in order.  Worker k runs blocks k, k + nslots, ... on its own stream with handle slot 1 + k as the thread's
    current slot (`_lib.thread_slot`), so everything a block calls lands on that handle.

    This is the form of `concurrent_blocks` for blocks whose library calls wait on the host: `ffgp_syevd` synchronises after its
    bulge chasing (the watchdog word), so a single thread cannot put a second HOGP block under the first one's 80 ms of
    latency-bound chase -- two threads can (ctypes drops the GIL inside the library; one thread per handle is the library's
    threading rule, include/ffgp.h).  The caller's stream is waited for before the workers start and waits for theirs at the
    end; the first exception of any block is raised after every worker has finished.  Grad mode is the caller's.  Tensors among the
    results (also inside lists / tuples / dicts) are marked as used by the caller's stream (`record_stream`): they were allocated on
    a worker's.  One call at a time per GPU (the worker slots are process-wide: a second caller waits); a call from INSIDE a worker
    runs its blocks inline on that worker's slot."""
    fns = list(fns)
    if device_index is None:
        device_index = torch.cuda.current_device()
    nslots = max(1, min(int(nslots), len(fns)))
    results, errors = [None] * len(fns), []
    if nslots <= 1 or _lib.current_slot() != 0:
        return [fn() for fn in fns]
    with _threaded_locks_guard:
        gate = _threaded_locks.setdefault(device_index, threading.Lock())
    with gate:
        return _threaded_blocks_run(fns, nslots, device_index, results, errors)


_threaded_locks = {}
_threaded_locks_guard = threading.Lock()


def _mark_used_on(obj, stream):
    if isinstance(obj, torch.Tensor):
        if obj.is_cuda:
            obj.record_stream(stream)
    elif isinstance(obj, (list, tuple)):
        for o in obj:
            _mark_used_on(o, stream)
    elif isinstance(obj, dict):
        for o in obj.values():
            _mark_used_on(o, stream)


def _threaded_blocks_run(fns, nslots, device_index, results, errors):
    origin = torch.cuda.current_stream(device_index)
    _lib.reserve_block_streams(device_index, max(4, nslots))      # (idempotent; best done up front: _lib.configure_queues)
    # one stream per worker slot for the life of the process: the caching allocator pools memory per stream (fresh streams would send
    # every step's temporaries back to hipMalloc) and the slot's handle stays bound to one stream
    streams = [_lib.block_stream(device_index, k) for k in range(nslots)]
    for st in streams:
        st.wait_stream(origin)
    grad = torch.is_grad_enabled()

    def work(k):
        try:
            torch.cuda.set_device(device_index)
            with torch.cuda.stream(streams[k]), _lib.thread_slot(1 + k), torch.set_grad_enabled(grad):
                for i in range(k, len(fns), nslots):
                    results[i] = fns[i]()
        except BaseException as e:   # noqa: BLE001  (re-raised in the caller's thread)
            errors.append(e)

    threads = [threading.Thread(target=work, args=(k,), name="ffgp-block-%d" % k) for k in range(nslots)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for st in streams:
        origin.wait_stream(st)
    if errors:
        raise errors[0]
    _mark_used_on(results, origin)
    return results


def _slot_args():
    cb = concurrent_blocks.active
    if cb is not None and cb.cur is not None:
        return dict(slot=cb.cur, defer=True)
    return dict(slot=_lib.current_slot(), defer=False)


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


# ----------------------------------------------------------------------------------------------------------------------
# composed kernels: K = k_a (+ | x) k_b -- and nested compositions of up to four leaves -- as descriptors evaluated in one tile
# pass (ffgp_assemble_tree / ffgp_kernel_grad_tree / ffgp_kernel_input_weights_tree and ffgp_problem.tree) -- SumKernel /
# ProductKernel of GaussianProcess/kernel.py:172-236
# ----------------------------------------------------------------------------------------------------------------------
FFGP_KFUN_LINEAR = 5
FFGP_KOP_SUM, FFGP_KOP_PRODUCT = 0, 1
FFGP_TREE_CHAIN, FFGP_TREE_BALANCED = 0, 1
_PAIR_KEYS = ("w", "amp", "kparam", "center")


def _tree_spec(op, nl):
    """`op`: one FFGP_KOP_* for two leaves, or (shape, (op0, op1[, op2])) for the canonical nested forms of include/ffgp.h"""
    if isinstance(op, int):
        if nl != 2:
            raise ValueError("a single operator composes exactly two kernels")
        return FFGP_TREE_CHAIN, (op,)
    shape, ops = op
    ops = tuple(int(o) for o in ops)
    if not 2 <= nl <= 4 or len(ops) != nl - 1 or any(o not in (FFGP_KOP_SUM, FFGP_KOP_PRODUCT) for o in ops):
        raise ValueError("a composed kernel takes 2-4 leaves and one operator per node")
    return int(shape), ops


def _pair_split(descs):
    """descriptor dicts {kfun, w, amp, clamp, kparam, center} -> (static meta, the 4 tensor-or-None autograd inputs of each)"""
    meta, tensors = [], []
    for dsc in descs:
        kp = dsc.get("kparam", 1.0)
        kp_t = kp if isinstance(kp, torch.Tensor) else None
        meta.append((int(dsc["kfun"]), float(dsc.get("clamp", NEG_INF)), float(kp.detach()) if kp_t is not None else float(kp)))
        tensors += [dsc["w"], dsc["amp"], kp_t, dsc.get("center")]
    return tuple(meta), tensors


def _pair_descs(dev, D, meta, tensors, keep, op):
    """-> KTree (by value; its leaf array and the staged device tensors -- (w, amp, center | None) per leaf, first entry of
    `keep` -- are appended to `keep`)"""
    nl = len(meta)
    shape, ops = _tree_spec(op, nl)
    arr = (KDesc * nl)()
    staged = []
    keep.append(staged)
    for e in range(nl):
        w, amp, _, cen = tensors[4 * e:4 * e + 4]
        wd = _weights(w, D, dev)
        ad = _dev(amp.reshape(-1)[:1], dev)
        arr[e].kfun, arr[e].clamp_min, arr[e].kparam = meta[e]
        arr[e].w_dev, arr[e].amp_dev = _ptr(wd), _ptr(ad)
        cd = None
        if cen is not None and meta[e][0] == FFGP_KFUN_LINEAR:
            cd = _weights(cen, D, dev)
            arr[e].center_dev = _ptr(cd)
        staged.append((wd, ad, cd))
    t = _lib.KTree()
    t.n_leaves, t.shape, t.leaf = nl, shape, arr
    for i, o in enumerate(ops):
        t.op[i] = o
    keep.append(arr)
    return t


def _pair_grad_buffers(dev, D, needs):
    """needs: 4 flags per leaf in the order of the tensor inputs -> (KDescGrads[nl] | None, the [nl, w (D) | center (D) | amp | kparam] buffer)"""
    if not any(needs):
        return None, None
    nl = len(needs) // 4
    arr = (KDescGrads * nl)()
    bufs = torch.empty((nl, 2 * D + 2), dtype=torch.float64, device=dev)
    step = bufs.element_size()
    for e in range(nl):
        base = bufs[e].data_ptr()
        nw, na, nk, nc = needs[4 * e:4 * e + 4]
        if nw:
            arr[e].g_w_dev = C.c_void_p(base)
        if nc:
            arr[e].g_center_dev = C.c_void_p(base + D * step)
        if na:
            arr[e].g_amp_dev = C.c_void_p(base + 2 * D * step)
        if nk:
            arr[e].g_kparam_dev = C.c_void_p(base + (2 * D + 1) * step)
    return arr, bufs


def _pair_grads_out(bufs, D, needs, metas, scale=None):
    """the 4 gradient outputs per leaf (None where not needed) from the buffer, reshaped to the inputs' shapes / devices"""
    if bufs is None:
        return [None] * len(needs)
    if scale is not None:
        bufs = bufs * scale.to(device=bufs.device, dtype=torch.float64)
    outs = []
    for e in range(len(needs) // 4):
        views = (bufs[e, :D], bufs[e, 2 * D:2 * D + 1], bufs[e, 2 * D + 1:2 * D + 2], bufs[e, D:2 * D])   # w, amp, kparam, center
        for k in range(4):
            m = metas[4 * e + k]
            if not needs[4 * e + k] or m is None:
                outs.append(None)
                continue
            shape, dtype, device = m
            t = views[k]
            if k in (0, 3) and math.prod(shape) == 1 and t.numel() > 1:
                t = t.sum().reshape(1)    # one value was broadcast over the D input dimensions
            outs.append(t.reshape(shape).to(device=device, dtype=dtype))
    return outs


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
