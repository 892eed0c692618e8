"""K Adam training steps per library call (`train_many`): the reference's hot loop -- per fidelity 100-1000 iterations of
`optimizer.zero_grad(); loss = -model.negative_log_likelihood(x, y); loss.backward(); optimizer.step()`
(FidelityFusion_Models/ResGP.py:78-112, AR_autoRegression.py:95-137, GaussianProcess/cigp_v10.py:92-104) at N = 16 ... 500 -- runs
as ONE call of ffgp_train_raw: likelihood, closed-form gradients and torch.optim.Adam's update of the raw parameters on the device,
the loss trace returned, the factorisation status read once.  Through the drop-in modules a step costs 0.28-0.32 ms at N <= 128 (one
Python round trip, one autograd graph, one status read-back); here it costs its GPU work.
"""
import ctypes as C

import torch

from . import _lib
from ._common import _raise_not_pd
from ._lib import FFGP_LL_V1, Problem, check, lib

TRAIN_MAX_MODELS = 16      # models per ffgp_train_raw call (include/ffgp.h); longer lists are trained in chunks
TRAIN_THREADS = 4          # host threads (handle + stream each) that train the larger models of one call side by side


class AdamState:
    """torch.optim.Adam's per-parameter state of the models of one `train_many` chunk, kept on the device between calls:
    buf[f] = [exp_avg (nw + 2) | exp_avg_sq (nw + 2)] in the order length scales, signal variance, log_beta; `step` = updates taken."""

    def __init__(self, buf, stride, step=0):
        self.buf, self.stride, self.step = buf, stride, step


def _jitter_and_pi():
    from .cigp_v10 import JITTER, PI
    return JITTER, PI


def _eligible(model, x, y):
    """the model's links when ffgp_train_raw can train it: a `cigp` with a library kernel whose raw-parameter path applies (everything
    on one GPU in fp64, no learnable profile parameter, no gradient-carrying inputs), all three parameters trainable"""
    from . import functional as F
    if isinstance(y, list):
        y, y_var = y[0], y[1]
    else:
        y_var = None
    if not (hasattr(model, "kernel") and hasattr(model, "log_beta")):
        return None
    if y_var is not None and not F.raw_ok(y_var):
        return None
    with torch.enable_grad():
        lk = F.raw_path(model.kernel, x, y, model.log_beta)
    if lk is None or isinstance(lk.get("kparam"), torch.Tensor) or y.requires_grad:
        return None
    if not (lk["w"].requires_grad and lk["amp"].requires_grad and model.log_beta.requires_grad):
        return None
    if {id(q) for q in model.parameters()} != {id(lk["w"]), id(lk["amp"]), id(model.log_beta)}:
        return None      # a kernel with further learnable parameters (MaternKernel's rho is a constant; RQ's alpha is not)
    return lk, y, y_var


def _reference_loop(models, xs, ys, steps, lr, betas, eps, opts):
    """the reference's loop itself (one torch.optim.Adam per model): models that the fused call cannot train"""
    trace = torch.empty((len(models), steps), dtype=torch.float64)
    for f, (m, x, y) in enumerate(zip(models, xs, ys)):
        if opts[f] is None:
            opts[f] = torch.optim.Adam(m.parameters(), lr=lr, betas=betas, eps=eps)
        for k in range(steps):
            opts[f].zero_grad()
            loss = -m.negative_log_likelihood(x, y)
            loss.backward()
            opts[f].step()
            trace[f, k] = float(loss.detach())
    return trace


def train_many(models, xs, ys, steps, lr=1e-2, betas=(0.9, 0.999), eps=1e-8, state=None):
    """`steps` Adam iterations on every model of `models` (independent `cigp` models, `xs[f]`, `ys[f]` their training data; y may be
    `[y, y_var]`), each exactly the reference's iteration (FidelityFusion_Models/ResGP.py:82-88): loss = -negative_log_likelihood,
    gradients of the three raw parameters, torch.optim.Adam(lr, betas, eps) update.  Returns `(trace, state)`:
    trace [F, steps] (float64, on the models' device) = the loss of model f at step k BEFORE that step's update -- what the reference
    prints -- and `state`, to be passed back in to continue the same optimisers (`state=None`: fresh optimisers).
    Models on one GPU in fp64 with a library kernel are trained by ffgp_train_raw, up to 16 per call (small models -- N <= 128, D,
    d <= 16 -- take ONE kernel launch for ALL their steps: csrc/train.hip, a persistent workgroup per model); anything else runs the
    reference's loop through the drop-in modules.
    A Sigma that is not positive definite raises torch.linalg.LinAlgError as the reference's loop would; the failing model's parameters
    then hold the values they had when that step began (the other small models of the same call have completed their steps; the
    optimiser state of a failed call is not advanced).
    Two differences from running the loop yourself: the parameters are updated in place on the device and are left WITHOUT `.grad`
    (there is no autograd pass), and the Adam moments live in the returned `state`, not in a `torch.optim.Adam` -- there is no
    `optimizer.state_dict()` to checkpoint; keep `state` (and the step count inside it) instead."""
    models, xs, ys = list(models), list(xs), list(ys)
    nF = len(models)
    if not (nF == len(xs) == len(ys)) or steps <= 0:
        raise ValueError("train_many: models, xs, ys must have one length and steps must be positive")
    elig = [_eligible(m, x, y) for m, x, y in zip(models, xs, ys)]
    fused = all(e is not None for e in elig) and len({x.device for x in xs}) == 1
    if state is None:
        state = {"fused": fused, "chunks": {}, "opts": [None] * nF}
    if not fused or not state["fused"]:
        if state["fused"]:
            raise ValueError("train_many: this state belongs to fused training; the models no longer qualify for it")
        return _reference_loop(models, xs, ys, steps, lr, betas, eps, state["opts"]), state
    JITTER, PI = _jitter_and_pi()
    from . import functional as F
    from .blocks import threaded_blocks
    dev = xs[0].device
    trace = torch.empty((nF, steps), dtype=torch.float64, device=dev)
    opt = _lib.Adam(float(lr), float(betas[0]), float(betas[1]), float(eps))

    def describe(f):
        lk, y, y_var = elig[f]
        x, m = xs[f], models[f]
        n, D = x.shape
        p = Problem()
        p.n, p.D, p.d = n, D, y.shape[1]
        p.X_dev, p.Y_dev, p.w_dev, p.amp_dev = x.data_ptr(), y.data_ptr(), lk["w"].data_ptr(), lk["amp"].data_ptr()
        p.diag_add_dev = m.log_beta.data_ptr()
        p.clamp_min = lk["clamp"]
        if y_var is not None:
            p.diag_stride = y_var.shape[1] + 1 if y_var.dim() == 2 else 1
            p.diag_vec_dev = y_var.data_ptr()
        p.ll_variant, p.pi_const = FFGP_LL_V1, PI
        kp = lk.get("kparam")
        p.kfun, p.kparam = lk["kfun"], (1.0 if kp is None else float(kp))
        ll = _lib.Links()
        ll.w_link, ll.w_c, ll.w_broadcast = lk["w_link"], lk["w_c"], 1 if lk["w"].numel() == 1 and D > 1 else 0
        ll.amp_link, ll.amp_c = lk["amp_link"], 0.0
        ll.dadd_link, ll.dadd_c = _lib.LINK_EXP_NEG, JITTER
        ll.out_scale = 1.0          # the value is the loss the reference minimises: -negative_log_likelihood = +nll
        return p, ll, lk["w"].numel()

    def run(idx):
        """one ffgp_train_raw call for the models `idx` (<= 16) on the calling thread's handle and stream; returns the status"""
        h = _lib.handle(dev.index)
        _lib.bind_stream(h, dev.index)
        P = (Problem * len(idx))()
        L = (_lib.Links * len(idx))()
        nws = []
        for j, f in enumerate(idx):
            P[j], L[j], nw = describe(f)
            nws.append(nw)
        stride = 2 * (max(nws) + 2)
        key = tuple(idx)
        st = state["chunks"].get(key)
        if st is None or st.stride != stride or st.buf.device != dev:
            st = AdamState(torch.zeros((len(idx), stride), dtype=torch.float64, device=dev), stride)
            state["chunks"][key] = st
        # the chunk's rows of the trace: contiguous when the models are consecutive, else through a staging block
        contiguous = list(idx) == list(range(idx[0], idx[0] + len(idx)))
        tr = trace[idx[0]:idx[0] + len(idx)] if contiguous else torch.empty((len(idx), steps), dtype=torch.float64, device=dev)
        rc = check(lib.ffgp_train_raw(h, len(idx), P, L, int(steps), C.byref(opt), st.buf.data_ptr(), stride, int(st.step),
                                      tr.data_ptr(), tr.stride(0)), "ffgp_train_raw")
        if rc == 0:      # (a call that failed leaves its optimisers where they were: the caller sees LinAlgError)
            st.step += steps
        if not contiguous:
            trace[list(idx)] = tr
        return rc

    # small models (one workgroup each: ONE launch per step for up to 16 of them) go together; every larger model is a call of its
    # own -- and, when there are several, they train SIDE BY SIDE from host threads with a handle and a stream each
    # (blocks.threaded_blocks: the calls only enqueue and wait once, ctypes drops the GIL inside them), so that one model's
    # latency-bound chain of small kernels runs in the gaps of the others'
    shapes = [(xs[f].shape[0], xs[f].shape[1], elig[f][1].shape[1]) for f in range(nF)]
    small = [f for f in range(nF) if shapes[f][0] <= F.SMALL_BATCH_MAX_N and shapes[f][1] <= F.SMALL_BATCH_MAX_D
             and shapes[f][2] <= F.SMALL_BATCH_MAX_d]
    large = [f for f in range(nF) if f not in set(small)]
    if len(small) == 1:      # (a lone small model gains nothing from the batch kernel: its own call folds the tail launches)
        large, small = sorted(large + small), []
    rcs = []
    for c0 in range(0, len(small), TRAIN_MAX_MODELS):
        rcs.append((small[c0:c0 + TRAIN_MAX_MODELS], run(small[c0:c0 + TRAIN_MAX_MODELS])))
    if len(large) >= 2 and _lib.current_slot() == 0:
        outs = threaded_blocks([(lambda f=f: run([f])) for f in large], nslots=min(TRAIN_THREADS, len(large)), device_index=dev.index)
        rcs += [([f], rc) for f, rc in zip(large, outs)]
    else:
        rcs += [([f], run([f])) for f in large]
    # the library wrote the parameters behind autograd's back: bump their version counters (cached posteriors key on them)
    bump = getattr(torch.autograd.graph, "increment_version", None)      # (torch >= 2.1: no kernel; else an in-place no-op add)
    with torch.no_grad():
        for m in models:
            for q in m.parameters():
                if bump is not None:
                    bump(q)
                else:
                    q.add_(0.0)
    for idx, rc in rcs:
        if rc > 0:
            _raise_not_pd(rc, "linalg.cholesky (train_many, model%s %s)" % ("s" if len(idx) > 1 else "", ", ".join(map(str, idx))))
    return trace, state
