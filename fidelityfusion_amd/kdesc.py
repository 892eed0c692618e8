"""Kernel descriptors for composed kernels (Sum / Product trees of up to four radial-profile or linear leaves): packing the leaves'
parameters into the library's `ffgp_kdesc` / `ffgp_ktree` structures and unpacking their gradients (include/ffgp.h).
"""
import ctypes as C
import math

import torch

from . import _lib
from ._common import NEG_INF, _dev, _ptr, _weights
from ._lib import KDesc, KDescGrads


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
