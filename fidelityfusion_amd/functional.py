"""Torch-facing operators over the libffgp C ABI (device memory, streams and autograd glue only) -- the one import the drop-in
modules use (`from . import functional as F`).

Everything numerical happens in the HIP library; these modules move pointers.  Inputs may live on the CPU (the reference's 2024
API is CPU-only, `torch.eye` without a device at GaussianProcess/cigp_v10.py:31,57): they are copied to the MI355X, results come
back on the input's device and dtype.  Arithmetic is fp64 on the device whatever the input dtype.

Round 4 split the former 1 600-line module by subject; this file re-exports every name:
    nlml.py       the fused likelihood calls (effective / raw parameters, batches, composed kernels), fused posterior
    linalg.py     kernel matrices, Cholesky with passenger rows, conditional Gaussian, GEMM, subset matching, small eigh
    posterior.py  the kept factor (`Posterior`, `PosteriorCache`)
    blocks.py     several blocks in flight on one GPU (`concurrent_blocks`, `threaded_blocks`)
    kdesc.py, _common.py   descriptor packing, shared plumbing
Module state lives where it is used: set `nlml_module.DEFER_RAW_ERRORS` through `defer_raw_errors(True / False)`.
"""
from . import _lib, blocks, kdesc, linalg, posterior
from . import nlml as nlml_module     # (the name `nlml` is the likelihood FUNCTION below, as it always was)
from ._common import NEG_INF, _check_same_D, _check_xy, _dev, _device_of, _ptr, _raise_not_pd, _split_kfun, _weights
from ._lib import FFGP_LL_V1, FFGP_LL_V2, FFGP_VAR_DIAG, FFGP_VAR_FULL, PI_TRUNC, Grads, KDesc, KDescGrads, Problem, check, lib
from .blocks import (_mark_used_on, _pending, _slot_args, _threaded_blocks_run, concurrent_blocks, configure_queues, reserve_block_streams,
                     threaded_blocks, wait)
from .kdesc import (FFGP_KFUN_LINEAR, FFGP_KOP_PRODUCT, FFGP_KOP_SUM, FFGP_TREE_BALANCED, FFGP_TREE_CHAIN, _PAIR_KEYS, _pair_descs,
                    _pair_grad_buffers, _pair_grads_out, _pair_split, _tree_spec)
from .linalg import (_CondGauss, _EighSmall, _GaussNLLFromCov, _gemm, _KernelMatrix, _KernelPair, _MatmulNT, _pad_ld, _syevj_small,
                     add_diagonal, cholesky, cholesky_with_rows, conditional_gaussian, eigh_small, gaussian_ll_v2,
                     gaussian_nll_from_cov, kernel_matrix, kernel_on_device, kernel_pair, matmul_nt, rows_in)
from .nlml import (RAGGED_CHAIN_MAX_N, SMALL_BATCH_MAX_d, SMALL_BATCH_MAX_D, SMALL_BATCH_MAX_N, _NLML, _NLMLPair, _NLMLRaw, _NLMLRawMany, _problem, _raw_pending,
                   _settle_raw, many_batchable, nlml, nlml_many, nlml_pair, nlml_raw, nlml_raw_many, pair_inputs_plain, predict, raw_many_ok,
                   raw_ok, raw_path)
from .posterior import Posterior, PosteriorCache, PosteriorCacheMixin, _PosteriorQuery

SUBMODULES = (nlml_module, linalg, posterior, blocks)     # every module that calls the library through its own `lib` name


def defer_raw_errors(on):
    """Opt in to / out of the deferred status of GPU-resident training steps (see nlml.py, DEFER_RAW_ERRORS); returns the previous
    setting."""
    prev = nlml_module.DEFER_RAW_ERRORS
    nlml_module.DEFER_RAW_ERRORS = bool(on)
    return prev


class patched_lib:
    """(tests) every submodule calls the library through its own `lib` name: this context manager points all of them at `obj` (a
    spy that forwards to `_lib.lib`) and restores them afterwards"""

    def __init__(self, obj):
        self.obj = obj

    def __enter__(self):
        self.saved = [(m, m.lib) for m in SUBMODULES]
        for m, _ in self.saved:
            m.lib = self.obj
        return self.obj

    def __exit__(self, *exc):
        for m, old in self.saved:
            m.lib = old
        return False
