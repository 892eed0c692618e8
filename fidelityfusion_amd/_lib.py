"""ctypes binding of libffgp.so (include/ffgp.h) -- the reference-side stub of the C ABI.

The library is built in-tree by `python __graft_entry__.py` (hipcc, gfx950) and loaded from the package
directory.  There is no CPU or torch fallback: if the shared object is missing the import fails, and if no
MI355X is visible `handle()` raises.
"""
import ctypes as C
import os
import threading

import logging

log = logging.getLogger("fidelityfusion_amd")

# PyTorch-ROCm ships its own libamdhip64; libffgp.so names the same SONAME.  torch must be imported BEFORE the
# library is dlopen'ed so that both bind to the ONE runtime already in the process -- loaded the other way round the
# process ends up with two HIP runtimes and hipGetDeviceCount() in the second one reports no device.
import torch  # noqa: F401  (device memory, streams)

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.environ.get("FFGP_LIB") or os.path.join(_HERE, "libffgp.so")   # FFGP_LIB: development builds (tools/)

FFGP_LL_V1, FFGP_LL_V2 = 1, 2
FFGP_VAR_FULL, FFGP_VAR_DIAG = 0, 1
FFGP_KFUN_SE, FFGP_KFUN_MATERN12, FFGP_KFUN_MATERN32, FFGP_KFUN_MATERN52, FFGP_KFUN_RQ = 0, 1, 2, 3, 4
PI_TRUNC = 3.1415  # GaussianProcess/cigp_v10.py:15 ; gp_computation_pack.py:17 ; MFGP_ver2023May/base_gp/cigp.py:6

ERRORS = {-1: "FFGP_ERR_ARG", -2: "FFGP_ERR_HIP", -3: "FFGP_ERR_ALLOC", -4: "FFGP_ERR_NODEVICE",
          -5: "FFGP_ERR_HANDOFF (a cross-stream hand-off of the look-ahead never arrived: a tool that serialises this process's kernels? "
              "FFGP_HANDOFF=events keeps the event pairs)"}
FFGP_ERR_ARG, FFGP_ERR_HIP, FFGP_ERR_ALLOC, FFGP_ERR_NODEVICE, FFGP_ERR_HANDOFF = -1, -2, -3, -4, -5

_dp = C.c_void_p  # device pointers travel as void*


class KDesc(C.Structure):
    """ffgp_kdesc: one part of a composed kernel (include/ffgp.h)"""
    _fields_ = [("kfun", C.c_int), ("w_dev", _dp), ("amp_dev", _dp), ("clamp_min", C.c_double), ("kparam", C.c_double),
                ("center_dev", _dp)]


class KDescGrads(C.Structure):
    _fields_ = [("g_w_dev", _dp), ("g_amp_dev", _dp), ("g_kparam_dev", _dp), ("g_center_dev", _dp)]


class KTree(C.Structure):
    """ffgp_ktree: a nested composition of 2-4 leaves in canonical form (include/ffgp.h)"""
    _fields_ = [("n_leaves", C.c_int), ("shape", C.c_int), ("op", C.c_int * 3), ("leaf", C.POINTER(KDesc))]


class Problem(C.Structure):
    _fields_ = [
        ("n", C.c_int), ("D", C.c_int), ("d", C.c_int),
        ("X_dev", _dp), ("Y_dev", _dp), ("w_dev", _dp), ("amp_dev", _dp),
        ("clamp_min", C.c_double),
        ("diag_add_dev", _dp), ("diag_vec_dev", _dp), ("diag_stride", C.c_long),
        ("add_mat_dev", _dp), ("ld_add", C.c_int),
        ("add_all", C.c_double), ("mean_jitter", C.c_double),
        ("ll_variant", C.c_int), ("pi_const", C.c_double),
        ("kfun", C.c_int), ("kparam", C.c_double),
        ("cov_dev", _dp), ("ld_cov", C.c_int),
        ("pair", C.POINTER(KDesc)), ("pair_op", C.c_int), ("tree", C.POINTER(KTree)),
    ]


class Links(C.Structure):
    """ffgp_links: elementwise maps raw parameter -> effective quantity (include/ffgp.h FFGP_LINK_*)"""
    _fields_ = [("w_link", C.c_int), ("w_c", C.c_double), ("w_broadcast", C.c_int), ("amp_link", C.c_int), ("amp_c", C.c_double),
                ("dadd_link", C.c_int), ("dadd_c", C.c_double), ("out_scale", C.c_double)]


class Adam(C.Structure):
    """ffgp_adam: torch.optim.Adam's hyper-parameters for ffgp_train_raw"""
    _fields_ = [("lr", C.c_double), ("beta1", C.c_double), ("beta2", C.c_double), ("eps", C.c_double)]


LINK_ID, LINK_INV_ABS_EPS, LINK_EXP_NEG, LINK_INV, LINK_ABS, LINK_EXP_SQ, LINK_SQUARE = range(7)


class Grads(C.Structure):
    _fields_ = [("g_w_dev", _dp), ("g_amp_dev", _dp), ("g_diag_add_dev", _dp), ("g_Y_dev", _dp),
                ("g_diag_vec_dev", _dp), ("g_cov_dev", _dp), ("ld_gcov", C.c_int), ("g_kparam_dev", _dp),
                ("g_pair", C.POINTER(KDescGrads))]


EXPORTS = {
    # name: (restype, argtypes)
    "ffgp_version": (C.c_char_p, []),
    "ffgp_has_dev_options": (C.c_int, []),
    "ffgp_graph_replays": (C.c_long, [C.c_void_p]),
    "ffgp_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "ffgp_destroy": (C.c_int, [C.c_void_p]),
    "ffgp_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ffgp_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_double]),
    "ffgp_prepare_streams": (C.c_int, [C.c_void_p]),
    "ffgp_assemble": (C.c_int, [C.c_void_p, _dp, C.c_int, _dp, C.c_int, C.c_int, _dp, _dp, C.c_double, _dp, _dp,
                                C.c_long, _dp, C.c_int, C.c_double, C.c_double, _dp, C.c_int, C.c_int, C.c_int, C.c_double]),
    "ffgp_assemble_pair": (C.c_int, [C.c_void_p, _dp, C.c_int, _dp, C.c_int, C.c_int, C.POINTER(KDesc), C.c_int, _dp, _dp, C.c_long,
                                     _dp, C.c_int, C.c_double, C.c_double, _dp, C.c_int, C.c_int]),
    "ffgp_kernel_grad_pair": (C.c_int, [C.c_void_p, _dp, C.c_int, _dp, C.c_int, C.c_int, C.POINTER(KDesc), C.c_int, _dp, C.c_int,
                                        C.POINTER(KDescGrads)]),
    "ffgp_assemble_tree": (C.c_int, [C.c_void_p, _dp, C.c_int, _dp, C.c_int, C.c_int, C.POINTER(KTree), _dp, _dp, C.c_long,
                                     _dp, C.c_int, C.c_double, C.c_double, _dp, C.c_int, C.c_int]),
    "ffgp_kernel_grad_tree": (C.c_int, [C.c_void_p, _dp, C.c_int, _dp, C.c_int, C.c_int, C.POINTER(KTree), _dp, C.c_int,
                                        C.POINTER(KDescGrads)]),
    "ffgp_kernel_input_weights_tree": (C.c_int, [C.c_void_p, _dp, C.c_int, _dp, C.c_int, C.c_int, C.POINTER(KTree), _dp, C.c_int, _dp,
                                                 C.c_int, C.c_long]),
    "ffgp_potrf": (C.c_int, [C.c_void_p, _dp, C.c_int, C.c_int]),
    "ffgp_potrf_rows": (C.c_int, [C.c_void_p, _dp, C.c_int, C.c_int, C.c_int]),
    "ffgp_kernel_grad": (C.c_int, [C.c_void_p, _dp, C.c_int, _dp, C.c_int, C.c_int, _dp, _dp, C.c_double, C.c_int, C.c_double,
                                   _dp, C.c_int, _dp, _dp, _dp]),
    "ffgp_kernel_input_weights": (C.c_int, [C.c_void_p, _dp, C.c_int, _dp, C.c_int, C.c_int, _dp, _dp, C.c_double, C.c_int,
                                            C.c_double, _dp, C.c_int, _dp, C.c_int]),
    "ffgp_syevj_small": (C.c_int, [C.c_void_p, _dp, C.c_int, C.c_int, C.c_int, C.c_long, _dp, C.c_int, C.c_long, _dp, C.c_long,
                                   C.c_int]),
    "ffgp_syevd": (C.c_int, [C.c_void_p, _dp, C.c_int, C.c_int, _dp, _dp, C.c_int]),
    "ffgp_sy2sb": (C.c_int, [C.c_void_p, _dp, C.c_int, C.c_int, _dp, _dp, C.c_int]),
    "ffgp_sb2st_reflector_doubles": (C.c_long, [C.c_int]),
    "ffgp_sb2st": (C.c_int, [C.c_void_p, _dp, C.c_int, _dp, _dp, _dp]),
    "ffgp_stedc": (C.c_int, [C.c_void_p, _dp, _dp, C.c_int, _dp, _dp, C.c_int]),
    "ffgp_ormq2": (C.c_int, [C.c_void_p, _dp, C.c_int, _dp, C.c_int, C.c_int]),
    "ffgp_ormq1": (C.c_int, [C.c_void_p, _dp, C.c_int, C.c_int, _dp, C.c_int, C.c_int]),
    "ffgp_gemm_batched": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, _dp, C.c_int, C.c_long, _dp, C.c_int, C.c_long, _dp,
                                    C.c_int, C.c_long, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int]),
    "ffgp_rows_in": (C.c_int, [C.c_void_p, _dp, C.c_int, _dp, C.c_int, C.c_int, _dp]),
    "ffgp_trtri_diag": (C.c_int, [C.c_void_p, _dp, C.c_int, C.c_int]),
    "ffgp_invalidate": (C.c_int, [C.c_void_p]),
    "ffgp_allreduce_sum": (C.c_int, [C.c_void_p, C.c_void_p, _dp, C.c_int]),
    "ffgp_trsm_lower": (C.c_int, [C.c_void_p, _dp, C.c_int, C.c_int, _dp, C.c_int, C.c_int]),
    "ffgp_trsm_lower_t": (C.c_int, [C.c_void_p, _dp, C.c_int, C.c_int, _dp, C.c_int, C.c_int]),
    "ffgp_potrs": (C.c_int, [C.c_void_p, _dp, C.c_int, C.c_int, _dp, C.c_int, C.c_int]),
    "ffgp_nll_reduce": (C.c_int, [C.c_void_p, C.c_int, _dp, C.c_int, C.c_int, _dp, C.c_int, C.c_int, C.c_double, _dp]),
    "ffgp_potri": (C.c_int, [C.c_void_p, _dp, C.c_int, C.c_int]),
    "ffgp_nlml_fused": (C.c_int, [C.c_void_p, C.POINTER(Problem), _dp, C.POINTER(Grads)]),
    "ffgp_nlml_fused_raw": (C.c_int, [C.c_void_p, C.POINTER(Problem), C.POINTER(Links), _dp, C.POINTER(Grads)]),
    "ffgp_nlml_fused_raw_async": (C.c_int, [C.c_void_p, C.POINTER(Problem), C.POINTER(Links), _dp, C.POINTER(Grads)]),
    "ffgp_nlml_fused_small_batch": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(Problem), C.POINTER(Links), _dp, C.POINTER(Grads)]),
    "ffgp_nlml_fused_small_batch_async": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(Problem), C.POINTER(Links), _dp, C.POINTER(Grads)]),
    "ffgp_nlml_fused_batch": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(Problem), C.POINTER(Links), _dp, C.POINTER(Grads), C.POINTER(C.c_int)]),
    "ffgp_train_raw": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(Problem), C.POINTER(Links), C.c_int, C.POINTER(Adam), _dp, C.c_long, C.c_long,
                                 _dp, C.c_long]),
    "ffgp_nlml_fused_async": (C.c_int, [C.c_void_p, C.POINTER(Problem), _dp, C.POINTER(Grads)]),
    "ffgp_wait": (C.c_int, [C.c_void_p]),
    "ffgp_predict": (C.c_int, [C.c_void_p, C.POINTER(Problem), _dp, C.c_int, C.c_int, C.c_double, _dp, _dp, C.c_int]),
    "ffgp_last_timings": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_char_p), C.c_int,
                                    C.POINTER(C.c_int)]),
    "ffgp_syrk_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_long),
                                  C.c_int]),
    "ffgp_mfma_f64_peak": (C.c_int, [C.c_void_p, C.POINTER(C.c_double)]),
    "ffgp_gemm": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, _dp, C.c_int, _dp, C.c_int, _dp, C.c_int,
                            C.c_int, C.c_int, C.c_int, C.c_double, C.c_double]),
}


class FFGPError(RuntimeError):
    """a negative library status; `code` is the FFGP_ERR_* value (-1 ARG, -2 HIP, -3 ALLOC, -4 NODEVICE)"""
    code = None


def _load():
    if not os.path.exists(_SO):
        raise ImportError(
            "fidelityfusion_amd: %s is missing -- build it with `python __graft_entry__.py` "
            "(hipcc --offload-arch=gfx950); there is no CPU fallback." % _SO)
    lib = C.CDLL(_SO)
    for name, (res, args) in EXPORTS.items():
        fn = getattr(lib, name)  # AttributeError here = the .so does not export what include/ffgp.h declares
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()


def has_dev_options():
    """True for the development build (`make -C fidelityfusion_amd/csrc dev`, loaded through FFGP_LIB): the switches of
    measured-and-rejected experiments are compiled in; the shipped library refuses their keys"""
    return bool(lib.ffgp_has_dev_options())

_handles = {}
_lock = threading.Lock()


def check(rc, what):
    """Negative status -> FFGPError; positive (pivot index) is returned to the caller."""
    if rc < 0:
        e = FFGPError("%s failed: %s (%d)" % (what, ERRORS.get(rc, "?"), rc))
        e.code = rc
        raise e
    return rc


_tls = threading.local()


class thread_slot:
    """Inside the block, every library call THIS host thread makes without naming a slot goes to handle slot `k` of its GPU
    (`handle(device_index)` resolves to it): how a worker thread of `functional.threaded_blocks` keeps a whole block -- its
    eigensolver, GEMMs, assemblies -- on the thread's own handle and stream.  Other threads are not affected."""

    def __init__(self, k):
        self.k = int(k)

    def __enter__(self):
        self.prev = getattr(_tls, "slot", 0)
        _tls.slot = self.k
        return self

    def __exit__(self, *exc):
        _tls.slot = self.prev


block_streams = {}   # (device, worker slot) -> the torch stream `functional.threaded_blocks`' worker of that slot always runs on
_reserving = threading.RLock()
DEFAULT_HW_QUEUES = 8


def configure_queues(max_hw_queues=DEFAULT_HW_QUEUES, reserve_worker_streams=4, device_index=None):
    """Explicit, logged set-up for SEVERAL BLOCKS IN FLIGHT on one GPU (`functional.threaded_blocks`, `concurrent_blocks`); nothing
    in the package calls it at import and a single-block user never needs it.  Call it once, before the process first touches the
    GPU (bench.py does, ahead of its sharded legs).

    1. Streams that share a hardware queue run their kernels in order.  ROCm maps all streams of a process onto GPU_MAX_HW_QUEUES
       queues, 4 by default; with a handle's own streams next to the workers' that is not enough -- gar8_hogp (four worker streams):
       1.75 / 1.77 s per step with 4 queues, 1.46 / 1.49 with 6, 1.47 / 1.51 with 8; 16 double cigar4.  Eight since the end of round 4:
       the main handle's third stream (step 2 below) needs a queue of its own as well, and with six the HOGP workers then share
       (1.72 s); 8 costs gar8 0.3-0.4 % (docs/concurrency.md).  The variable is process-wide (every HIP user of the process sees it) and only read
       when HIP initialises: an explicit setting in the environment wins, and a process whose HIP is already up gets a warning
       instead of a silent no-op.  `max_hw_queues=None` leaves the environment alone.
    2. `reserve_worker_streams` > 0 (and a GPU is present): the GPU's default handle is created and its side streams are used once
       (`ffgp_prepare_streams`), THEN `reserve_block_streams` -- the handle's look-ahead chain and the head of its triangular inverse,
       then the worker streams, claim their hardware queues before anything else of the process does.
    Returns the value of GPU_MAX_HW_QUEUES in effect for this process (None: ROCm's default)."""
    import torch

    if max_hw_queues is not None:
        cur = os.environ.get("GPU_MAX_HW_QUEUES")
        if cur is not None:
            log.info("GPU_MAX_HW_QUEUES=%s is set in the environment: kept (asked for %d)", cur, max_hw_queues)
        elif torch.cuda.is_initialized():
            log.warning("configure_queues(%d) called after HIP was initialised: GPU_MAX_HW_QUEUES cannot change any more; "
                        "blocks in flight will share ROCm's default 4 hardware queues", max_hw_queues)
        else:
            os.environ["GPU_MAX_HW_QUEUES"] = str(int(max_hw_queues))
            log.info("GPU_MAX_HW_QUEUES=%d set for this process (several blocks in flight per GPU)", max_hw_queues)
    if reserve_worker_streams and torch.cuda.is_available():
        # the GPU's default handle first: its side streams (look-ahead chain, head of the triangular inverse) bind their hardware
        # queues at creation, the workers take theirs after -- reserved the other way round, the handle's third stream shared the
        # caller's queue (a training step at N = 4096: 3.4-3.5 instead of 3.1 ms, tools/queue_probe.py)
        check(lib.ffgp_prepare_streams(handle(device_index, 0)), "ffgp_prepare_streams")
        reserve_block_streams(device_index, reserve_worker_streams)
    return os.environ.get("GPU_MAX_HW_QUEUES")


def block_stream(device_index, k):
    import torch

    with _reserving:
        st = block_streams.get((device_index, k))
        if st is None:
            st = block_streams[(device_index, k)] = torch.cuda.Stream(device_index)
        return st


def reserve_block_streams(device_index=None, n=4):
    """Create the worker streams of `functional.threaded_blocks` and put one tiny kernel on each, so that they claim their hardware
    queues before other multi-stream work of the process does.  ROCm binds a stream to one of GPU_MAX_HW_QUEUES hardware queues when
    the stream is first used, and streams on one queue run in order: worker streams that first appeared AFTER a `concurrent_blocks`
    workload (two more handles, each with its own streams) shared queues -- the same eight HOGP blocks then took 1.68 instead of
    1.45 s per step on one box, and the earlier workloads are not affected either way (docs/concurrency.md).  Called by
    `configure_queues` and by the first `threaded_blocks` of a GPU; idempotent; waits only for its own streams."""
    import torch

    if device_index is None:
        device_index = torch.cuda.current_device()
    with _reserving:
        fresh = [k for k in range(n) if (device_index, k) not in block_streams]
        if not fresh:
            return
        x = torch.zeros(64, device=torch.device("cuda", device_index))
        torch.cuda.current_stream(device_index).synchronize()
        for k in fresh:
            st = block_stream(device_index, k)
            with torch.cuda.stream(st):
                x.add_(1.0)
            st.synchronize()
        log.info("reserved %d worker streams on cuda:%d", len(fresh), device_index)


def current_slot():
    """the calling thread's handle slot: 0 outside a `thread_slot` block"""
    return getattr(_tls, "slot", 0)


def handle(device_index=None, slot=None):
    """ffgp handles of this process: one per (GPU, slot).  Slot 0 is the default (slot=None: the calling thread's current
    slot, 0 outside a `thread_slot` block); extra slots carry independent
    GP blocks that should overlap on the same GPU (each handle owns its workspace and side stream).

    Threading rule (include/ffgp.h): a handle carries mutable state (stream binding, workspaces, cached inverses), so one
    host thread at a time per handle.  The handles here are process-wide: a second Python thread that drives the GPU path
    concurrently must use its own slot(s) -- `functional.nlml(..., slot=k)` / `concurrent_blocks` -- not slot 0."""
    import torch

    if not torch.cuda.is_available():
        raise FFGPError("fidelityfusion_amd needs an MI355X (gfx950) GPU: torch.cuda.is_available() is False "
                        "and the package has no CPU path")
    if device_index is None:
        device_index = torch.cuda.current_device()
    if slot is None:
        slot = current_slot()
    with _lock:
        h = _handles.get((device_index, slot))
        if h is None:
            out = C.c_void_p()
            check(lib.ffgp_create(int(device_index), C.byref(out)), "ffgp_create")
            h = out
            _handles[(device_index, slot)] = h
    return h


def _device_handles(device_index):
    """every handle this process holds on that GPU (slot 0 is created on demand)"""
    h0 = handle(device_index, 0)
    if device_index is None:
        import torch
        device_index = torch.cuda.current_device()
    with _lock:
        return [h0] + [h for (dv, sl), h in sorted(_handles.items(), key=lambda kv: kv[0]) if dv == device_index and sl != 0]


def set_option(key, value, device_index=None, all_slots=False):
    """library option on the default handle of the GPU; all_slots: on every handle held there (concurrent blocks)"""
    for h in (_device_handles(device_index) if all_slots else [handle(device_index)]):
        check(lib.ffgp_set_option(h, key.encode(), float(value)), "ffgp_set_option(%s)" % key)


def set_option_handle(h, key, value):
    check(lib.ffgp_set_option(h, key.encode(), float(value)), "ffgp_set_option(%s)" % key)


def bind_stream(h, device_index):
    import torch

    s = torch.cuda.current_stream(device_index).cuda_stream
    check(lib.ffgp_set_stream(h, C.c_void_p(s)), "ffgp_set_stream")


def last_timings(device_index=None):
    h = handle(device_index)
    ms = (C.c_float * 16)()
    names = (C.c_char_p * 16)()
    n = C.c_int(0)
    check(lib.ffgp_last_timings(h, ms, names, 16, C.byref(n)), "ffgp_last_timings")
    return {names[i].decode(): float(ms[i]) for i in range(n.value)}


def syrk_stats(reset=False, device_index=None, all_slots=False):
    """launch statistics of the trailing-update SYRK (the roofline kernel); all_slots sums over every handle on the GPU"""
    tot = {"flops": 0.0, "ms": 0.0, "launches": 0}
    for h in (_device_handles(device_index) if all_slots else [handle(device_index)]):
        fl, ms, ln = C.c_double(0), C.c_double(0), C.c_long(0)
        check(lib.ffgp_syrk_stats(h, C.byref(fl), C.byref(ms), C.byref(ln), 1 if reset else 0), "ffgp_syrk_stats")
        tot["flops"] += fl.value
        tot["ms"] += ms.value
        tot["launches"] += ln.value
    return tot


def mfma_f64_peak(device_index=None):
    h = handle(device_index)
    out = C.c_double(0)
    check(lib.ffgp_mfma_f64_peak(h, C.byref(out)), "ffgp_mfma_f64_peak")
    return out.value
