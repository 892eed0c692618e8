"""Several independent GP blocks in flight on one GPU: `concurrent_blocks` (one host thread, one handle slot and stream per block,
status collected at the end), `threaded_blocks` (one host thread per slot: for blocks whose library calls wait on the host, the HOGP
blocks' eigensolver) and the stream / hardware-queue set-up they share.  Reference: the per-fidelity loops
FidelityFusion_Models/CIGAR.py:99-134, GAR.py:76-126.
"""
import threading

import torch

from . import _lib
from ._common import _raise_not_pd
from ._lib import check, lib


_pending = {}   # (device, slot) -> staging tensors of enqueued-but-not-waited calls (kept alive until wait)


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
