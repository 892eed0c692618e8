import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the HIP library is built in-tree (git-ignored); build it once if a fresh checkout has not done so yet
    so = os.path.join(ROOT, "fidelityfusion_amd", "libffgp.so")
    if not os.path.exists(so):
        import __graft_entry__
        __graft_entry__.build()


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return dict(np.load(os.path.join(GOLDEN, name + ".npz")))

    return load


@pytest.fixture(scope="session", autouse=True)
def _gpu_noise():
    """FFGP_TEST_NOISE=1: the whole GPU suite runs while a background host thread keeps the chip busy with eigensolver and GEMM work on
    its own handle slot and stream.  Kernels whose workgroups are only correct when their waves run undisturbed (a missing barrier: two
    found so far, both invisible on an idle GPU) then fail the tests they already have.  Off by default: the suite's timing-sensitive
    comparisons and the round-end run stay as they are."""
    if not os.environ.get("FFGP_TEST_NOISE"):
        yield
        return
    import threading
    import time
    import torch
    if not torch.cuda.is_available():
        yield
        return
    from fidelityfusion_amd import _lib
    from fidelityfusion_amd import eigh as E
    from fidelityfusion_amd import functional as F
    stop = threading.Event()
    count = [0]
    # torch.linalg's vendor routines are the REFERENCES of many tests, and on this image they are not safe beside GPU work from
    # another host thread: torch.linalg.cholesky of a 1500 x 1500 SPD matrix returned wrong factors (up to 4e-2 relative) in 2 of 250
    # calls next to a plain torch.matmul loop on another stream, in 8-84 of 250 next to this library's eigh / nlml, while the
    # library's own results on the same inputs never moved (docs/experiments.md).  So the load pauses while a comparator runs: every
    # round of the load and every wrapped vendor call take the same lock.
    vendor = threading.Lock()
    import functools
    patched = []
    for name in ("cholesky", "cholesky_ex", "eigh", "eigvalsh", "solve_triangular", "inv", "solve", "slogdet", "det", "qr", "svd", "lstsq"):
        real = getattr(torch.linalg, name, None)
        if real is None:
            continue

        def quiet(*a, _real=real, **kw):
            with vendor:
                out = _real(*a, **kw)
                torch.cuda.synchronize()
                return out
        functools.update_wrapper(quiet, real)
        setattr(torch.linalg, name, quiet)
        patched.append((name, real))

    def noise():
        torch.cuda.set_device(0)
        dev = torch.device("cuda:0")
        st = torch.cuda.Stream(0)
        with torch.cuda.stream(st), _lib.thread_slot(7), torch.no_grad():
            g = torch.Generator(device=dev).manual_seed(99)
            X = torch.rand((2048, 6), generator=g, device=dev, dtype=torch.float64)
            d = torch.cdist(X, X)
            K = torch.exp(-0.5 * d * d)
            B = torch.randn((3072, 2048), generator=g, device=dev, dtype=torch.float64)
            Y = torch.randn((2048, 3), generator=g, device=dev, dtype=torch.float64)
            w = torch.ones(6, device=dev, dtype=torch.float64)
            amp = torch.ones(1, device=dev, dtype=torch.float64)
            dadd = torch.full((1,), 0.05, device=dev, dtype=torch.float64)
            while not stop.is_set():           # three kinds of neighbour in rotation: which kernels share a CU decides what gets disturbed
                kind = count[0] % 3
                with vendor:
                    if kind == 0:
                        E.eigh(K)                  # 1024-thread QR workgroups, the chase, divide & conquer, both back-transformations
                    elif kind == 1:
                        for _ in range(6):
                            F.matmul_nt(B, B)      # 128 x 128 GEMM tiles on every CU
                    else:
                        for _ in range(8):
                            F.nlml(X, Y, w, amp, diag_add=dadd, clamp=1e-30)   # assembly, the blocked factorisation's chain, reductions
                    st.synchronize()
                count[0] += 1
                time.sleep(0.002)              # (outside the lock: a comparator that is waiting gets its turn -- Python's locks are not fair)

    t = threading.Thread(target=noise, name="ffgp-test-noise", daemon=True)
    t.start()
    yield
    stop.set()
    t.join(timeout=60)
    for name, real in patched:
        setattr(torch.linalg, name, real)
    print("\n[ffgp] background noise thread ran %d rounds during the session" % count[0])
