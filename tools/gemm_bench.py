#!/usr/bin/env python3
"""GPU micro-benchmarks of the fp64 MFMA GEMM / SYRK kernel and of the factorisation stages (interleaved rounds
in one process, min and median over rounds).  Usage on the GPU box:  python tools/gemm_bench.py [what ...]"""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _devlib  # noqa: F401,E402  (development build: these switches are not in the shipped library)

import numpy as np
import torch

from fidelityfusion_amd import _lib


def p(t):
    return C.c_void_p(t.data_ptr())


def timeit(fn, rounds=5):
    ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1))
    return min(ts), float(np.median(ts))


def main():
    h = _lib.handle(0)
    _lib.bind_stream(h, 0)
    what = sys.argv[1:] or ["peak", "gemm", "syrk", "potrf"]
    res = {}
    if "peak" in what:
        res["mfma_f64_stream_tflops"] = _lib.mfma_f64_peak(0)
    dev = "cuda:0"
    g = torch.Generator(device=dev).manual_seed(0)
    if "gemm" in what:
        for (m, n, k) in [(8192, 8192, 512), (8192, 8192, 2048), (8192, 8192, 8192), (16384, 128, 128), (16384, 384, 128)]:
            A = torch.rand((m, k), generator=g, device=dev, dtype=torch.float64) - 0.5
            B = torch.rand((n, k), generator=g, device=dev, dtype=torch.float64) - 0.5
            Cm = torch.zeros((m, n), device=dev, dtype=torch.float64)
            for beta in (0.0, 1.0):
                fn = lambda: _lib.lib.ffgp_gemm(h, 0, 0, 0, 0, p(A), k, p(B), k, p(Cm), n, m, n, k, -1.0, beta)
                fn()
                tmin, tmed = timeit(fn)
                res["gemm_nt_%dx%dx%d_beta%d" % (m, n, k, int(beta))] = {"ms_min": tmin, "ms_med": tmed,
                                                                        "tflops": 2.0 * m * n * k / tmin / 1e9}
    if "syrk" in what:
        for (m, k) in [(16384, 512), (16384, 1024), (8192, 512), (4096, 512), (2048, 512), (1024, 512)]:
            A = torch.rand((m, k), generator=g, device=dev, dtype=torch.float64) - 0.5
            Cm = torch.zeros((m, m), device=dev, dtype=torch.float64)
            fn = lambda: _lib.lib.ffgp_gemm(h, 0, 0, 1, 0, p(A), k, p(A), k, p(Cm), m, m, m, k, -1.0, 1.0)
            fn()
            tmin, tmed = timeit(fn)
            res["syrk_lower_%dx%d" % (m, k)] = {"ms_min": tmin, "ms_med": tmed, "tflops": 1.0 * m * (m + 1) * k / tmin / 1e9}
    if "potrf" in what:
        for n in (2048, 4096, 8192, 16384):
            X = torch.rand((n, 8), generator=g, device=dev, dtype=torch.float64)
            w = torch.ones(8, device=dev, dtype=torch.float64)
            amp = torch.ones(1, device=dev, dtype=torch.float64)
            dadd = torch.full((1,), 0.37, device=dev, dtype=torch.float64)
            W = torch.empty((n, n), device=dev, dtype=torch.float64)
            for nbo, la in ((512, 0), (256, 1), (512, 1), (1024, 1)):
                _lib.set_option("nb_outer", nbo, 0)
                _lib.set_option("lookahead", la, 0)

                def fn():
                    _lib.lib.ffgp_assemble(h, p(X), n, p(X), n, 8, p(w), p(amp), 1e-30, p(dadd), None, 0, None, 0, 0.0, 0.0,
                                           p(W), n, 1, 0, 1.0)
                    rc = _lib.lib.ffgp_potrf(h, p(W), n, n)
                    assert rc == 0, rc
                fn()
                tmin, tmed = timeit(fn, rounds=3)
                res["assemble+potrf_n%d_nbo%d_la%d" % (n, nbo, la)] = {"ms_min": tmin, "tflops": n ** 3 / 3.0 / tmin / 1e9}
            _lib.set_option("nb_outer", 512, 0)
            _lib.set_option("lookahead", 1, 0)
    for k_, v in res.items():
        print(k_, json.dumps(v))


if __name__ == "__main__":
    main()


def diag_ablation():
    """time potrf_diag128 alone (inverse-only and factor entry) with phases masked out"""
    h = _lib.handle(0)
    _lib.bind_stream(h, 0)
    n = 128 * 64
    dev = "cuda:0"
    g = torch.Generator(device=dev).manual_seed(0)
    X = torch.rand((n, 8), generator=g, device=dev, dtype=torch.float64)
    w = torch.ones(8, device=dev, dtype=torch.float64)
    amp = torch.ones(1, device=dev, dtype=torch.float64)
    dadd = torch.full((1,), 0.37, device=dev, dtype=torch.float64)
    W = torch.empty((n, n), device=dev, dtype=torch.float64)
    _lib.lib.ffgp_assemble(h, p(X), n, p(X), n, 8, p(w), p(amp), 1e-30, p(dadd), None, 0, None, 0, 0.0, 0.0, p(W), n, 1, 0, 1.0)
    assert _lib.lib.ffgp_potrf(h, p(W), n, n) == 0
    # 64 sequential launches of the inverse-only entry (phases 0,3,4,5)
    for mask in (0, 8, 16, 32, 128, 8 + 16 + 32 + 128):
        _lib.set_option("diag_dbg", mask, 0)
        fn = lambda: _lib.lib.ffgp_trtri_diag(h, p(W), n, n)
        fn()
        tmin, _ = timeit(fn)
        print("diag inverse-only entry, mask %3d: %.1f us per launch" % (mask, tmin * 1e3 / 64))
    _lib.set_option("diag_dbg", 0, 0)


if "diag" in sys.argv[1:]:
    diag_ablation()


def diag_factor_ablation():
    h = _lib.handle(0)
    _lib.bind_stream(h, 0)
    dev = "cuda:0"
    n = 128
    g = torch.Generator(device=dev).manual_seed(0)
    X = torch.rand((n, 8), generator=g, device=dev, dtype=torch.float64)
    w = torch.ones(8, device=dev, dtype=torch.float64)
    amp = torch.ones(1, device=dev, dtype=torch.float64)
    dadd = torch.full((1,), 0.37, device=dev, dtype=torch.float64)
    W0 = torch.empty((n, n), device=dev, dtype=torch.float64)
    _lib.lib.ffgp_assemble(h, p(X), n, p(X), n, 8, p(w), p(amp), 1e-30, p(dadd), None, 0, None, 0, 0.0, 0.0, p(W0), n, 1, 0, 1.0)
    Ws = [W0.clone() for _ in range(50)]
    _lib.set_option("lookahead", 0, 0)
    for mask in (0, 1, 2, 4, 64, 1 + 2 + 4 + 64, 255):
        _lib.set_option("diag_dbg", mask, 0)

        def fn():
            for Wt in Ws:
                Wt.copy_(W0)
            for Wt in Ws:
                _lib.lib.ffgp_potrf(h, p(Wt), n, n)
        fn()
        tmin, _ = timeit(fn)
        print("potrf(n=128) x50 incl. host sync, mask %3d: %.1f us per call" % (mask, tmin * 1e3 / 50))
    _lib.set_option("diag_dbg", 0, 0)
    _lib.set_option("lookahead", 1, 0)


if "diagf" in sys.argv[1:]:
    diag_factor_ablation()


def potrf_sweep():
    h = _lib.handle(0)
    _lib.bind_stream(h, 0)
    dev = "cuda:0"
    g = torch.Generator(device=dev).manual_seed(0)
    for n in (16384, 8192):
        X = torch.rand((n, 16), generator=g, device=dev, dtype=torch.float64)
        w = torch.ones(16, device=dev, dtype=torch.float64)
        amp = torch.ones(1, device=dev, dtype=torch.float64)
        dadd = torch.full((1,), 0.37, device=dev, dtype=torch.float64)
        W = torch.empty((n, n), device=dev, dtype=torch.float64)
        for nbo in (512, 768, 1024):
            for split in (0, 1):
                for thr in (256, 384, 640):
                    _lib.set_option("nb_outer", nbo, 0)
                    _lib.set_option("la_split", split, 0)
                    _lib.set_option("small_tile_threshold", thr, 0)

                    def fn():
                        _lib.lib.ffgp_assemble(h, p(X), n, p(X), n, 16, p(w), p(amp), 1e-30, p(dadd), None, 0, None, 0, 0.0,
                                               0.0, p(W), n, 1, 0, 1.0)
                        assert _lib.lib.ffgp_potrf(h, p(W), n, n) == 0
                    fn()
                    tmin, tmed = timeit(fn, rounds=3)
                    print("n=%d nb_outer=%d split=%d thr=%d: %.2f ms (%.1f TF/s)" % (n, nbo, split, thr, tmin, n ** 3 / 3.0 / tmin / 1e9))
    _lib.set_option("nb_outer", 512, 0)
    _lib.set_option("la_split", 0, 0)
    _lib.set_option("small_tile_threshold", 384, 0)


if "sweep" in sys.argv[1:]:
    potrf_sweep()


def layout_bench():
    h = _lib.handle(0)
    _lib.bind_stream(h, 0)
    dev = "cuda:0"
    g = torch.Generator(device=dev).manual_seed(0)
    m = n = k = 8192
    A = torch.rand((m, k), generator=g, device=dev, dtype=torch.float64) - 0.5
    B = torch.rand((n, k), generator=g, device=dev, dtype=torch.float64) - 0.5
    Cm = torch.zeros((m, n), device=dev, dtype=torch.float64)
    for opa in (0, 1):
        for opb in (0, 1):
            for lower, tri in ((0, 0), (1, 0)):
                if lower and opa != opb:
                    continue
                fn = lambda: _lib.lib.ffgp_gemm(h, opa, opb, lower, tri, p(A), k, p(B), k, p(Cm), n, m, n, k, -1.0, 0.0)
                assert fn() == 0
                tmin, tmed = timeit(fn, rounds=3)
                fl = 2.0 * m * n * k * (0.5 if lower else 1.0)
                print("gemm 8192^3 opa=%d opb=%d lower=%d: %.2f ms %.1f TF/s" % (opa, opb, lower, tmin, fl / tmin / 1e9))
    # LAUUM / TRTRI shaped
    X = torch.tril(A)
    fn = lambda: _lib.lib.ffgp_gemm(h, 1, 1, 1, 1, p(X), k, p(X), k, p(Cm), n, m, n, k, 1.0, 0.0)
    assert fn() == 0
    tmin, _ = timeit(fn, rounds=3)
    print("lauum-shaped 8192 (MN,MN,lower,lo_i): %.2f ms %.1f TF/s (n^3/3 flops)" % (tmin, m ** 3 / 3.0 / tmin / 1e9))


if "layout" in sys.argv[1:]:
    layout_bench()


def prio_bench():
    h = _lib.handle(0)
    _lib.bind_stream(h, 0)
    dev = "cuda:0"
    g = torch.Generator(device=dev).manual_seed(0)
    n = 16384
    X = torch.rand((n, 16), generator=g, device=dev, dtype=torch.float64)
    w = torch.ones(16, device=dev, dtype=torch.float64)
    amp = torch.ones(1, device=dev, dtype=torch.float64)
    dadd = torch.full((1,), 0.37, device=dev, dtype=torch.float64)
    W = torch.empty((n, n), device=dev, dtype=torch.float64)
    for pr in (1, 0, 1, 0):
        _lib.set_option("aux_prio", pr, 0)

        def fn():
            _lib.lib.ffgp_assemble(h, p(X), n, p(X), n, 16, p(w), p(amp), 1e-30, p(dadd), None, 0, None, 0, 0.0, 0.0, p(W), n, 1, 0, 1.0)
            assert _lib.lib.ffgp_potrf(h, p(W), n, n) == 0
        fn()
        tmin, tmed = timeit(fn, rounds=4)
        print("n=%d aux_prio=%d: min %.2f ms med %.2f ms" % (n, pr, tmin, tmed))
    _lib.set_option("aux_prio", 1, 0)


if "prio" in sys.argv[1:]:
    prio_bench()


def multiblock_bench():
    """aggregate throughput of F independent GP blocks on ONE GPU: one at a time vs overlapped on separate slots"""
    import time
    from fidelityfusion_amd import functional as F
    from oracle import gp_oracle as O
    dev = torch.device("cuda:0")
    for (n, D, d, nblk) in [(8192, 8, 1, 4), (8192, 8, 1024, 4), (16384, 16, 1, 2), (4096, 8, 1, 8)]:
        data = []
        for f in range(nblk):
            X, Y = O.synthetic_xy(n, D, d, seed=f)
            data.append((torch.tensor(X, device=dev), torch.tensor(Y, device=dev)))
        w = torch.ones(D, dtype=torch.float64, device=dev)
        amp = torch.ones(1, dtype=torch.float64, device=dev)
        dadd = torch.tensor([np.exp(-1.0) + 1e-6], dtype=torch.float64, device=dev)
        flops = nblk * (n ** 3 / 3.0 + float(n) * n * d + 2.0 * n * n * D)
        for nslots in (1, 2, 3, 4):
            def run():
                if nslots == 1:
                    return [F.nlml(x, y, w, amp, diag_add=dadd, clamp=1e-30) for x, y in data]
                outs = []
                with F.concurrent_blocks(nslots=nslots) as cb:
                    for f, (x, y) in enumerate(data):
                        with cb.slot(f):
                            outs.append(F.nlml(x, y, w, amp, diag_add=dadd, clamp=1e-30, **F._slot_args()))
                return outs
            run()
            torch.cuda.synchronize()
            best = 1e9
            for _ in range(3):
                t0 = time.perf_counter()
                outs = run()
                torch.cuda.synchronize()
                best = min(best, time.perf_counter() - t0)
            print("N=%d d=%d blocks=%d slots=%d: %.2f ms total, %.1f TF/s aggregate, nll0=%.6f" % (
                n, d, nblk, nslots, best * 1e3, flops / best / 1e12, float(outs[0])))


if "multiblock" in sys.argv[1:]:
    multiblock_bench()



def tile32_sweep():
    h = _lib.handle(0)
    _lib.bind_stream(h, 0)
    dev = "cuda:0"
    g = torch.Generator(device=dev).manual_seed(0)
    for n in (16384, 8192, 4096):
        X = torch.rand((n, 16), generator=g, device=dev, dtype=torch.float64)
        w = torch.ones(16, device=dev, dtype=torch.float64)
        amp = torch.ones(1, device=dev, dtype=torch.float64)
        dadd = torch.full((1,), 0.37, device=dev, dtype=torch.float64)
        W = torch.empty((n, n), device=dev, dtype=torch.float64)
        for thr in (0, 256, 512, 1024, 2048, 4096, 100000):
            _lib.set_option("tile32_threshold", thr, 0)

            def fn():
                _lib.lib.ffgp_assemble(h, p(X), n, p(X), n, 16, p(w), p(amp), 1e-30, p(dadd), None, 0, None, 0, 0.0, 0.0, p(W), n, 1,
                                       0, 1.0)
                assert _lib.lib.ffgp_potrf(h, p(W), n, n) == 0
            fn()
            tmin, tmed = timeit(fn, rounds=3)
            print("n=%d tile32_threshold=%d: %.2f ms (%.1f TF/s)" % (n, thr, tmin, n ** 3 / 3.0 / tmin / 1e9))
    _lib.set_option("tile32_threshold", 1024, 0)


if "tile32" in sys.argv[1:]:
    tile32_sweep()


def split_sweep():
    """split tail of the 128-tile launches: the trailing SYRK at every size the N = 16384 factorisation sees, and the
    whole factorisation, for split_rem_max = 0 (off) / 100 / 180 / 255"""
    h = _lib.handle(0)
    _lib.bind_stream(h, 0)
    dev = "cuda:0"
    g = torch.Generator(device=dev).manual_seed(0)
    k = 512
    A = torch.rand((16384, k), generator=g, device=dev, dtype=torch.float64) - 0.5
    Cm = torch.zeros((16384, 16384), device=dev, dtype=torch.float64)
    opts = (0, 100, 180, 255)
    tot = {o: 0.0 for o in opts}
    for m in range(15872, 2047, -512):
        T = m // 128
        tiles = T * (T + 1) // 2
        row = []
        for o in opts:
            _lib.set_option("split_rem_max", o, 0)
            fn = lambda: _lib.lib.ffgp_gemm(h, 0, 0, 1, 0, p(A), k, p(A), k, p(Cm), 16384, m, m, k, -1.0, 1.0)
            fn()
            tmin, _ = timeit(fn, rounds=7)
            tot[o] += tmin
            row.append("%.3f" % tmin)
        print("m=%5d tiles=%5d rem=%3d: ms %s" % (m, tiles, tiles % 256, " / ".join(row)))
    print("sum over sizes: " + " / ".join("%d: %.2f ms" % (o, tot[o]) for o in opts))
    n = 16384
    X = torch.rand((n, 16), generator=g, device=dev, dtype=torch.float64)
    w = torch.ones(16, device=dev, dtype=torch.float64)
    amp = torch.ones(1, device=dev, dtype=torch.float64)
    dadd = torch.full((1,), 0.37, device=dev, dtype=torch.float64)
    for o in opts + opts:
        _lib.set_option("split_rem_max", o, 0)

        def fn():
            _lib.lib.ffgp_assemble(h, p(X), n, p(X), n, 16, p(w), p(amp), 1e-30, p(dadd), None, 0, None, 0, 0.0, 0.0, p(Cm), n, 1, 0,
                                   1.0)
            assert _lib.lib.ffgp_potrf(h, p(Cm), n, n) == 0
        fn()
        tmin, tmed = timeit(fn, rounds=5)
        print("assemble+potrf n=%d split_rem_max=%d: %.2f ms min, %.2f med" % (n, o, tmin, tmed))
    _lib.set_option("split_rem_max", 180, 0)


if "split" in sys.argv[1:]:
    split_sweep()
