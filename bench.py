#!/usr/bin/env python3
"""Headline benchmark: GP NLML + Cholesky throughput (fp64) on MI355X -- BASELINE.json's metric.

A step = one pass of the hot path over one batch of GP blocks: covariance assembly -> blocked Cholesky (Y^T riding as
passenger rows, so Gamma = L^-1 Y comes out of the factorisation's own GEMMs) -> log-det / ||Gamma||^2 reductions ->
NLML scalar, with X, Y and the hyper-parameters already resident in HBM.

Workloads (--workload):
  headline (default)  BASELINE configs[2]: single-fidelity cigp, ARD kernel, N = 16384, D = 16, d = 1 -- one such block
                      per rank (weak scaling; fidelity f = rank), ONE 8*F-byte all-reduce of the per-block values per step.
  c2                  BASELINE configs[1]: N = 4096, D = 8, d = 1 (same sharding as headline).
  cigar4              BASELINE configs[3]: F = 4 fixed blocks of N = 8192, D = 8, d = 1024 (strong scaling).
  gar8                BASELINE configs[4]: F = 8 fixed blocks of N = 8192, D = 8, d = 4096 (strong scaling).
  gar8_hogp           the same configuration with the blocks as the reference writes them (FidelityFusion_Models/GAR.py:76-126):
                      HOGP_simple.log_likelihood on N = 8192, d = 64 x 64 -- eigendecomposition of the N x N input kernel by the
                      library's own two-stage solver (ffgp_syevd) + the mode products on the fp64 GEMM ("HOGP variant
                      reported separately").
For the fixed-F workloads the blocks are dealt to the ranks by longest-processing-time-first
(fidelityfusion_amd.sharding.partition_lpt -- the reference's per-fidelity loop, FidelityFusion_Models/CIGAR.py:99-134,
GAR.py:76-126; the sum MFGP_ver2023May/ResGP.py:232-246); a rank that owns several blocks overlaps them on its GPU
(functional.concurrent_blocks from one host thread; the HOGP blocks, whose eigensolver waits on the host after its bulge chasing,
from `--hogp-slots` host threads with a handle and a stream each: functional.threaded_blocks); one all-reduce(SUM) of the F-vector
per step.  The default run also times both fixed-F
workloads for a few steps after the headline and reports them under "sharded", so that one `--gpus N` sweep carries the
GAR-8 / CIGAR-4 curve next to the headline.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload W]

With --gpus N > 1 and no WORLD_SIZE in the environment the script launches its own N ranks (child processes, one per GPU,
rendezvous on 127.0.0.1) BEFORE importing torch or touching the GPU; under `python -m torch.distributed.run
--nproc-per-node N ... bench.py --gpus N` it is one of the ranks.  Rank 0 prints ONE JSON line.

--backend gloo --dry runs the launch / partition / all-reduce / JSON plumbing on CPU with a stand-in block value (no GPU,
no likelihood arithmetic): the CPU test of the multi-rank path (tests/test_bench_launch.py).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FP64_MFMA_PEAK_TFLOPS = 78.6   # MI355X fp64 matrix peak (vendor spec; 256 CU x 4 SIMD x 32 FLOP/clk x 2.4 GHz)
ROOFLINE_KERNEL = "ffgp_gemm_f64<0, 0, 1, 1, 128, 128>"   # trailing SYRK update of the blocked Cholesky

WORKLOADS = {   # name: (F or None = one block per rank, N, D, d, BASELINE config index)
    "headline": (None, 16384, 16, 1, 2),
    "c2": (None, 4096, 8, 1, 1),
    "cigar4": (4, 8192, 8, 1024, 3),
    "gar8": (8, 8192, 8, 4096, 4),
    "gar8_hogp": (8, 8192, 8, 4096, 4),
}
HOGP_MODES = (64, 64)   # output shape of a gar8_hogp block (prod = d)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="headline")
    ap.add_argument("--n", type=int, default=None, help="override the workload's N (development)")
    ap.add_argument("--D", type=int, default=None)
    ap.add_argument("--d", type=int, default=None)
    ap.add_argument("--blocks", type=int, default=None, help="override F of a fixed-F workload (development / --dry)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sharded", action="store_true", help="skip the fixed-F legs after the headline")
    ap.add_argument("--cpu-budget-s", type=float, default=40.0, help="wall-clock bound of the CPU baseline leg")
    ap.add_argument("--with-grad", action="store_true", help="time forward + closed-form gradients instead")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl")
    ap.add_argument("--dry", action="store_true", help="CPU plumbing run (needs --backend gloo): no GPU work")
    ap.add_argument("--slots", type=int, default=2, help="blocks of one rank that overlap on its GPU (fixed-F workloads)")
    ap.add_argument("--slot-lookahead", action="store_true", help="keep every overlapped block's own look-ahead side stream")
    ap.add_argument("--no-chain-batch", action="store_true", help="fixed-F workloads: overlap a rank's blocks through streams "
                    "(functional.concurrent_blocks) instead of sharing ONE factorisation chain (functional.nlml_many)")
    ap.add_argument("--hogp-slots", type=int, default=4, help="host threads that drive the HOGP blocks of one rank (gar8_hogp): "
                    "1 = one block after another (2.46 s/step), 2: 1.87, 3: 1.66, 4: 1.54, 8: 1.58 s/step on one MI355X (sweep taken with 8 hardware queues; the default of 6 measures the same)")
    ap.add_argument("--opt", action="append", default=[], help="library option key=value (development A/B runs)")
    return ap.parse_args(argv)


# ---------------------------------------------------------------------------------------------------------------
# self-launch: the parent never imports torch, never touches the GPU, never exec's
# ---------------------------------------------------------------------------------------------------------------
def spawn_ranks(args):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL needs it on this driver
        env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 1) // args.gpus)))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    try:
        pending = list(procs)
        while pending:
            for p in list(pending):
                r = p.poll()
                if r is None:
                    continue
                pending.remove(p)
                if r != 0:                  # one rank died: the others would wait in a collective for ever
                    rc = rc or r
                    for q in pending:
                        q.terminate()
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc


def gpu_numa_cpus(local_rank, sysfs=None):
    """CPUs of the NUMA node the local_rank-th AMD GPU hangs off, read from sysfs without touching HIP (cards in PCI-address
    order, which is the order ROCm enumerates them in on one node); None when sysfs does not say.  `sysfs` (or FFGP_BENCH_SYSFS)
    points the lookup at another tree: the CPU pre-flight test of an 8-GPU, 2-socket node (tests/test_bench_launch.py)."""
    import glob
    sysfs = sysfs or os.environ.get("FFGP_BENCH_SYSFS", "/sys")
    cards = []
    for dev in glob.glob(os.path.join(sysfs, "class/drm/card[0-9]*/device")):
        try:
            if open(os.path.join(dev, "vendor")).read().strip() != "0x1002":
                continue
            cards.append((os.path.basename(os.path.realpath(dev)), dev))
        except OSError:
            continue
    cards.sort()
    if local_rank >= len(cards):
        return None
    try:
        node = int(open(os.path.join(cards[local_rank][1], "numa_node")).read().strip())
        if node < 0:
            return None
        cpus = set()
        for part in open(os.path.join(sysfs, "devices/system/node/node%d/cpulist" % node)).read().strip().split(","):
            lo, _, hi = part.partition("-")
            cpus.update(range(int(lo), int(hi or lo) + 1))
        return sorted(cpus)
    except (OSError, ValueError):
        return None


def rank_cpu_slice(local_rank, local_world, allowed, sysfs=None):
    """The CPUs rank `local_rank` of `local_world` pins itself to (pure: nothing is applied): its own slice of the CPUs of its GPU's
    NUMA node, shared evenly with the other ranks whose GPUs hang off the same node; an even slice of `allowed` when sysfs is silent."""
    node = gpu_numa_cpus(local_rank, sysfs)
    pool = [c for c in (node or allowed) if c in set(allowed)] or allowed
    # the ranks whose GPUs share this pool split it evenly (with one NUMA node per GPU pair, two ranks share a pool)
    sharers = [r for r in range(local_world) if (gpu_numa_cpus(r, sysfs) or allowed) == (node or allowed)] or [local_rank]
    k = sharers.index(local_rank) if local_rank in sharers else 0
    per = max(1, len(pool) // len(sharers))
    return pool[k * per:(k + 1) * per] or pool


def block_path(n_mine, no_chain_batch=False):
    """Which evaluation a rank's step takes for its `n_mine` blocks of one fixed-F workload: "single" = the one-block fused call
    (F.nlml -- what every GPU runs when the blocks are dealt one per GPU), "chain" = ONE shared factorisation chain for the rank's
    blocks (F.nlml_many), "streams" = the blocks overlapped through streams, "idle" = the rank owns no block."""
    if n_mine <= 0:
        return "idle"
    if n_mine == 1:
        return "single"
    return "streams" if no_chain_batch else "chain"


def pin_rank(local_rank, local_world):
    """Per-rank CPU affinity, set BEFORE the process touches the GPU (the factorisation is a host-driven chain of ~400 launches per
    29 ms step: a launch thread that migrates, or shares its core with another rank's, arrives late and the GPU idles -- 29 -> 35 ms
    on a busy host).  The rank gets its own slice of the CPUs of its GPU's NUMA node (an even slice of the allowed
    CPUs when sysfs does not name the node); FFGP_BENCH_AFFINITY=0 leaves the affinity alone.  Returns the CPUs pinned to, or None."""
    if os.environ.get("FFGP_BENCH_AFFINITY", "1") == "0" or not hasattr(os, "sched_setaffinity"):
        return None
    mine = rank_cpu_slice(local_rank, local_world, sorted(os.sched_getaffinity(0)))
    try:
        os.sched_setaffinity(0, mine)
    except OSError:
        return None
    return mine


# ---------------------------------------------------------------------------------------------------------------
# workload pieces
# ---------------------------------------------------------------------------------------------------------------
def synthetic_xy(n, D, d, seed=0):
    """The benchmark's deterministic workload: X ~ U[0,1)^D, Y = sin(2 pi X w) + 0.1 randn, normalised
    the way FidelityFusion_Models/MF_data.py:26-28 normalises y (mean / unbiased std over all entries)."""
    import numpy as np
    rng = np.random.default_rng(seed)
    X = rng.random((n, D))
    W = rng.random((D, d))
    Y = np.sin(2.0 * np.pi * (X @ W)) + 0.1 * rng.standard_normal((n, d))
    Y = (Y - Y.mean()) / (Y.std(ddof=1) + 1e-10)
    return X, Y


def synthetic_xy_device(n, D, d, seed, dev):
    """Same recipe generated on the device (the d = 4096 blocks of the fixed-F workloads: 268 MB of targets each)."""
    import math
    import torch
    g = torch.Generator(device=dev).manual_seed(1000 + seed)
    X = torch.rand((n, D), generator=g, device=dev, dtype=torch.float64)
    W = torch.rand((D, d), generator=g, device=dev, dtype=torch.float64)
    Y = torch.sin(2.0 * math.pi * (X @ W)) + 0.1 * torch.randn((n, d), generator=g, device=dev, dtype=torch.float64)
    Y = (Y - Y.mean()) / (Y.std() + 1e-10)
    return X, Y


def nlml_flops(n, D, d):
    """N^3/3 (Cholesky) + N^2 d (Gamma = L^-1 Y) + 2 N^2 D (distance contractions)."""
    return n ** 3 / 3.0 + float(n) * n * d + 2.0 * n * n * D


def hogp_flops_canonical(n, modes):
    """The same block priced with a canonical LAPACK-style syevd count -- tridiagonalisation 4/3 N^3, divide & conquer ~4/3 N^3 as
    GEMMs, ONE back-transformation 2 N^3 (= 4 2/3 N^3) -- plus the mode products: comparable with a rocSOLVER / LAPACK figure,
    unlike `hogp_flops` (what this build executes: two-stage overhead and the second back-transformation included)."""
    pd = 1.0
    for m in modes:
        pd *= m
    return (4.0 / 3.0 + 4.0 / 3.0 + 2.0) * float(n) ** 3 + 2.0 * 2.0 * float(n) * n * pd + 2.0 * 2.0 * float(n) * pd * sum(modes)


def hogp_flops(n, modes):
    """One HOGP_simple.log_likelihood forward as this build executes it: the two-stage eigensolver of the N x N input kernel --
    band reduction 2 N^3 (A V products 2/3, rank-64 updates 4/3), divide & conquer merges as dense GEMMs 8/3 N^3, the two
    back-transformations 2 N^3 each (useful flops; the staircase blocks of the second execute twice that) -- and the mode
    products: two mode-0 products 2 N^2 prod(d) each (T_1 = Y x_0 U^T, g = W x_0 U) plus the small per-mode ones."""
    pd = 1.0
    for m in modes:
        pd *= m
    eig = (2.0 + 8.0 / 3.0 + 2.0 + 2.0) * float(n) ** 3
    return eig + 2.0 * 2.0 * float(n) * n * pd + 2.0 * 2.0 * float(n) * pd * sum(modes)


def recorded_traffic(n):
    """HBM bytes per launch of the roofline kernel, REPLAYED from the committed rocprofv3 PMC passes of this same
    command (profiles/*_pmc_summary.json; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for 16-B/lane streams on
    gfx950) -- bench.py cannot collect PMC counters itself.  None when no recorded pass covers this workload; raises
    if a recorded pass exists but no longer contains the roofline kernel (a renamed kernel must not go unnoticed)."""
    import glob
    if n != 16384:
        return None, None
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_summary.json")))
    if not files:
        return None, None
    c = json.load(open(files[-1]))["counters"]
    keys = [k for k in c["FETCH_SIZE"] if k.startswith("void " + ROOFLINE_KERNEL)]
    if not keys:
        raise RuntimeError("%s holds no counters for the roofline kernel %s -- re-record the PMC pass"
                           % (files[-1], ROOFLINE_KERNEL))
    byt = (2.0 * c["FETCH_SIZE"][keys[0]]["per_launch"] + c["WRITE_SIZE"][keys[0]]["per_launch"]) * 1024.0
    return byt, os.path.relpath(files[-1], ROOT)


PARITY_POINTS = {"c2": 256, "headline": 64}    # posterior query points of the in-run parity columns


def parity_points(D, nt):
    """deterministic query points of the parity columns (same numbers on the GPU and in the CPU leg)"""
    import numpy as np
    return np.random.default_rng(4242).random((nt, D))


def rel_err(a, b):
    """max |a - b| / max |b| (the tests' measure)"""
    import numpy as np
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(float(np.abs(b).max()), 1e-300))


def gpu_parity_values(dev, names):
    """OUTSIDE the timed region: the drop-in `cigp` module (reference: GaussianProcess/cigp_v10.py:24-69) at the reference's initial
    parameters on the C2 / C3 inputs -- +LL, the gradients `loss.backward()` leaves on the raw parameters and on Y
    (FidelityFusion_Models/ResGP.py:84-88), and the posterior mean / full covariance at PARITY_POINTS query points.  The CPU leg
    computes the same quantities with the reference's torch-CPU operator sequence and reports the relative errors."""
    import torch
    from fidelityfusion_amd import kernel as K_
    from fidelityfusion_amd.cigp_v10 import cigp
    out = {}
    for name in names:
        _, n, D, d, _ = WORKLOADS[name]
        X, Y = synthetic_xy(n, D, d, seed=0)
        Xt = torch.tensor(X, dtype=torch.float64, device=dev)
        Yt = torch.tensor(Y, dtype=torch.float64, device=dev, requires_grad=True)
        Xs = torch.tensor(parity_points(D, PARITY_POINTS[name]), dtype=torch.float64, device=dev)
        m = cigp(K_.ARDKernel(D), 1.0).double().to(dev)     # length_scales = 1, signal_variance = 1, log_beta = 1
        ll = m.negative_log_likelihood(Xt, Yt)
        ll.backward()
        with torch.no_grad():
            mean, var = m(Xt, Yt.detach(), Xs)
        out[name] = {"ll": float(ll.detach()),
                     "grads": {"length_scales": m.kernel.length_scales.grad.cpu().numpy(),
                               "signal_variance": m.kernel.signal_variance.grad.cpu().numpy(),
                               "log_beta": m.log_beta.grad.cpu().numpy(), "Y": Yt.grad.cpu().numpy()},
                     "mean": mean.cpu().numpy(), "var": var.cpu().numpy()}
        del m, Xt, Yt, mean, var
        torch.cuda.empty_cache()
    return out


def vendor_potrf_ms(dev, sizes=(4096, 16384), reps=3):
    """The reference's own GPU path, stated beside the headline and never on the product path: the 2023 API moves its tensors with
    `.cuda()` (MFGP_ver2023May/mfgp_demo.py:88-94), so `torch.linalg.cholesky` there is the vendor solver on this very GPU.  Single
    host thread, idle GPU, the library's own factorisation (ffgp_potrf via functional.cholesky) on the same matrix next to it;
    min of `reps` after a warm-up, whole call including the vendor path's host synchronisation (tools/potrf_vs_vendor.py sweeps more sizes)."""
    import torch
    from fidelityfusion_amd import functional as F
    out = {}
    for n in sizes:
        X = torch.tensor(synthetic_xy(n, 8, 1, seed=1)[0], dtype=torch.float64, device=dev)
        S = torch.exp(-0.5 * torch.cdist(X, X) ** 2)
        S.diagonal().add_(0.368)

        def t(fn):
            fn()
            torch.cuda.synchronize()
            best = 1e9
            for _ in range(reps):
                t0 = time.perf_counter()
                fn()
                torch.cuda.synchronize()
                best = min(best, time.perf_counter() - t0)
            return round(best * 1e3, 3)

        out[str(n)] = {"vendor_potrf_ms": t(lambda: torch.linalg.cholesky(S)), "ffgp_potrf_ms": t(lambda: F.cholesky(S))}
        del X, S
        torch.cuda.empty_cache()
    return out


def cpu_baseline(budget_s, gpu_ref):
    """The reference's torch-CPU operator sequence (oracle/torch_cpu_ref.py: cdist -> exp -> eye adds -> linalg.cholesky
    -> solve_triangular -> V1 formula; autograd for the backward) timed on this box's host cores at C2 and C3 (SURVEY
    8d), 1 warm-up + min of 3 where the budget allows, next to the GPU's value on the same inputs (in-run parity).
    Thread count: SURVEY asks for os.cpu_count(); on a 2 x 64-core SMT host that oversubscribes MKL badly (measured:
    C2 forward 3.1 s on 256 threads), so a short sweep at N = 2048 picks the fastest of {all, physical, 64, 32, 16}
    logical CPUs this process may use and the protocol runs with that -- the sweep is reported."""
    import torch
    from oracle import torch_cpu_ref as T
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    one = lambda k: torch.ones(k, dtype=torch.float64)
    t_leg = time.perf_counter()
    Xs, Ys = (torch.tensor(a) for a in synthetic_xy(2048, 8, 1, seed=0))
    sweep = {}
    for t in sorted({avail, max(1, avail // 2), min(avail, 64), min(avail, 32), min(avail, 16)}, reverse=True):
        torch.set_num_threads(t)
        r = T.time_cigp(Xs, Ys, one(8), one(1), one(1), repeats=2, with_backward=False, budget_s=2.0)
        sweep[t] = round(r["fwd_s"] * 1e3, 2)
    threads = min(sweep, key=sweep.get)
    # the big configurations are Cholesky-bound, and the thread count that wins the whole forward at N = 2048 need not win
    # dpotrf at N = 16384: a second sweep on the factorisation alone at N = 4096 over {16, 32, 64, 128} (and the N = 2048 winner)
    g4 = torch.Generator().manual_seed(4)
    B4 = torch.randn((4096, 4096), generator=g4, dtype=torch.float64)
    S4 = B4 @ B4.T / 4096.0 + torch.eye(4096, dtype=torch.float64)
    del B4
    sweep4 = {}
    for t in sorted({threads} | {c for c in (16, 32, 64, 128) if c <= avail}):
        torch.set_num_threads(t)
        torch.linalg.cholesky(S4)
        best4 = 1e9
        for _ in range(2):
            t0 = time.perf_counter()
            torch.linalg.cholesky(S4)
            best4 = min(best4, time.perf_counter() - t0)
        sweep4[t] = round(best4 * 1e3, 2)
    del S4
    threads_big = min(sweep4, key=sweep4.get)
    torch.set_num_threads(threads)
    host = T.host_description()
    host["usable_cpus"] = avail
    out = {"unit": "GF/s", "cores": int(threads_big), "kind": "port", "host": host,
           "thread_sweep_ms_at_N2048": {str(k): v for k, v in sweep.items()},
           "potrf_thread_sweep_ms_at_N4096": {str(k): v for k, v in sweep4.items()}, "configs": {}}
    pred = None                                       # predicted C3 forward seconds, from C2's stages
    for name, share in (("c2", 0.25), ("headline", 1.0)):
        _, n, D, d, _ = WORKLOADS[name]
        left = budget_s - (time.perf_counter() - t_leg)
        if left < 1.0:
            break
        if name == "headline" and pred is not None:
            while n > 4096 and 1.3 * pred * (n / 16384.0) ** 3 > left:      # the warm-up alone would not fit: bounded sample
                n //= 2
            if n <= 4096:
                break
            name = "headline" if n == 16384 else "headline_sample_N%d" % n
        X, Y = synthetic_xy(n, D, d, seed=0)
        Xt, Yt = torch.tensor(X), torch.tensor(Y)
        use = threads if name == "c2" else threads_big     # C2 keeps the whole-forward winner, the Cholesky-bound sizes the potrf winner
        torch.set_num_threads(use)
        kept = {}
        r = T.time_cigp(Xt, Yt, one(D), one(1), one(1), repeats=3, with_backward=True, budget_s=left * share, keep=kept)
        fl = nlml_flops(n, D, d)
        c = {"N": n, "D": D, "d": d, "fwd_ms": round(r["fwd_s"] * 1e3, 2), "fwd_gflops": round(fl / r["fwd_s"] / 1e9, 1),
             "fwd_bwd_ms": None if r["fwd_bwd_s"] is None else round(r["fwd_bwd_s"] * 1e3, 2),
             "fwd_bwd_gflops": None if r["fwd_bwd_s"] is None else round(3.0 * fl / r["fwd_bwd_s"] / 1e9, 1),
             "stage_ms": None if r["stages_s"] is None else {k: round(v * 1e3, 2) for k, v in r["stages_s"].items()},
             "cpu_ll": r["ll"], "threads": use,
             "threads_from": "whole forward at N=2048 (thread_sweep_ms_at_N2048)" if name == "c2"
                             else "dpotrf at N=4096 (potrf_thread_sweep_ms_at_N4096)"}
        if name != "c2" and use != 16 and 16 <= avail and r["stages_s"]:
            # the factorisation stage once more at 16 threads (what earlier rounds' sweep picked), when the budget still allows
            if budget_s - (time.perf_counter() - t_leg) > 3.0 * r["stages_s"]["potrf"]:
                torch.set_num_threads(16)
                st16 = {}
                with torch.no_grad():
                    T.cigp_ll(Xt, Yt, one(D), one(1), one(1), stages=st16)
                c["potrf_ms_by_threads"] = {str(use): round(r["stages_s"]["potrf"] * 1e3, 2), "16": round(st16["potrf"] * 1e3, 2)}
                torch.set_num_threads(use)
        if name == "c2" and r["stages_s"]:
            pred = 64.0 * r["stages_s"]["potrf"] + 32.0 * r["stages_s"]["assemble"]
        g = gpu_ref.get(name)
        if g is not None:    # same inputs, same parameters: the in-run parity columns (north_star's gate is 1e-4 relative)
            c["gpu_ll"] = g["ll"]
            c["rel_err"] = abs(c["gpu_ll"] - c["cpu_ll"]) / abs(c["cpu_ll"])
            if "grads" in kept and "grads" in g:     # the autograd backward the reference really runs (ResGP.py:84-88)
                c["rel_err_grad"] = {k: rel_err(g["grads"][k], v.numpy()) for k, v in kept["grads"].items()}
            if "mean" in g and "L" in kept:          # cigp.forward (cigp_v10.py:24-48) on the CPU leg's own factor
                with torch.no_grad():
                    Xq = torch.tensor(parity_points(D, PARITY_POINTS[name]))
                    mean_c, var_c = T.cigp_forward(Xt, Yt, Xq, one(D), one(1), one(1), L=kept["L"])
                c["posterior_points"] = int(Xq.shape[0])
                c["rel_err_mean"] = rel_err(g["mean"], mean_c.numpy())
                c["rel_err_var"] = rel_err(g["var"], var_c.numpy())
        del kept
        out["configs"][name] = c
    best = ([v for k, v in out["configs"].items() if k.startswith("headline")] or [out["configs"].get("c2")])[0]
    out["value"] = best["fwd_gflops"] if best else None
    out["cores"] = int(best["threads"]) if best else out["cores"]
    out["sample"] = ("torch-CPU port of cigp.negative_log_likelihood (oracle/torch_cpu_ref.py), fp64, %d threads (fastest dpotrf of the "
                     "N=4096 sweep over {16, 32, 64, 128}) on %s, %s: forward at N=%d (min of up to 3 after a warm-up, bounded by "
                     "--cpu-budget-s); per-config fwd / fwd+bwd / stages / threads under configs"
                     % (out["cores"], host["cpu_model"], host["blas"], best["N"] if best else 0))
    return out


# ---------------------------------------------------------------------------------------------------------------
# one rank
# ---------------------------------------------------------------------------------------------------------------
def run_rank(args):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    pinned = None
    if world > 1:      # before torch is imported (its worker threads inherit the mask) and long before the first GPU call;
        pinned = pin_rank(local_rank, int(os.environ.get("LOCAL_WORLD_SIZE", world)))   # a lone rank keeps the whole host

    import numpy as np
    import torch
    import torch.distributed as dist

    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch N ranks, or drop WORLD_SIZE to let bench.py "
                         "launch them itself)" % (args.gpus, world))
    if args.dry:
        if args.backend != "gloo":
            raise SystemExit("bench.py: --dry is the CPU plumbing run; use it with --backend gloo")
        dev = torch.device("cpu")
    else:
        from fidelityfusion_amd import _lib
        _lib.configure_queues(reserve_worker_streams=0)      # GPU_MAX_HW_QUEUES: before this process first touches the GPU
        ndev = torch.cuda.device_count()                      # (counting devices does not initialise HIP on this image)
        if os.environ.get("FFGP_BENCH_ONE_GPU") == "1" and args.backend == "gloo" and ndev >= 1:
            # rehearsal of the multi-rank step on a one-GPU box: every rank drives cuda:0, the F-vector is reduced by gloo on the host
            # (tests/test_bench_launch.py); never a measurement
            local_rank = 0
        elif ndev < max(args.gpus, local_rank + 1):
            raise SystemExit("bench.py: --gpus %d (LOCAL_RANK %d) but only %d GPU%s visible to this process -- check "
                             "HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES or lower --gpus" % (args.gpus, local_rank, ndev, "" if ndev == 1 else "s"))
        assert torch.cuda.is_available(), "bench.py needs an MI355X (there is no CPU path; --dry only checks plumbing)"
        torch.cuda.set_device(local_rank)
        _lib.configure_queues(max_hw_queues=None, device_index=local_rank)   # the worker streams claim their hardware queues first
        dev = torch.device("cuda", local_rank)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")
    coll = {"backend": (dist.get_backend() if world > 1 else None), "ranks": (dist.get_world_size() if world > 1 else 1)}

    if not args.dry:
        from fidelityfusion_amd import _lib
        from fidelityfusion_amd import functional as F
        for kv in args.opt:   # development switches (tools/): --opt la_split=1 --opt tile32_threshold=0 ...
            key, val = kv.split("=")
            _lib.set_option(key, float(val), local_rank)
    from fidelityfusion_amd.sharding import block_cost, partition_lpt

    red_dev = dev if (args.backend == "nccl" and not args.dry) else torch.device("cpu")

    def fence():
        if world > 1:
            dist.barrier()
        if not args.dry:
            torch.cuda.synchronize()

    def params(D, with_grad):
        # reference initial hyper-parameters: length_scales = 1 (kernel.py:84), signal_variance = 1, log_beta = 1 (ResGP.py:27)
        w = torch.ones(D, dtype=torch.float64, device=dev, requires_grad=with_grad)
        amp = torch.ones(1, dtype=torch.float64, device=dev, requires_grad=with_grad)
        dadd = torch.tensor([np.exp(-1.0) + 1e-6], dtype=torch.float64, device=dev, requires_grad=with_grad)
        return w, amp, dadd

    deal = {}     # workload name -> how its blocks were dealt (owner of every block, the evaluation path of every rank)

    def make_workload(name, n=None, D=None, d=None, blocks=None, with_grad=None):
        """-> (step(), F_total, n, D, d, scaling): step() evaluates this rank's blocks and all-reduces the F-vector.
        with_grad: the step also produces every gradient `loss.backward()` leaves in the reference's training iteration
        (FidelityFusion_Models/ResGP.py:84-88) -- Y, length scales, amplitude, noise -- by the closed forms of the same fused call."""
        with_grad = args.with_grad if with_grad is None else with_grad
        F_fixed, n0, D0, d0, _ = WORKLOADS[name]
        n, D, d = n or n0, D or D0, d or d0
        if F_fixed is None:                                   # one block per rank: fidelity f = rank
            F_total, mine, scaling = world, [rank], "weak"
            owner = list(range(world))
        else:
            F_total = blocks or F_fixed
            owner = partition_lpt([block_cost(n, d)] * F_total, world)
            mine, scaling = [f for f in range(F_total) if owner[f] == rank], "strong"
        deal[name] = {"owner": [int(o) for o in owner],
                      "rank_paths": [block_path(sum(1 for o in owner if o == r), args.no_chain_batch) for r in range(world)]}
        joint = torch.zeros(F_total, dtype=torch.float64, device=red_dev)
        if args.dry:
            data = {f: synthetic_xy(n, D, d, seed=f) for f in mine}

            def step():                                       # stand-in block value: plumbing only
                joint.zero_()
                for f in mine:
                    joint[f] = float(np.square(data[f][1]).sum()) + f
                if world > 1:
                    dist.all_reduce(joint)
                return joint
            return step, F_total, n, D, d, scaling
        if name == "gar8_hogp":   # the blocks as HOGP_simple (GAR's per-fidelity model): own eigensolver + mode products
            from fidelityfusion_amd import kernel as K_
            from fidelityfusion_amd.hogp_simple import HOGP_simple
            modes = HOGP_MODES if d == HOGP_MODES[0] * HOGP_MODES[1] else (d, 1)
            hdata, models = {}, {}
            for f in mine:
                X, Y = synthetic_xy_device(n, D, d, f, dev)
                hdata[f] = (X, Y.reshape(n, *modes))
                models[f] = HOGP_simple(K_.ARDKernel(D), 1.0, list(modes)).double().to(dev)

            hslots = max(1, min(args.hogp_slots, len(mine)))

            def step():
                joint.zero_()
                with torch.no_grad():   # blocks overlap from `--hogp-slots` host threads (ffgp_syevd waits for its chase on the host)
                    vals = F.threaded_blocks([(lambda f=f: models[f].log_likelihood(hdata[f][0], hdata[f][1])) for f in mine],
                                             nslots=hslots, device_index=local_rank)
                for f, v in zip(mine, vals):
                    joint[f] = v.to(joint.device)
                if world > 1:
                    dist.all_reduce(joint)
                return joint
            return step, F_total, n, D, d, scaling
        data = {}
        for f in mine:
            if d <= 16:   # host recipe: the CPU baseline leg reads the very same numbers
                X, Y = synthetic_xy(n, D, d, seed=f)
                data[f] = (torch.tensor(X, dtype=torch.float64, device=dev), torch.tensor(Y, dtype=torch.float64, device=dev))
            else:
                data[f] = synthetic_xy_device(n, D, d, f, dev)
            if with_grad:
                data[f][1].requires_grad_(True)
        w, amp, dadd = params(D, with_grad)
        nslots = max(1, min(args.slots, len(mine)))

        path = block_path(len(mine), args.no_chain_batch)

        def step():
            vals = {}
            if path == "single":
                f = mine[0]
                vals[f] = F.nlml(data[f][0], data[f][1], w, amp, diag_add=dadd, clamp=1e-30)
            elif path == "chain":   # the rank's blocks have one shape: ONE factorisation chain for all of them
                ctx = torch.enable_grad() if with_grad else torch.no_grad()
                with ctx:
                    out = F.nlml_many([data[f][0] for f in mine], [data[f][1] for f in mine], [w] * len(mine), [amp] * len(mine),
                                      [dadd] * len(mine), clamp=1e-30)
                for i, f in enumerate(mine):
                    vals[f] = out[i]
            elif path == "streams":   # several owned blocks overlap on this GPU
                ctx = torch.enable_grad() if with_grad else torch.no_grad()
                with ctx, F.concurrent_blocks(nslots=nslots, device_index=local_rank, lookahead=args.slot_lookahead) as cb:
                    for i, f in enumerate(mine):
                        with cb.slot(i):
                            vals[f] = F.nlml(data[f][0], data[f][1], w, amp, diag_add=dadd, clamp=1e-30, **F._slot_args())
            joint.zero_()
            for f, v in vals.items():
                joint[f] = v.detach()
            if world > 1:
                dist.all_reduce(joint)      # the joint NLML: one 8*F-byte sum over xGMI
            return joint
        return step, F_total, n, D, d, scaling

    def timed(step, steps, warmup):
        for _ in range(warmup):
            step()
        fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            joint = step()
        fence()
        dt = time.perf_counter() - t0
        t = torch.tensor([dt], dtype=torch.float64, device=red_dev)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t), joint.clone()

    # ---- the timed workload ------------------------------------------------------------------------------------
    step, F_total, n, D, d, scaling = make_workload(args.workload, args.n, args.D, args.d, args.blocks)
    for i in range(args.warmup):
        step()
        if i == 0 and not args.dry:
            for kv in args.opt:   # again, now that the handles of the concurrent slots exist
                key, val = kv.split("=")
                _lib.set_option(key, float(val), local_rank, all_slots=True)
    if not args.dry:
        # event pairs around every trailing-update launch, on the stream the kernel runs on, inside the timed region (no host
        # syncs): the contract's live roofline measurement; costs ~0.1 ms of the 29 ms step by A/B (tools/ab_forward.py)
        _lib.set_option("timing", 2, local_rank, all_slots=True)
        _lib.syrk_stats(reset=True, device_index=local_rank, all_slots=True)
    dt, joint = timed(step, args.steps, 0)
    stats, stages = {"flops": 0.0, "ms": 0.0, "launches": 0}, {}
    if not args.dry:
        stats = _lib.syrk_stats(reset=True, device_index=local_rank, all_slots=True)
        _lib.set_option("timing", 0, local_rank, all_slots=True)
        if WORKLOADS[args.workload][0] is None:
            _lib.set_option("timing", 1, local_rank)
            step()
            stages = _lib.last_timings(local_rank)
            _lib.set_option("timing", 0, local_rank)
    ms_per_step = dt / args.steps * 1e3
    ms_no_events = None
    if not args.dry and args.steps >= 4:
        # the same steps once more WITHOUT the per-launch event pairs of the live roofline measurement (timing = 0): what the event
        # pairs cost the timed region is stated, not assumed -- `value` stays the timed region's, events included
        dt0, _ = timed(step, args.steps, 0)
        ms_no_events = dt0 / args.steps * 1e3
    flops_step = nlml_flops(n, D, d) * (3.0 if args.with_grad else 1.0) * F_total   # fwd+bwd ~ N^3
    if args.workload == "gar8_hogp":
        flops_step = hogp_flops(n, HOGP_MODES if d == HOGP_MODES[0] * HOGP_MODES[1] else (d, 1)) * F_total
    value = flops_step / (dt / args.steps) / 1e9
    stock = args.n is None and args.D is None and args.d is None and args.blocks is None
    gpu_ref, vendor = {}, None

    # ---- the fixed-F sharding workloads, a few steps each (default run only) ----------------------------------------
    sharded = {}
    if args.workload == "headline" and not args.no_sharded and not args.with_grad and (stock or args.dry):
        del step
        if not args.dry:
            torch.cuda.empty_cache()
        for name in ("cigar4", "gar8", "gar8_hogp"):
            kw = dict(n=64, d=8) if args.dry else {}
            s_step, sF, sn, sD, sd, _ = make_workload(name, **kw)
            hog = name == "gar8_hogp"
            ssteps, swarm = (2, 1) if hog else (4, 3)     # (the first steps grow the workspaces of the concurrent slots)
            sdt, sjoint = timed(s_step, ssteps, swarm)
            sfl = hogp_flops(sn, HOGP_MODES if sd == HOGP_MODES[0] * HOGP_MODES[1] else (sd, 1)) if hog else nlml_flops(sn, sD, sd)
            sharded[name] = {"blocks": sF, "N": sn, "D": sD, "d": sd, "ms_per_step": round(sdt / ssteps * 1e3, 3),
                             "value": round(sfl * sF / (sdt / ssteps) / 1e9, 1), "unit": "GF/s", "scaling": "strong",
                             "blocks_per_rank": -(-sF // world), "owner": deal[name]["owner"],
                             "rank_paths": ["threads" if hog and p_ in ("chain", "streams") else p_ for p_ in deal[name]["rank_paths"]],
                             "blocks_in_flight_per_rank": min(args.hogp_slots if hog else args.slots, -(-sF // world)),
                             "config": "BASELINE configs[%d]" % WORKLOADS[name][4] + (
                                 " as HOGP blocks (d = %d x %d): eigh of the N x N input kernel on ffgp_syevd + mode products; PRIMARY "
                                 "figure: ms_per_step; `value` prices the executed flops (8.67 N^3 + 4 N^2 d per block, hogp_flops), "
                                 "`value_canonical_syevd` a LAPACK-style count (4.67 N^3 + 4 N^2 d)" % HOGP_MODES if hog else "")}
            if hog:    # HOGP_simple.log_likelihood returns +NLL / (N prod d) per block (hogp_simple.py:92-117): not a joint NLL
                sharded[name]["sum_block_loss"] = float(sjoint.sum())
                sharded[name]["primary"] = "ms_per_step"
                sharded[name]["value_canonical_syevd"] = round(
                    hogp_flops_canonical(sn, HOGP_MODES if sd == HOGP_MODES[0] * HOGP_MODES[1] else (sd, 1)) * sF / (sdt / ssteps) / 1e9, 1)
            else:
                sharded[name]["joint_nll"] = float(sjoint.sum())
            del s_step
            if not args.dry:
                torch.cuda.empty_cache()

    # ---- the reference's training iteration (forward + every gradient) at C2, C3 and configs[3], a few steps each ------------------
    train_step = {}
    if args.workload == "headline" and not args.no_sharded and not args.with_grad and stock and not args.dry:
        for name in ("c2", "headline", "cigar4"):
            t_step, tF, tn, tD, td, _ = make_workload(name, with_grad=True)
            tsteps, twarm = (10, 3) if name == "c2" else (4, 3)      # (eight hardware queues: the first steps of a chained leg place its streams)
            tdt, _ = timed(t_step, tsteps, twarm)
            tfl = 3.0 * nlml_flops(tn, tD, td) * tF          # forward + backward ~ N^3 + ...
            train_step[name] = {"blocks": tF, "N": tn, "D": tD, "d": td, "ms_per_step": round(tdt / tsteps * 1e3, 3),
                                "steps": tsteps, "warmup": twarm,
                                "value": round(tfl / (tdt / tsteps) / 1e9, 1), "unit": "GF/s",
                                "frac_of_mfma_peak": round(tfl / (tdt / tsteps) / 1e12 / world / FP64_MFMA_PEAK_TFLOPS, 4),
                                "what": "likelihood + the gradients loss.backward() leaves (Y, length scales, amplitude, noise) in one "
                                        "fused call; flops = 3 x the forward's (N^3 + 2 N^2 d + 4 N^2 D, SURVEY 8d)",
                                "config": "BASELINE configs[%d]" % WORKLOADS[name][4]}
            del t_step
            torch.cuda.empty_cache()

    # ---- the per-GPU unit of the sharded configs and C2, forward, one block per rank (what one GPU of an 8-GPU gar8 / 4-GPU cigar4
    #      run executes per step: the single-block F.nlml path) --------------------------------------------------------------------
    blocks_leg = {}
    if args.workload == "headline" and not args.no_sharded and not args.with_grad and stock and not args.dry:
        for key, (bn, bD, bd, cfg) in (("c2", (4096, 8, 1, 1)), ("n8192_d1", (8192, 8, 1, None)), ("n8192_d1024", (8192, 8, 1024, 3)),
                                       ("n8192_d4096", (8192, 8, 4096, 4))):
            b_step, bF, _, _, _, _ = make_workload("c2", n=bn, D=bD, d=bd, with_grad=False)
            bdt, _ = timed(b_step, 10, 3)
            bfl = nlml_flops(bn, bD, bd) * bF
            blocks_leg[key] = {"N": bn, "D": bD, "d": bd, "blocks_per_rank": 1, "ms_per_step": round(bdt / 10 * 1e3, 3),
                               "value": round(bfl / (bdt / 10) / 1e9, 1), "unit": "GF/s",
                               "frac_of_mfma_peak": round(bfl / (bdt / 10) / 1e12 / world / FP64_MFMA_PEAK_TFLOPS, 4),
                               "what": "forward NLML of ONE block per rank (10 steps after 3 warm-ups)" + (
                                   "; BASELINE configs[%d]%s" % (cfg, "" if cfg == 1 else ": the unit each GPU runs when the F blocks are "
                                                                "dealt one per GPU") if cfg is not None else "")}
            del b_step
            torch.cuda.empty_cache()

    # ---- the reference's training loop at the sizes its experiments run: K Adam steps per library call, small models in ONE launch ----
    train_small = {}
    if args.workload == "headline" and not args.no_sharded and not args.with_grad and stock and not args.dry:
        from fidelityfusion_amd import kernel as K_
        from fidelityfusion_amd.cigp_v10 import cigp, train_many
        for key, (tn, tD, td, tFm) in (("n128", (128, 5, 1, 1)), ("n32", (32, 5, 1, 1)), ("n64_x16", (64, 5, 1, 16))):
            ms_, xs_, ys_ = [], [], []
            for f in range(tFm):
                X, Y = synthetic_xy(tn, tD, td, seed=f)
                ms_.append(cigp(K_.ARDKernel(tD), 1.0).double().to(dev))
                xs_.append(torch.tensor(X, dtype=torch.float64, device=dev))
                ys_.append(torch.tensor(Y, dtype=torch.float64, device=dev))
            train_many(ms_, xs_, ys_, 5)
            fence()
            t0 = time.perf_counter()
            tr_, _ = train_many(ms_, xs_, ys_, 200)
            fence()
            tdt = time.perf_counter() - t0
            train_small[key] = {"N": tn, "D": tD, "d": td, "models": tFm, "steps": 200, "ms_per_call": round(tdt * 1e3, 3),
                                "us_per_model_step": round(tdt * 1e6 / 200 / tFm, 2), "final_loss": float(tr_[0, -1]),
                                "what": "cigp_v10.train_many -> ffgp_train_raw: 200 iterations of zero_grad / loss = -negative_log_likelihood / "
                                        "backward / Adam step (FidelityFusion_Models/ResGP.py:78-112) as ONE call, wall clock around the call"}

    if not args.dry and not args.no_cpu_baseline and world == 1 and stock and args.workload in ("headline", "c2"):
        # after every timed leg: the GPU side of the parity columns, and the vendor factorisation as a stated side number
        gpu_ref = gpu_parity_values(dev, ("c2", "headline") if args.workload == "headline" else ("c2",))
        if args.workload in gpu_ref and abs(gpu_ref[args.workload]["ll"] + float(joint[0])) > 1e-9 * abs(float(joint[0])):
            raise SystemExit("bench.py: the timed step's value %r and the drop-in module's %r differ" % (float(joint[0]), gpu_ref[args.workload]["ll"]))
        vendor = vendor_potrf_ms(dev)
    if rank == 0:
        achieved = stats["flops"] / (stats["ms"] * 1e-3) / 1e12 if stats["ms"] > 0 else 0.0
        traffic, traffic_src, traffic_err = None, None, None
        if not args.dry:
            try:    # a replayed annotation must never cost the measured line (or leave the other ranks in the barrier)
                traffic, traffic_src = recorded_traffic(n)
            except Exception as e:   # noqa: BLE001
                traffic_err = "%s: %s" % (type(e).__name__, e)
        cfg_idx = WORKLOADS[args.workload][4] if stock else None
        out = {
            "metric": "GP NLML+Cholesky throughput (NxN fp64 GF/s, %%MFMA-roofline) at N=%d" % n,
            "value": round(value, 1), "unit": "GF/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3),
            "ms_per_step_without_launch_events": None if ms_no_events is None else round(ms_no_events, 3),
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%s: %s NLML %s, ARD kernel, %d block%s of N=%d D=%d d=%d%s"
                                   % (args.workload, "single-fidelity cigp" if F_total == world and scaling == "weak" else
                                      ("per-fidelity HOGP_simple blocks (eigh on ffgp_syevd + mode products)," if args.workload == "gar8_hogp"
                                       else "per-fidelity cigp blocks,"),
                                      "forward+gradients" if args.with_grad else "forward", F_total, "" if F_total == 1 else "s", n, D, d,
                                      " (BASELINE configs[%d])" % cfg_idx if cfg_idx is not None else ""),
                       "N": n, "D": D, "d": d, "blocks": F_total, "owner": deal[args.workload]["owner"],
                       "rank_paths": deal[args.workload]["rank_paths"],
                       "blocks_in_flight_per_rank": (min(args.hogp_slots if args.workload == "gar8_hogp" else args.slots, -(-F_total // world))
                                                     if scaling == "strong" else 1),
                       "parallelism": "fidelity-shard x%d (LPT partition, one %d-byte all-reduce per step)" % (world, 8 * F_total)},
            "collective": coll,
            "cpu_affinity": (None if pinned is None else {"rank0_cpus": len(pinned), "first": pinned[0], "last": pinned[-1],
                                                          "source": "NUMA node of the rank's GPU (sysfs)" if gpu_numa_cpus(local_rank) else "even split"}),
            "rccl_ranks": coll["ranks"] if coll["backend"] == "nccl" else 0,
            "roofline": {"bound": "mfma", "achieved": round(achieved, 2), "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(achieved / FP64_MFMA_PEAK_TFLOPS, 4), "traffic": traffic,
                         "traffic_unit": "bytes/launch (PMC pass of this command, replayed from the committed profile)",
                         "traffic_source": traffic_src, "traffic_error": traffic_err,
                         "traffic_calibration": "2 x FETCH_SIZE checked on this kernel's own loads (profiles/r05a_traffic_calibration.txt): a "
                                                "k = 16 launch at m = 16384 reads 1082 MB of C tiles by construction and reports 2 x FETCH = "
                                                "1123 MB (TCC_EA0_RDREQ x 128 B, no 32-B requests); WRITE_SIZE exact (1074.0 vs 1073.7 MB)",
                         "algorithmic_bytes_per_launch": round(8.0 * stats["flops"] / max(stats["launches"], 1) / 512.0 * (1.0 + 1.0 / 16.0)),
                         "kernel": ROOFLINE_KERNEL + " (trailing SYRK update of the blocked Cholesky, K = 512)",
                         "launches": stats["launches"], "avg_launch_ms": round(stats["ms"] / max(stats["launches"], 1), 4),
                         "avg_launch_gflop": round(stats["flops"] / max(stats["launches"], 1) / 1e9, 3)},
            "whole_path_frac_of_mfma_peak": round(value / world / 1e3 / FP64_MFMA_PEAK_TFLOPS, 4),
            "stage_ms": {k: round(v, 3) for k, v in stages.items()},
            "nll": float(joint[0]), "joint_nll": float(joint.sum()),
        }
        if args.workload == "gar8_hogp":   # blocks return +NLL / (N prod d) each (hogp_simple.py:92-117); s/step is the primary figure
            out["sum_block_loss"] = out.pop("joint_nll")
            out.pop("nll")
            out["primary"] = "ms_per_step"
            out["value_canonical_syevd"] = round(hogp_flops_canonical(n, HOGP_MODES if d == HOGP_MODES[0] * HOGP_MODES[1] else (d, 1))
                                                 * F_total / (dt / args.steps) / 1e9, 1)
        if sharded:
            out["sharded"] = sharded
        if train_step:
            out["train_step"] = train_step
        if blocks_leg:
            out["blocks"] = blocks_leg
        if train_small:
            out["train_small"] = train_small
        if vendor is not None:
            out["vendor_potrf_ms"] = dict(vendor, note="torch.linalg.cholesky on this GPU (the reference's own .cuda() path, MFGP_ver2023May/"
                                          "mfgp_demo.py:88-94) vs ffgp_potrf on the same matrix; outside the timed region, never on the product path")
        if not args.no_cpu_baseline and world == 1 and not args.dry:   # reported on rank 0 of the single-GPU run only
            out["cpu_baseline"] = cpu_baseline(args.cpu_budget_s, gpu_ref)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))          # before torch is imported: the parent never initialises the GPU
    run_rank(args)


if __name__ == "__main__":
    main()
