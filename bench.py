#!/usr/bin/env python3
"""Headline benchmark: GP NLML + Cholesky throughput (fp64) on MI355X -- BASELINE.json's metric on its config C3
(single-fidelity cigp, ARD kernel, N = 16384, D = 16, d = 1).

A step = one pass of the hot path over one GP block: covariance assembly -> blocked Cholesky (Y^T riding as a
passenger row, so Gamma = L^-1 Y comes out of the factorisation's own GEMMs) -> log-det / ||Gamma||^2 reductions
-> NLML scalar, with X, Y and the hyper-parameters already resident in HBM.  With N > 1 ranks every rank runs one
such block per step (per-fidelity sharding: independent blocks, no data-path collective) and the F = N per-block
values are summed by ONE 8*N-byte all-reduce (RCCL) per step -- weak scaling.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--n 16384] [--D 16] [--d 1]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FP64_MFMA_PEAK_TFLOPS = 78.6   # MI355X fp64 matrix peak (vendor spec; 256 CU x 4 SIMD x 32 FLOP/clk x 2.4 GHz)
ROOFLINE_KERNEL = "ffgp_gemm_f64<0, 0, 1, 1, 128, 128>"   # trailing SYRK update of the blocked Cholesky


def recorded_traffic(n):
    """HBM bytes per launch of the roofline kernel from the committed rocprofv3 PMC passes of this same command
    (profiles/*_pmc_summary.json; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for 16-B/lane streams on
    gfx950).  bench.py cannot collect PMC counters itself; null when no recorded pass matches this workload."""
    import glob
    if n != 16384:
        return None, None
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_summary.json")))
    if not files:
        return None, None
    try:
        c = json.load(open(files[-1]))["counters"]
        key = [k for k in c["FETCH_SIZE"] if k.startswith("void " + ROOFLINE_KERNEL)][0]
        byt = (2.0 * c["FETCH_SIZE"][key]["per_launch"] + c["WRITE_SIZE"][key]["per_launch"]) * 1024.0
        return byt, os.path.relpath(files[-1], ROOT)
    except Exception:
        return None, None


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


def nlml_flops(n, D, d):
    """N^3/3 (Cholesky) + N^2 d (Gamma = L^-1 Y) + 2 N^2 D (distance contractions)."""
    return n ** 3 / 3.0 + float(n) * n * d + 2.0 * n * n * D


def cpu_baseline(D, d, n_sample):
    """The oracle ("port" of the reference's torch-CPU path) timed on this box's host cores on a bounded sample."""
    import numpy as np
    from oracle import gp_oracle as O
    try:
        from threadpoolctl import threadpool_info
        cores = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:
        cores = os.cpu_count() or 1
    X, Y = synthetic_xy(n_sample, D, d, seed=0)
    ls, sv, lb = np.ones(D), [1.0], [1.0]
    O.nlml_forward_ard(X[:512], Y[:512], ls, sv, lb)   # warm the BLAS threads
    t0 = time.perf_counter()
    ll = O.nlml_forward_ard(X, Y, ls, sv, lb)
    dt = time.perf_counter() - t0
    return {"value": round(nlml_flops(n_sample, D, d) / dt / 1e9, 2), "unit": "GF/s", "cores": int(cores), "kind": "port",
            "sample": "1 NLML forward of the oracle at N=%d D=%d d=%d (%.2f s), numpy/OpenBLAS" % (n_sample, D, d, dt),
            "ll": float(ll)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--n", type=int, default=16384)
    ap.add_argument("--D", type=int, default=16)
    ap.add_argument("--d", type=int, default=1)
    ap.add_argument("--cpu-sample-n", type=int, default=8192)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--with-grad", action="store_true", help="time forward + closed-form gradients instead")
    ap.add_argument("--opt", action="append", default=[], help="library option key=value (development A/B runs)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus

    from fidelityfusion_amd import _lib
    from fidelityfusion_amd import functional as F
    for kv in args.opt:   # development switches (tools/): --opt la_split=1 --opt tile32_threshold=0 ...
        key, val = kv.split("=")
        _lib.set_option(key, float(val), local_rank)

    n, D, d = args.n, args.D, args.d
    # one independent block per rank (fidelity f = rank): same shape, different seed
    X, Y = synthetic_xy(n, D, d, seed=rank)
    Xd = torch.tensor(X, dtype=torch.float64, device=dev)
    Yd = torch.tensor(Y, dtype=torch.float64, device=dev, requires_grad=args.with_grad)
    # reference initial hyper-parameters: length_scales = 1 (kernel.py:84), signal_variance = 1, log_beta = 1 (ResGP.py:27)
    w = torch.ones(D, dtype=torch.float64, device=dev, requires_grad=args.with_grad)
    amp = torch.ones(1, dtype=torch.float64, device=dev, requires_grad=args.with_grad)
    dadd = torch.tensor([np.exp(-1.0) + 1e-6], dtype=torch.float64, device=dev, requires_grad=args.with_grad)
    joint = torch.zeros(world, dtype=torch.float64, device=dev)

    def step():
        nll = F.nlml(Xd, Yd, w, amp, diag_add=dadd, clamp=1e-30)
        if world > 1:
            joint.zero_()
            joint[rank] = nll.detach()
            dist.all_reduce(joint)      # the joint NLML: one 8*F-byte sum over xGMI
        return nll

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    _lib.set_option("timing", 2, local_rank)     # event pairs around every trailing-update launch (no host syncs)
    _lib.syrk_stats(reset=True, device_index=local_rank)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        nll = step()
    fence()
    dt = time.perf_counter() - t0
    stats = _lib.syrk_stats(reset=True, device_index=local_rank)
    _lib.set_option("timing", 1, local_rank)
    step()
    stages = _lib.last_timings(local_rank)
    _lib.set_option("timing", 0, local_rank)

    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax)
    ms_per_step = dt / args.steps * 1e3
    flops_step = nlml_flops(n, D, d) * (3.0 if args.with_grad else 1.0)  # fwd+bwd ~ N^3
    value = flops_step * world / (dt / args.steps) / 1e9

    if rank == 0:
        achieved = stats["flops"] / (stats["ms"] * 1e-3) / 1e12 if stats["ms"] > 0 else 0.0
        traffic, traffic_src = recorded_traffic(n)
        out = {
            "metric": "GP NLML+Cholesky throughput (NxN fp64 GF/s, %%MFMA-roofline) at N=%d" % n,
            "value": round(value, 1), "unit": "GF/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "single-fidelity cigp NLML %s, ARD kernel, N=%d D=%d d=%d per GPU%s"
                                   % ("forward+gradients" if args.with_grad else "forward", n, D, d,
                                      " (BASELINE configs[2])" if (n, D, d) == (16384, 16, 1) else
                                      " (BASELINE configs[1])" if (n, D, d) == (4096, 8, 1) else ""),
                       "N": n, "D": D, "d": d, "blocks": world, "parallelism": "fidelity-shard x%d" % world},
            "roofline": {"bound": "mfma", "achieved": round(achieved, 2), "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(achieved / FP64_MFMA_PEAK_TFLOPS, 4), "traffic": traffic,
                         "traffic_unit": "bytes/launch (PMC, recorded)", "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": round(8.0 * stats["flops"] / max(stats["launches"], 1) / 512.0 * (1.0 + 1.0 / 16.0)),
                         "kernel": ROOFLINE_KERNEL + " (trailing SYRK update of the blocked Cholesky, K = 512)",
                         "launches": stats["launches"], "avg_launch_ms": round(stats["ms"] / max(stats["launches"], 1), 4),
                         "avg_launch_gflop": round(stats["flops"] / max(stats["launches"], 1) / 1e9, 3)},
            "whole_path_frac_of_mfma_peak": round(value / world / 1e3 / FP64_MFMA_PEAK_TFLOPS, 4),
            "stage_ms": {k: round(v, 3) for k, v in stages.items()},
            "nll": float(nll),
        }
        if not args.no_cpu_baseline and world == 1:   # reported on rank 0 of the single-GPU run only
            out["cpu_baseline"] = cpu_baseline(D, d, min(args.cpu_sample_n, n))
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
