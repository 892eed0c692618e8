"""The headline forward (N = 16384 by default) with the host idle and with every CPU of the box busy, issued launch by launch and as
one captured graph (option fwd_graph): python tools/host_load_probe.py [N [busy_processes]].  The busy processes are plain Python spin
loops started with multiprocessing (no GPU use); they stand in for `stress-ng --cpu`."""
import multiprocessing as mp
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def spin(stop):
    x = 0
    while not stop.is_set():
        for _ in range(100000):
            x += 1


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
    nbusy = int(sys.argv[2]) if len(sys.argv) > 2 else 2 * len(os.sched_getaffinity(0))
    ctx = mp.get_context("spawn")          # children must never inherit a GPU context
    stop = ctx.Event()
    import numpy as np
    import torch
    from bench import synthetic_xy
    from fidelityfusion_amd import _lib
    from fidelityfusion_amd import functional as F
    dev = torch.device("cuda", 0)
    D = 16
    X, Y = synthetic_xy(n, D, 1, seed=0)
    Xd, Yd = torch.tensor(X, device=dev), torch.tensor(Y, device=dev)
    w = torch.ones(D, dtype=torch.float64, device=dev)
    amp = torch.ones(1, dtype=torch.float64, device=dev)
    dadd = torch.tensor([np.exp(-1.0) + 1e-6], dtype=torch.float64, device=dev)

    def run(steps):
        with torch.no_grad():
            for _ in range(3):
                v = F.nlml(Xd, Yd, w, amp, diag_add=dadd, clamp=1e-30)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                v = F.nlml(Xd, Yd, w, amp, diag_add=dadd, clamp=1e-30)
            torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps * 1e3, float(v)

    steps = 20 if n >= 8192 else 100
    print("N = %d, %d CPUs usable, %d busy processes" % (n, len(os.sched_getaffinity(0)), nbusy), flush=True)
    res = {}
    for load in (0, 1, 0, 1):
        procs = []
        if load:
            procs = [ctx.Process(target=spin, args=(stop,), daemon=True) for _ in range(nbusy)]
            for p in procs:
                p.start()
            time.sleep(2.0)
        for graph in (0, 1):
            _lib.set_option("fwd_graph", float(graph), 0)
            ms, v = run(steps)
            res.setdefault((load, graph), []).append(ms)
            print("  host %s  %s: %8.3f ms/step   value %.12g" % ("busy" if load else "idle", "one graph launch" if graph else "launch by launch",
                                                                 ms, v), flush=True)
        if load:
            stop.set()
            for p in procs:
                p.join(timeout=10)
            stop.clear()
    _lib.set_option("fwd_graph", 0.0, 0)
    h = _lib.handle(0)
    print("  replays served: %d" % _lib.lib.ffgp_graph_replays(h))
    for k in sorted(res):
        print("host %s, %s: %s ms" % ("busy" if k[0] else "idle", "graph" if k[1] else "plain", " / ".join("%.3f" % x for x in res[k])))


if __name__ == "__main__":
    main()
