#!/usr/bin/env python3
"""Kernel time of ONE shared-chain call (ffgp_nlml_fused_batch), by kernel and by queue.

  run (under the profiler):  rocprofv3 --kernel-trace -d gpurun_out/bt -o bt --output-format csv -- python3 tools/batch_trace.py run [N [F [grad]]]
  read:                      python3 tools/batch_trace.py read 'gpurun_out/bt/**/*kernel_trace.csv'

`run` does three warm calls, sleeps 80 ms and does the one measured call, so `read` finds it as everything after the largest
gap between kernel starts."""
import csv
import glob
import os
import re
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(n, nF, grad):
    import torch
    from bench import synthetic_xy
    from fidelityfusion_amd import kernel
    from fidelityfusion_amd.cigp_v10 import cigp, negative_log_likelihood_many
    D, d = 8, 1
    dev = torch.device("cuda", 0)
    torch.set_default_dtype(torch.float64)
    models, xs, ys = [], [], []
    for f in range(nF):
        X, Y = synthetic_xy(n, D, d, seed=f)
        models.append(cigp(kernel.ARDKernel(D), 1.0).to(dev))
        xs.append(torch.tensor(X, device=dev))
        ys.append(torch.tensor(Y, device=dev))

    for kv in os.environ.get("BT_OPTS", "").split():
        from fidelityfusion_amd import _lib
        k, v = kv.split("=")
        _lib.set_option(k, float(v), 0)

    def call():
        ctx = torch.enable_grad() if grad else torch.no_grad()
        with ctx:
            v = negative_log_likelihood_many(models, xs, ys)
            if grad:
                v.sum().backward()
        torch.cuda.synchronize()

    for _ in range(3):
        call()
    time.sleep(0.08)
    t0 = time.perf_counter()
    call()
    print("measured call: %.3f ms wall (%d x N=%d, %s)" % ((time.perf_counter() - t0) * 1e3, nF, n, "fwd+grad" if grad else "forward"))


def short(name):
    name = re.sub(r"\(.*$", "", name)
    name = re.sub(r"^void ", "", name)
    return name[:70]


def read(pattern):
    f = max(glob.glob(pattern, recursive=True), key=os.path.getmtime)
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    gaps = [(int(rows[i]["Start_Timestamp"]) - int(rows[i - 1]["End_Timestamp"]), i) for i in range(1, len(rows))]
    cut = max(gaps)[1]
    sub = rows[cut:]
    t0 = int(sub[0]["Start_Timestamp"])
    tend = max(int(r["End_Timestamp"]) for r in sub)
    print("last call: %d kernels, %.3f ms from first start to last end" % (len(sub), (tend - t0) / 1e6))
    by = {}
    for r in sub:
        k = short(r["Kernel_Name"])
        e = by.setdefault(k, [0, 0.0])
        e[0] += 1
        e[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    for k, (c, us) in sorted(by.items(), key=lambda kv: -kv[1][1]):
        print("  %8.1f us  %5d x  %s" % (us, c, k))
    qs = {}
    for r in sub:
        q = qs.setdefault(r["Queue_Id"], [0, 0.0, int(r["Start_Timestamp"]), 0])
        q[0] += 1
        q[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        q[3] = max(q[3], int(r["End_Timestamp"]))
    for qid, (c, us, s, e) in sorted(qs.items()):
        print("  queue %s: %d kernels, busy %.1f us, active %.1f .. %.1f us" % (qid, c, us, (s - t0) / 1e3, (e - t0) / 1e3))
    if os.environ.get("BT_SEQ"):          # the launches of every queue in order: start, duration, gap to the previous end on that queue
        lo, hi = [float(x) for x in os.environ["BT_SEQ"].split(":")]
        last = {}
        for r in sub:
            s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            qid = r["Queue_Id"]
            gap = (s - last[qid]) / 1e3 if qid in last else 0.0
            last[qid] = e
            if lo <= (s - t0) / 1e6 <= hi:
                print("  q%s %8.1f us  +%6.1f  gap %6.1f  %s  grid %s" % (qid, (s - t0) / 1e3, (e - s) / 1e3, gap, short(r["Kernel_Name"])[:44],
                                                                        r.get("Grid_Size", "?")))
    # timeline in 20 slices: which kernels were running
    nsl = 20
    for i in range(nsl):
        a = t0 + (tend - t0) * i // nsl
        b = t0 + (tend - t0) * (i + 1) // nsl
        occ = {}
        for r in sub:
            s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            ov = min(e, b) - max(s, a)
            if ov > 0:
                k = short(r["Kernel_Name"])[:34]
                occ[k] = occ.get(k, 0) + ov
        top = sorted(occ.items(), key=lambda kv: -kv[1])[:3]
        print("  %6.2f-%6.2f ms  " % ((a - t0) / 1e6, (b - t0) / 1e6) + "  ".join("%s %.0f%%" % (k, 100.0 * v / (b - a)) for k, v in top))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(int(sys.argv[2]) if len(sys.argv) > 2 else 4096, int(sys.argv[3]) if len(sys.argv) > 3 else 8,
            (sys.argv[4] if len(sys.argv) > 4 else "1") != "0")
    else:
        read(sys.argv[2])
