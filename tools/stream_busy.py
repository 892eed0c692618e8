#!/usr/bin/env python3
"""Per-queue busy time of the last factorisation in a rocprofv3 --kernel-trace CSV, split into outer-step windows."""
import csv
import glob
import os
import sys


def main(pattern):
    f = max(glob.glob(pattern, recursive=True), key=os.path.getmtime)
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    idx = [i for i, r in enumerate(rows) if 'assemble' in r['Kernel_Name']][-1]
    sub = rows[idx:]
    t0 = int(sub[0]['Start_Timestamp'])
    tend = max(int(r['End_Timestamp']) for r in sub)
    print("total_us %.1f" % ((tend - t0) / 1e3))
    qs = {}
    for r in sub:
        qs.setdefault(r['Queue_Id'], []).append(r)
    for q, rs in qs.items():
        busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in rs) / 1e3
        names = {}
        for r in rs:
            k = r['Kernel_Name'].replace('ffgp_', '').replace('(GemmArgs)', '').replace('void ', '')[:34]
            d = names.setdefault(k, [0, 0.0])
            d[0] += 1
            d[1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        print("queue %s: %d kernels, busy %.1f us" % (q, len(rs), busy))
        for k, (c, d) in sorted(names.items(), key=lambda kv: -kv[1][1]):
            print("    %-36s n=%4d  %9.1f us  avg %7.1f" % (k, c, d, d / c))
    # windows of 2 ms: busy fraction of each queue
    W = 2000.0
    nwin = int((tend - t0) / 1e3 / W) + 1
    print("window(ms)  " + "  ".join("q%s" % q[-3:] for q in qs))
    for w in range(nwin):
        a, b = t0 + w * W * 1e3, t0 + (w + 1) * W * 1e3
        fr = []
        for q, rs in qs.items():
            t = 0
            for r in rs:
                s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
                t += max(0, min(e, b) - max(s, a))
            fr.append(t / (W * 1e3))
        print("%4.0f-%4.0f   " % (w * W / 1e3, (w + 1) * W / 1e3) + "  ".join("%.2f" % x for x in fr))


if __name__ == "__main__":
    main(sys.argv[1])
