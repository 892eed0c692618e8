#!/usr/bin/env python3
"""Per-queue busy time of the last evaluation of a rocprofv3 --kernel-trace CSV in windows of W us (default 500):
queue_windows.py '<glob>' [W].  A column per hardware queue: kernel time started in the window (can exceed W when kernels overlap)."""
import csv
import glob
import os
import sys


def main(pattern, W=500.0):
    f = max(glob.glob(pattern, recursive=True), key=os.path.getmtime)
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    idx = [i for i, r in enumerate(rows) if 'assemble' in r['Kernel_Name']][-1]
    sub = rows[idx:]
    t0 = int(sub[0]['Start_Timestamp'])
    tend = max(int(r['End_Timestamp']) for r in sub)
    qs = sorted({r['Queue_Id'] for r in sub})
    print("total %.1f us; queues %s" % ((tend - t0) / 1e3, " ".join(qs)))
    nwin = int((tend - t0) / 1e3 / W) + 1
    for w in range(nwin):
        cells = []
        for q in qs:
            b = 0.0
            for r in sub:
                if r['Queue_Id'] != q:
                    continue
                s, e = (int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - t0) / 1e3
                lo, hi = max(s, w * W), min(e, (w + 1) * W)
                if hi > lo:
                    b += hi - lo
            cells.append("%7.1f" % b)
        print("%7.1f-%7.1f us: %s" % (w * W, (w + 1) * W, " ".join(cells)))


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 500.0)
