#!/usr/bin/env python3
"""Print the kernel timeline (start, duration, queue, grid) of the last factorisation in a rocprofv3 --kernel-trace CSV."""
import csv
import glob
import os
import sys


def main(pattern, lo, hi):
    f = max(glob.glob(pattern, recursive=True), key=os.path.getmtime)
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    idx = [i for i, r in enumerate(rows) if 'assemble' in r['Kernel_Name']][-1]
    sub = rows[idx:]
    t0 = int(sub[0]['Start_Timestamp'])
    qs = {}
    prev_end = {}
    print("file", f, "kernels", len(sub), "total_us", (int(sub[-1]['End_Timestamp']) - t0) / 1e3)
    for i, r in enumerate(sub[lo:hi]):
        q = qs.setdefault(r['Queue_Id'], len(qs))
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        name = r['Kernel_Name'].replace('ffgp_', '').replace('(GemmArgs)', '').replace('void ', '')[:40]
        gap = (s - prev_end[q]) / 1e3 if q in prev_end else 0.0
        prev_end[q] = e
        print("%4d q%d t=%9.1f dur=%7.1f gap=%6.1f grid=%6s %s" % (lo + i, q, (s - t0) / 1e3, (e - s) / 1e3, gap,
                                                                 r.get('Grid_Size_X', r.get('Grid_Size', '?')), name))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]), int(sys.argv[3]))
