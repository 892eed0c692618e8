#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV of one bench.py run: per outer step of the last factorisation,
the trailing-update launches (main queue) against the look-ahead chain (side queue)."""
import csv
import glob
import os
import sys


def dur(r):
    return (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3


def main(pattern):
    f = max(glob.glob(pattern), key=os.path.getmtime)
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    idx = [i for i, r in enumerate(rows) if 'assemble' in r['Kernel_Name']][-1]
    sub = rows[idx:]
    t0 = int(sub[0]['Start_Timestamp'])
    syrk = [r for r in sub if 'Li1ELi1E' in r['Kernel_Name'] or '<0, 0, 1, 1' in r['Kernel_Name']]
    mainq = syrk[0]['Queue_Id']
    aux = [r for r in sub if r['Queue_Id'] != mainq]
    print("file", f, "n syrk", len(syrk), "n aux", len(aux))
    for i in range(0, len(syrk) - 1, 2):
        a, b = syrk[i], syrk[i + 1]
        nxt = int(syrk[i + 2]['Start_Timestamp']) if i + 2 < len(syrk) else 1 << 62
        ch = [r for r in aux if int(a['End_Timestamp']) <= int(r['Start_Timestamp']) < nxt]
        if not ch:
            continue
        span = (int(ch[-1]['End_Timestamp']) - int(ch[0]['Start_Timestamp'])) / 1e3
        d = [round(dur(r)) for r in ch if 'diag' in r['Kernel_Name']]
        g = [round(dur(r)) for r in ch if 'gemm' in r['Kernel_Name']]
        print("step %2d t=%8.1f S_i=%6.1f S_ii=%7.1f chain=%7.1f gap_to_next=%6.1f diag=%s gemms=%s" % (
            i // 2, (int(a['Start_Timestamp']) - t0) / 1e3, dur(a), dur(b), span,
            (nxt - max(int(b['End_Timestamp']), int(ch[-1]['End_Timestamp']))) / 1e3 if nxt < (1 << 61) else 0, d, g))
    print("total_us", (int(sub[-1]['End_Timestamp']) - t0) / 1e3)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else '/root/repo/gpurun_out/prof_la/*/*kernel_trace.csv')
