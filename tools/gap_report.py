#!/usr/bin/env python3
"""Idle gaps of the main (trailing-update) queue during the last factorisation of a rocprofv3 --kernel-trace CSV:
where the trailing update waited for the look-ahead chain.  Usage: gap_report.py '<glob of *_kernel_trace.csv>'"""
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
    # the main queue is the one that runs the tagged SYRK instantiation (<0, 0, 1, 1, 128, 128>)
    mainq = [r['Queue_Id'] for r in sub if 'Li0ELi0ELi1ELi1ELi128ELi128' in r['Kernel_Name'] or '<0, 0, 1, 1, 128, 128>' in r['Kernel_Name']]
    mainq = max(set(mainq), key=mainq.count)
    main = [r for r in sub if r['Queue_Id'] == mainq]
    other = [r for r in sub if r['Queue_Id'] != mainq]
    print("total %.1f us; main queue %s: %d kernels, other queues: %d kernels" % ((tend - t0) / 1e3, mainq, len(main), len(other)))
    busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in main) / 1e3
    print("main busy %.1f us, idle %.1f us" % (busy, (tend - t0) / 1e3 - busy))
    prev = t0
    acc = []
    for r in main:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        gap = (s - prev) / 1e3
        name = r['Kernel_Name'].replace('ffgp_', '').replace('(GemmArgs)', '').replace('void ', '')[:44]
        acc.append((gap, (s - t0) / 1e3, (e - s) / 1e3, r.get('Grid_Size_X', r.get('Grid_Size', '?')), name))
        prev = max(prev, e)
    # windows of 2 ms
    W = 2000.0
    nwin = int((tend - t0) / 1e3 / W) + 1
    for w in range(nwin):
        g = sum(a[0] for a in acc if w * W <= a[1] < (w + 1) * W)
        b = sum(a[2] for a in acc if w * W <= a[1] < (w + 1) * W)
        ob = sum((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in other
                 if w * W <= (int(r['Start_Timestamp']) - t0) / 1e3 < (w + 1) * W)
        print("window %5.1f-%5.1f ms: main busy %7.1f us, main gaps %7.1f us, side-queue kernel time %7.1f us" % (w * W / 1e3, (w + 1) * W / 1e3, b, g, ob))
    if len(sys.argv) > 2:
        for a in acc:
            if a[0] > float(sys.argv[2]):
                print("gap %.1f us before t=%.1f dur=%.1f grid=%s %s" % a)


if __name__ == "__main__":
    main(sys.argv[1])
