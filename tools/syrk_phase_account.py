#!/usr/bin/env python3
"""What the chip delivers while the trailing updates run, from a rocprofv3 --kernel-trace CSV of one forward evaluation
(tools/forward_trace_target.py N D):

    syrk_phase_account.py '<glob of *_kernel_trace.csv>' N [panel width = 512]

The bench line's `roofline.frac` divides the 128-tile SYRK's flops by its in-situ duration.  In situ the look-ahead chain's
kernels (diagonal blocks, TRSMs, the panel's own updates, the strip of the next panel) run on the same CUs at the same time,
and their flops are not in the numerator.  This tool counts both over the window [first SYRK starts, last SYRK ends]:

    SYRK p (main queue)   : (m_p - 2 nb)^2 nb flops     (m_p = N - p nb rows are left when panel p starts; the count bench.py uses:
                            the trailing matrix without the next panel's own columns)
    strip p (chain)       : nb^2 (2 m_p - 3 nb)         (the rest of update p: the next panel's nb columns)
    panel p + 1 (chain)   : (m_p - nb) nb^2 - 2/3 nb^3  (diagonal blocks, TRSMs and the updates inside the panel)
    sum over p = the factorisation's m^3 / 3.

With look-ahead, strip p and panel p + 1 run under SYRK p: the window holds all three for p = 0 .. (launches - 1).
"""
import csv
import glob
import os
import sys

PEAK = 78.6e12   # fp64 MFMA, MI355X


def main(pattern, n, nb=512):
    f = max(glob.glob(pattern, recursive=True), key=os.path.getmtime)
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    first = [i for i, r in enumerate(rows) if 'assemble' in r['Kernel_Name']][-1]
    sub = rows[first:]
    syrk = [r for r in sub if '<0, 0, 1, 1, 128, 128>' in r['Kernel_Name'] or 'Li0ELi0ELi1ELi1ELi128ELi128' in r['Kernel_Name']]
    if not syrk:
        sys.exit("no tagged SYRK launch in the last evaluation")
    t0 = int(syrk[0]['Start_Timestamp'])
    t1 = int(syrk[-1]['End_Timestamp'])
    nsy = len(syrk)
    dur = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in syrk]
    gaps = [(int(b['Start_Timestamp']) - int(a['End_Timestamp'])) / 1e3 for a, b in zip(syrk, syrk[1:])]
    m = [n - p * nb for p in range(n // nb + 2)]
    syrk_fl = [(m[p] - 2 * nb) ** 2 * nb for p in range(nsy)]
    chain_fl = [nb * nb * (2 * m[p] - 3 * nb) + (m[p] - nb) * nb * nb - 2.0 / 3.0 * nb ** 3 for p in range(nsy)]
    in_window_syrk = sum(syrk_fl)
    in_window_chain = sum(chain_fl)
    window = (t1 - t0) / 1e3
    others = [r for r in sub if t0 <= int(r['Start_Timestamp']) < t1 and r not in syrk]
    other_time = sum((min(int(r['End_Timestamp']), t1) - int(r['Start_Timestamp'])) / 1e3 for r in others)
    print("file %s" % f)
    print("N = %d, panel width %d: %d tagged SYRK launches (squares of %d ... %d rows)" % (n, nb, nsy, m[0] - 2 * nb, m[nsy - 1] - 2 * nb))
    print("window (first SYRK starts -> last SYRK ends)   %9.1f us" % window)
    print("  SYRK kernel time (sum of durations)          %9.1f us   average %.1f us" % (sum(dur), sum(dur) / nsy))
    print("  gaps between consecutive SYRKs               %9.1f us   (largest %.1f)" % (sum(gaps), max(gaps) if gaps else 0.0))
    print("  other kernels started inside the window      %9d      their kernel time %.1f us (they overlap the SYRKs)" % (len(others), other_time))
    print("flops in the window: trailing updates %.4f TF, the panels factored under them %.4f TF (%.1f %% on top)" %
          (in_window_syrk / 1e12, in_window_chain / 1e12, 100.0 * in_window_chain / in_window_syrk))
    k_frac = in_window_syrk / (sum(dur) * 1e-6) / PEAK
    chip = (in_window_syrk + in_window_chain) / (window * 1e-6) / PEAK
    print("SYRK flops / SYRK kernel time                  %6.1f TF/s = %.3f of peak   (the bench line's roofline.frac)" % (in_window_syrk / sum(dur) / 1e6, k_frac))
    print("all flops in the window / window               %6.1f TF/s = %.3f of peak   (what the chip delivers while the SYRKs run)" %
          ((in_window_syrk + in_window_chain) / window / 1e6, chip))
    tend = max(int(r['End_Timestamp']) for r in sub)
    tbeg = int(sub[0]['Start_Timestamp'])
    print("evaluation %.1f us: before the window %.1f, window %.1f, after it (chain-bound tail, solves) %.1f" %
          ((tend - tbeg) / 1e3, (t0 - tbeg) / 1e3, window, (tend - t1) / 1e3))
    return syrk_fl, dur


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]), int(sys.argv[3]) if len(sys.argv) > 3 else 512)
