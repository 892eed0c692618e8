#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc counter_collection CSVs (one directory per pass) into one JSON: per counter, per kernel
{launches, sum, per_launch}.   python tools/pmc_summary.py OUT.json DIR [DIR ...]"""
import csv
import glob
import json
import os
import sys


def main(out, dirs):
    res = {}
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                c = res.setdefault(r["Counter_Name"], {})
                k = c.setdefault(r["Kernel_Name"], {"launches": 0, "sum": 0.0})
                k["launches"] += 1
                k["sum"] += float(r["Counter_Value"])
    for c in res.values():
        for k in c.values():
            k["per_launch"] = k["sum"] / max(1, k["launches"])
    doc = {
        "command": "rocprofv3 --pmc <counters> --output-format csv -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline "
                   "(separate passes: FETCH_SIZE | WRITE_SIZE | SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE)",
        "units": "FETCH_SIZE/WRITE_SIZE in KiB as reported by rocprofv3; on gfx950 FETCH_SIZE counts 64 B per 128-B request "
                 "for 16-B/lane streams: double it (MI355X_MICROARCH.md, HBM)",
        "counters": res,
    }
    json.dump(doc, open(out, "w"), indent=1)
    syrk = [k for k in res.get("FETCH_SIZE", {}) if "<0, 0, 1, 1, 128, 128>" in k]
    for k in syrk:
        f = res["FETCH_SIZE"][k]["per_launch"] * 1024 * 2
        w = res.get("WRITE_SIZE", {}).get(k, {}).get("per_launch", 0.0) * 1024
        print("SYRK per launch: fetch(x2) %.1f MB, write %.1f MB, launches %d" % (f / 1e6, w / 1e6, res["FETCH_SIZE"][k]["launches"]))
        mb = res.get("SQ_VALU_MFMA_BUSY_CYCLES", {}).get(k)
        ga = res.get("GRBM_GUI_ACTIVE", {}).get(k)
        if mb and ga:   # MFMA-busy cycles are summed over the 1024 SIMDs, GRBM_GUI_ACTIVE over the 8 XCDs
            util = mb["sum"] / (ga["sum"] / 8.0 * 1024.0)
            doc["syrk_mfma_busy_fraction"] = util
            print("SYRK MFMA-busy fraction of SIMD cycles = %.3f" % util)
    json.dump(doc, open(out, "w"), indent=1)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2:])
