"""Interleaved A/B of library options on the C3 forward (one process, same box): python tools/ab_forward.py "k=v,k=v" "k=v" ..."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _devlib  # noqa: F401,E402  (development build: these switches are not in the shipped library)
import numpy as np
import torch

from bench import synthetic_xy
from fidelityfusion_amd import _lib
from fidelityfusion_amd import functional as F

n, D = int(os.environ.get("AB_N", "16384")), 16
dev = torch.device("cuda", 0)
X, Y = synthetic_xy(n, D, 1, seed=0)
Xd = torch.tensor(X, device=dev)
Yd = torch.tensor(Y, device=dev)
w = torch.ones(D, dtype=torch.float64, device=dev)
amp = torch.ones(1, dtype=torch.float64, device=dev)
dadd = torch.tensor([np.exp(-1.0) + 1e-6], dtype=torch.float64, device=dev)
configs = [dict(kv.split("=") for kv in c.split(",") if kv) for c in sys.argv[1:]]
keys = sorted({k for c in configs for k in c})
defaults = {"nb_big": 0, "nb_big_until": 0, "polite_m": 6144, "split_rem_max": 180, "band_log2": 3, "nb_outer": 512,
            "tile32_threshold": 1024, "polite_pad_kb": 40, "small_tile_threshold": 640, "la_split": 1, "aux_prio": 1, "diag_v2": 4, "la_carry": 2, "lookahead": 1, "la_min_n": 1024, "tail_mask_m": 0, "tail_mask_cus": 8, "pass_split_min": 0, "syrk_h64": 0, "syrk_direct": 0, "trsm128": 1, "polite64_pad_kb": 60, "polite32_pad_kb": 46, "ho_values": 1, "ho_defer": 2, "la_carry_rows": 8192, "diag_excl_rows": 4096, "la_carry_n": 12288}
res = {i: [] for i in range(len(configs))}
for rnd in range(4):
    for i, c in enumerate(configs):
        for k in keys:
            _lib.set_option(k, float(c.get(k, defaults[k])), 0)
        for _ in range(2):
            F.nlml(Xd, Yd, w, amp, diag_add=dadd, clamp=1e-30)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            nll = F.nlml(Xd, Yd, w, amp, diag_add=dadd, clamp=1e-30)
        torch.cuda.synchronize()
        res[i].append((time.perf_counter() - t0) / 10 * 1e3)
vals = []
for i, c in enumerate(configs):
    for k in keys:
        _lib.set_option(k, float(c.get(k, defaults[k])), 0)
    vals.append(float(F.nlml(Xd, Yd, w, amp, diag_add=dadd, clamp=1e-30)))
print("values:", " ".join("%.15g" % v for v in vals), "| max rel spread %.2e" % ((max(vals) - min(vals)) / abs(vals[0])))
for i, c in enumerate(configs):
    print("%-44s min %.3f  med %.3f ms  (%s)" % (c or "defaults", min(res[i]), float(np.median(res[i])), " ".join("%.2f" % v for v in res[i])))
