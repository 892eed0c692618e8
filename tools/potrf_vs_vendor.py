#!/usr/bin/env python3
"""The reference's own GPU path next to this library's, as a stated side number.

The 2023 API of the reference moves its tensors with `.cuda()` (MFGP_ver2023May/mfgp_demo.py:88-94), so its
`torch.linalg.cholesky` (base_gp/cigp.py:83,129) is the vendor solver on this very GPU.  This tool times that call and
`ffgp_potrf` (functional.cholesky) on the same kernel matrices: single host thread, idle GPU, N in {1024, 4096, 8192, 16384},
min of 5 after a warm-up, whole call (the vendor path synchronises to read its status; so does the library).  The vendor path
is never on the product path.  Also prints the factor's agreement.

    python tools/potrf_vs_vendor.py [--sizes 1024,4096,8192,16384] > gpurun_out/potrf_vs_vendor.json
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from fidelityfusion_amd import functional as F  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", default="1024,4096,8192,16384")
    ap.add_argument("--reps", type=int, default=5)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    out = {"device": torch.cuda.get_device_name(0), "torch": torch.__version__, "sizes": {}}
    for n in [int(x) for x in a.sizes.split(",")]:
        g = torch.Generator(device=dev).manual_seed(n)
        X = torch.rand((n, 8), generator=g, device=dev, dtype=torch.float64)
        S = torch.exp(-0.5 * torch.cdist(X, X) ** 2)
        S.diagonal().add_(0.368)

        def t(fn):
            r = fn()
            torch.cuda.synchronize()
            best = 1e9
            for _ in range(a.reps):
                t0 = time.perf_counter()
                fn()
                torch.cuda.synchronize()
                best = min(best, time.perf_counter() - t0)
            return best * 1e3, r

        tv, Lv = t(lambda: torch.linalg.cholesky(S))
        tf, Lf = t(lambda: F.cholesky(S))
        fl = n ** 3 / 3.0
        out["sizes"][str(n)] = {"vendor_potrf_ms": round(tv, 3), "ffgp_potrf_ms": round(tf, 3), "vendor_tflops": round(fl / tv / 1e9, 2),
                                "ffgp_tflops": round(fl / tf / 1e9, 2), "speedup": round(tv / tf, 2),
                                "factor_rel_diff": float((torch.tril(Lf) - Lv).abs().max() / Lv.abs().max())}
        del X, S, Lv, Lf
        torch.cuda.empty_cache()
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
