"""Repeat sequential vs concurrent evaluation of a few GP blocks and report any disagreement (race hunting)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from fidelityfusion_amd import _lib
from fidelityfusion_amd import functional as F
from fidelityfusion_amd import kernel
from fidelityfusion_amd.cigp_v10 import cigp

torch.set_default_dtype(torch.float64)
DEV = "cuda:0"
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
for kv in sys.argv[2:]:
    k, v = kv.split("=")
    for slot in range(4):
        _lib.lib.ffgp_set_option(_lib.handle(0, slot), k.encode(), float(v))
rng = np.random.default_rng(0)
blocks = []
for f, (n, D, d) in enumerate([(900, 4, 3), (1300, 6, 1), (700, 3, 8), (1100, 5, 2)]):
    X = rng.random((n, D))
    Y = np.sin(2 * np.pi * X @ rng.random((D, d))) + 0.1 * rng.standard_normal((n, d))
    blocks.append((torch.tensor(X, device=DEV), torch.tensor(Y, device=DEV), D))
models = [cigp(kernel.ARDKernel(D), 0.7).to(DEV) for (_, _, D) in blocks]


def run(concurrent, grad):
    losses = [None] * len(blocks)
    ctx = torch.enable_grad() if grad else torch.no_grad()
    with ctx:
        if concurrent:
            with F.concurrent_blocks(nslots=3) as cb:
                for f, m in enumerate(models):
                    with cb.slot(f):
                        losses[f] = -m.negative_log_likelihood(blocks[f][0], blocks[f][1])
        else:
            for f, m in enumerate(models):
                losses[f] = -m.negative_log_likelihood(blocks[f][0], blocks[f][1])
    torch.cuda.synchronize()
    return np.array([float(l) for l in losses])


ref = run(False, False)
bad = 0
for mode in ("seq", "conc", "seq+grad", "conc+grad"):
    nbad = 0
    for r in range(reps):
        v = run(mode.startswith("conc"), mode.endswith("grad"))
        if not np.allclose(v, ref, rtol=1e-12, atol=0):
            nbad += 1
            if nbad <= 3:
                print("  %s rep %d: diff %s" % (mode, r, (v - ref)))
    print("%s: %d / %d runs disagree with the first sequential run" % (mode, nbad, reps))
    bad += nbad
print("TOTAL disagreements:", bad)
