"""Randomised race hunt: GP blocks of random (n, D, d) evaluated concurrently on 2-4 handles must reproduce their
sequential values bit for bit (values and gradients)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from fidelityfusion_amd import functional as F
from fidelityfusion_amd import kernel
from fidelityfusion_amd.cigp_v10 import cigp

torch.set_default_dtype(torch.float64)
DEV = "cuda:0"
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
BIG = "big" in sys.argv[3:]
LOOSE = "loose" in sys.argv[3:]   # the slots' default schedule (no look-ahead of their own): equal to 1e-11, not bit for bit
from fidelityfusion_amd import _lib
for kv in sys.argv[3:]:
    if "=" in kv:
        key, val = kv.split("=")
        for slot in range(6):
            _lib.lib.ffgp_set_option(_lib.handle(0, slot), key.encode(), float(val))
bad = 0
for r in range(rounds):
    nb = int(rng.integers(2, 6))
    blocks, models = [], []
    for f in range(nb):
        n = int(rng.choice([rng.integers(130, 700), rng.integers(700, 2600), 128 * rng.integers(2, 18) + rng.integers(1, 127)]))
        if BIG:
            n = int(rng.integers(3000, 9500))
        D, d = int(rng.integers(1, 9)), int(rng.choice([1, 2, 3, 8, 40, 130]))
        X = rng.random((n, D))
        Y = np.sin(2 * np.pi * X @ rng.random((D, d))) + 0.1 * rng.standard_normal((n, d))
        blocks.append((torch.tensor(X, device=DEV), torch.tensor(Y, device=DEV, requires_grad=True)))
        models.append(cigp(kernel.ARDKernel(D), 0.7).to(DEV))

    def run(concurrent):
        for m in models:
            for p in m.parameters():
                p.grad = None
        for _, y in blocks:
            y.grad = None
        losses = [None] * nb
        if concurrent:
            # (look-ahead on in the slots as well: the sequential reference runs on the default handle with look-ahead, and
            #  only the same kernel sequence reproduces it bit for bit)
            with F.concurrent_blocks(nslots=int(rng.integers(2, 5)), lookahead=not LOOSE) as cb:
                for f, m in enumerate(models):
                    with cb.slot(f):
                        losses[f] = -m.negative_log_likelihood(*blocks[f])
        else:
            for f, m in enumerate(models):
                losses[f] = -m.negative_log_likelihood(*blocks[f])
        torch.stack(losses).sum().backward()
        torch.cuda.synchronize()
        out = [l.detach().cpu().numpy() for l in losses]
        out += [m.kernel.length_scales.grad.cpu().numpy().copy() for m in models]
        out += [y.grad.cpu().numpy().copy() for _, y in blocks]
        return out

    print("round", r, [(b[0].shape[0], b[1].shape[1]) for b in blocks], flush=True)
    ref = run(False)
    for rep in range(4):
        got = run(True)
        for i, (a, b) in enumerate(zip(ref, got)):
            if not (np.allclose(a, b, rtol=1e-11, atol=1e-11 * np.abs(a).max()) if LOOSE else np.array_equal(a, b)):
                bad += 1
                print("round %d rep %d item %d (block %d, n=%d d=%d): max abs diff %.3e" % (
                    r, rep, i, i % nb, blocks[i % nb][0].shape[0], blocks[i % nb][1].shape[1], np.abs(a - b).max()))
                break
print("rounds %d, mismatches %d" % (rounds, bad))
