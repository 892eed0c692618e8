"""Randomised check of the blocked factorisation's stream choreography: factor random SPD matrices of random sizes (with and
without passenger rows, both look-ahead forms, look-ahead off) many times and compare every factor with torch's; a missed
cross-stream dependency shows up as a rare large residual.  python tools/potrf_fuzz.py [trials] [nmax] [seed]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from fidelityfusion_amd import _lib
from fidelityfusion_amd._lib import check, lib

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 200
nmax = int(sys.argv[2]) if len(sys.argv) > 2 else 6000
rng = np.random.default_rng(int(sys.argv[3]) if len(sys.argv) > 3 else 0)
dev = torch.device("cuda", 0)
h = _lib.handle(0)
_lib.bind_stream(h, 0)
worst = 0.0
for t in range(trials):
    n = int(rng.integers(1, nmax))
    rows = int(rng.choice([0, 1, 3, 64, 200]))
    carry = int(rng.choice([0, 1, 2]))
    la = int(rng.choice([0, 1, 1, 1]))
    nb = int(rng.choice([256, 512, 512, 768]))
    hv, hd = int(rng.choice([0, 1, 1])), int(rng.choice([0, 1, 2, 2]))      # hand-offs: event pairs / values; publication deferred to the next kernels or not
    rows_co, n_co = int(rng.choice([0, 1500, 3000, 8192])), int(rng.choice([0, 2000, 12288]))      # change-over of the look-ahead form
    for k, v in (("la_carry", carry), ("lookahead", la), ("nb_outer", nb), ("la_min_n", int(rng.choice([0, 1024, 3584]))), ("ho_values", hv),
                 ("ho_defer", hd), ("la_carry_rows", rows_co), ("la_carry_n", n_co)):
        _lib.set_option(k, v, 0)
    g = torch.Generator(device=dev).manual_seed(int(rng.integers(1 << 30)))
    X = torch.rand((n, 6), generator=g, device=dev, dtype=torch.float64)
    S = torch.exp(-0.5 * torch.cdist(X, X) ** 2) + 0.3 * torch.eye(n, device=dev, dtype=torch.float64)
    B = torch.randn((rows, n), generator=g, device=dev, dtype=torch.float64)
    ld = (n + 15) // 16 * 16
    W = torch.zeros((n + rows, ld), device=dev, dtype=torch.float64)
    W[:n, :n] = S
    W[n:, :n] = B
    check(lib.ffgp_potrf_rows(h, C.c_void_p(W.data_ptr()), n, n + rows, ld), "ffgp_potrf_rows")
    L = torch.linalg.cholesky(S)
    err = float((torch.tril(W[:n, :n]) - L).abs().max())
    if rows:
        want = torch.linalg.solve_triangular(L, B.T, upper=False).T
        err = max(err, float((W[n:, :n] - want).abs().max() / (1.0 + want.abs().max())))
    worst = max(worst, err)
    if err > 1e-9:
        print("MISMATCH trial %d n=%d rows=%d carry=%d lookahead=%d nb_outer=%d ho=%d/%d co=%d/%d: %.3e" % (t, n, rows, carry, la, nb, hv, hd, rows_co, n_co, err))
    if t % 50 == 49:
        print("  ... %d trials, worst %.3e" % (t + 1, worst), flush=True)
for k, v in (("la_carry", 2), ("lookahead", 1), ("nb_outer", 512), ("la_min_n", 1024), ("ho_values", 1), ("ho_defer", 2), ("la_carry_rows", 8192),
             ("la_carry_n", 12288)):
    _lib.set_option(k, v, 0)
print("potrf_fuzz: %d trials, n < %d, worst |L - L_torch| = %.3e" % (trials, nmax, worst))
