"""The trailing update's SYRK alone on the chip, at exactly the sizes of the headline factorisation's launches
(N = 16384, panel width 512: squares of 15360 ... 4608 rows, k = 512) -- the figure tools/syrk_phase_account.py's
"all flops in the window / window" is to be read against.  Usage: syrk_standalone.py [N] [nb] [launches]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from fidelityfusion_amd import _lib
from tools.gemm_bench import p, timeit

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 512
launches = int(sys.argv[3]) if len(sys.argv) > 3 else 22
PEAK = 78.6e12
h = _lib.handle(0)
_lib.bind_stream(h, 0)
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
A = torch.rand((n, nb), generator=g, device=dev, dtype=torch.float64) - 0.5
Cm = torch.zeros((n, n), device=dev, dtype=torch.float64)
tot_t = tot_f = 0.0
for q in range(launches):
    m = n - (q + 2) * nb
    fn = lambda: _lib.lib.ffgp_gemm(h, 0, 0, 1, 0, p(A), nb, p(A), nb, p(Cm), n, m, m, nb, -1.0, 1.0)
    fn()
    tmin, tmed = timeit(fn, rounds=9)
    fl = float(m) * m * nb
    tot_t += tmed
    tot_f += fl
    print("launch %2d  m = %5d  %.4f ms (median of 9; min %.4f)  %.1f TF/s = %.3f of peak" % (q, m, tmed, tmin, fl / tmed / 1e9, fl / tmed / 1e9 / (PEAK / 1e12)))
print("sum over %d launches: %.3f ms for %.4f TF = %.1f TF/s = %.3f of peak  (same flop count as the bench line: m^2 k)" %
      (launches, tot_t, tot_f / 1e12, tot_f / tot_t / 1e9, tot_f / tot_t / 1e9 / (PEAK / 1e12)))
