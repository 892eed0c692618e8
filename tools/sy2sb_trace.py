"""Only the band reduction (ffgp_sy2sb) of one kernel matrix, a few times: meant to sit under `rocprofv3 --kernel-trace --stats` so the
per-kernel totals are the stage's alone.  usage: python tools/sy2sb_trace.py [n=8192] [reps=3]; FFGP_EIGH_OPTS as in eigh_bench.py."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _devlib  # noqa: F401,E402  (development build: these switches are not in the shipped library)
import torch
from fidelityfusion_amd import eigh as E
from fidelityfusion_amd import _lib

for _kv in os.environ.get("FFGP_EIGH_OPTS", "").split(","):
    if _kv:
        _lib.set_option(_kv.split("=")[0], float(_kv.split("=")[1]))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
X = torch.rand((n, 8), generator=g, device=dev, dtype=torch.float64)
d = torch.cdist(X, X)
K = torch.exp(-0.5 * d * d)
del d
E.sy2sb(K)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    E.sy2sb(K)
torch.cuda.synchronize()
print("sy2sb n=%d: %.1f ms per call over %d calls (+1 warm-up call in the trace)" % (n, (time.perf_counter() - t0) / reps * 1e3, reps), flush=True)
