"""A few forward evaluations (assembly -> factorisation -> solves -> NLML) of one block, for a kernel trace:
    rocprofv3 --kernel-trace -d OUT -o t --output-format csv -- python3 tools/forward_trace_target.py N D [reps] [d]
then tools/chain_timeline.py / gap_report.py / syrk_phase_account.py on OUT/*kernel_trace.csv (they take the last evaluation)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from bench import synthetic_xy
from fidelityfusion_amd import functional as F

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
D = int(sys.argv[2]) if len(sys.argv) > 2 else 8
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 6
d = int(sys.argv[4]) if len(sys.argv) > 4 else 1
dev = torch.device("cuda", 0)
for kv in os.environ.get("FFGP_OPTS", "").split(","):      # library options, "k=v,k=v"
    if kv:
        from fidelityfusion_amd import _lib
        _lib.set_option(kv.split("=")[0], float(kv.split("=")[1]), 0)
X, Y = synthetic_xy(n, D, d, seed=0)
Xd, Yd = torch.tensor(X, device=dev), torch.tensor(Y, device=dev)
w = torch.ones(D, dtype=torch.float64, device=dev)
amp = torch.ones(1, dtype=torch.float64, device=dev)
dadd = torch.tensor([np.exp(-1.0) + 1e-6], dtype=torch.float64, device=dev)
with torch.no_grad():
    for _ in range(reps):
        F.nlml(Xd, Yd, w, amp, diag_add=dadd, clamp=1e-30)
torch.cuda.synchronize()
