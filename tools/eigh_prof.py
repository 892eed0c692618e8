"""one ffgp_syevd call at N = 8192 (kernel matrix, D = 8) for rocprofv3 --kernel-trace --stats"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fidelityfusion_amd import eigh as E
from fidelityfusion_amd import _lib
for _kv in os.environ.get("FFGP_OPTS", "").split(","):      # library options, "k=v,k=v"
    if _kv:
        _lib.set_option(_kv.split("=")[0], float(_kv.split("=")[1]), 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
X = torch.rand((n, 8), generator=g, device=dev, dtype=torch.float64)
d = torch.cdist(X, X)
K = torch.exp(-0.5 * d * d)
del d
for _ in range(2):
    W, Z = E.eigh(K)
torch.cuda.synchronize()
print(float(W[-1]))
