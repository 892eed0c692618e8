"""kernel durations of the diagonal-block kernels alone (n = 128, 64, 32: one block per call) and inside a chain (n = 2048):
   cd /tmp && rocprofv3 --kernel-trace --stats -d out -- python3 tools/diag4_prof.py"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from fidelityfusion_amd import _lib

dev = torch.device("cuda:0")
h = _lib.handle(0)
lib = _lib.lib
for n in (128, 2048):
    g = torch.Generator(device=dev).manual_seed(n)
    R = torch.randn(n, 64, dtype=torch.float64, device=dev, generator=g)
    S = R @ R.T / 64.0
    S.diagonal().add_(2.0)
    for v4 in (0, 1):
        _lib.set_option("diag_v4", v4, 0)
        for _ in range(100 if n == 128 else 20):
            W = S.clone()
            rc = lib.ffgp_potrf_rows(h, C.c_void_p(W.data_ptr()), n, n, n)
            assert rc == 0
torch.cuda.synchronize()
