"""placement sweep of the bulge chase at N = 8192: option chase_pack (every pack-th workgroup works)"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _devlib  # noqa: F401,E402  (development build: these switches are not in the shipped library)
import torch
from fidelityfusion_amd import eigh as E, _lib
n = 8192
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
X = torch.rand((n, 8), generator=g, device=dev, dtype=torch.float64)
d = torch.cdist(X, X); K = torch.exp(-0.5 * d * d); del d
AB, Y = E.sy2sb(K)
for pack in (8, 4, 2, 1):
    _lib.set_option("chase_pack", pack)
    E.sb2st(AB); torch.cuda.synchronize()
    t0 = time.perf_counter(); E.sb2st(AB); torch.cuda.synchronize()
    print("pack", pack, "sb2st %.1f ms" % ((time.perf_counter() - t0) * 1e3), flush=True)
_lib.set_option("chase_pack", 0)
