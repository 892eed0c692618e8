import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from fidelityfusion_amd import eigh as E, _lib
n = 8192
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
X = torch.rand((n, 8), generator=g, device=dev, dtype=torch.float64)
d = torch.cdist(X, X); K = torch.exp(-0.5 * d * d); del d
AB, Y = E.sy2sb(K); dd, ee, refl = E.sb2st(AB)
Z = torch.eye(n, device=dev, dtype=torch.float64)
for dbg in (0, 1, 2, 8, 10, 4, 15):
    _lib.set_option("diag_dbg", dbg)
    E.ormq2(refl, Z); torch.cuda.synchronize()
    t0 = time.perf_counter(); E.ormq2(refl, Z); torch.cuda.synchronize()
    print("dbg", dbg, "ormq2 %.1f ms" % ((time.perf_counter() - t0) * 1e3), flush=True)
_lib.set_option("diag_dbg", 0)
