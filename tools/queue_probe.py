import os, sys, time
sys.path.insert(0, "/root/repo")
q, res = int(sys.argv[1]), int(sys.argv[2])
from fidelityfusion_amd import _lib
_lib.configure_queues(max_hw_queues=q, reserve_worker_streams=res)
import torch
from bench import synthetic_xy
from fidelityfusion_amd import functional as F
torch.set_default_dtype(torch.float64)
dev = torch.device("cuda", 0)
n, D, d = 4096, 8, 1
X, Y = synthetic_xy(n, D, d, seed=0)
X = torch.tensor(X, device=dev); Y = torch.tensor(Y, device=dev)
w = torch.ones(D, device=dev, requires_grad=True); amp = torch.ones(1, device=dev, requires_grad=True)
dadd = torch.full((1,), 0.37, device=dev, requires_grad=True)
def step():
    for t in (w, amp, dadd): t.grad = None
    F.nlml(X, Y, w, amp, diag_add=dadd, clamp=1e-30).backward()
for _ in range(10): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(50): step()
torch.cuda.synchronize()
print("queues %d reserve %d: C2 training step %.3f ms" % (q, res, (time.perf_counter() - t0) / 50 * 1e3))
