"""One repeated posterior query at N=16384 (for a rocprofv3 --kernel-trace timeline)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fidelityfusion_amd import kernel
from fidelityfusion_amd.cigp_v10 import cigp
torch.set_default_dtype(torch.float64)
dev = "cuda:0"
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
g = torch.Generator(device=dev).manual_seed(0)
X = torch.rand((n, 16), generator=g, device=dev); Y = torch.randn((n, 1), generator=g, device=dev)
Xs = torch.rand((256, 16), generator=g, device=dev)
m = cigp(kernel.ARDKernel(16), 1.0).to(dev)
with torch.no_grad():
    for _ in range(4):
        m(X, Y, Xs)
    torch.cuda.synchronize()
