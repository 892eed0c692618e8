import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fidelityfusion_amd import kernel
from fidelityfusion_amd.gp_basic import GP_basic
torch.set_default_dtype(torch.float64)
dev = "cuda:0"
n, D = 16384, 16
g = torch.Generator(device=dev).manual_seed(0)
X = torch.rand((n, D), generator=g, device=dev); Y = torch.randn((n, 1), generator=g, device=dev)
m = GP_basic(kernel.ARDKernel(D), 0.6).to(dev)
with torch.no_grad():
    for _ in range(3):
        m.log_likelihood(X, Y)
torch.cuda.synchronize()
