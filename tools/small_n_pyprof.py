"""cProfile of the Python side of one small-N training step (module -> raw-parameter call -> backward): where the host time goes."""
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from fidelityfusion_amd import kernel
from fidelityfusion_amd.cigp_v10 import cigp

torch.set_default_dtype(torch.float64)
dev = torch.device("cuda", 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
X = torch.rand(n, 2, device=dev)
Y = torch.sin(X.sum(1, keepdim=True)) + 0.05 * torch.rand(n, 1, device=dev)
m = cigp(kernel.ARDKernel(2), 1.0).to(dev)


def step():
    for p in m.parameters():
        p.grad = None
    (-m.negative_log_likelihood(X, Y)).backward()


for _ in range(20):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(reps):
    step()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(22)
