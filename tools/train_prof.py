"""Profile target: K Adam steps of ONE N = 128 model in one library call (for rocprofv3 --kernel-trace --stats: which kernels a step is made of).
python tools/train_prof.py [N [steps]]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from bench import synthetic_xy
from fidelityfusion_amd import kernel
from fidelityfusion_amd.cigp_v10 import cigp, train_many

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
dev = torch.device("cuda", 0)
torch.set_default_dtype(torch.float64)
X, Y = synthetic_xy(n, 5, 1, seed=0)
m = cigp(kernel.ARDKernel(5), 1.0).to(dev)
x, y = torch.tensor(X, device=dev), torch.tensor(Y, device=dev)
train_many([m], [x], [y], 5)
torch.cuda.synchronize()
trace, _ = train_many([m], [x], [y], steps)
torch.cuda.synchronize()
print("loss", float(trace[0, 0]), "->", float(trace[0, -1]))
