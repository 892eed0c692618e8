import os, sys, time
sys.path.insert(0, "/root/repo")
import torch
from bench import synthetic_xy
from fidelityfusion_amd import kernel
from fidelityfusion_amd.cigp_v10 import cigp, train_many
torch.set_default_dtype(torch.float64)
dev = torch.device("cuda", 0)
for n, D, d in ((32, 5, 1), (64, 5, 1), (128, 5, 1), (128, 16, 4)):
    X, Y = synthetic_xy(n, D, d, seed=0)
    m = cigp(kernel.ARDKernel(D), 1.0).to(dev)
    x, y = torch.tensor(X, device=dev), torch.tensor(Y, device=dev)
    train_many([m], [x], [y], 5)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    train_many([m], [x], [y], 200)
    torch.cuda.synchronize()
    print(n, D, d, "200 steps %.2f ms" % ((time.perf_counter() - t0) * 1e3), flush=True)
