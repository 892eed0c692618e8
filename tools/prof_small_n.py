import cProfile, pstats, os, sys
sys.path.insert(0, os.getcwd())
import torch
from fidelityfusion_amd import kernel
from fidelityfusion_amd.cigp_v10 import cigp
torch.set_default_dtype(torch.float64)
dev="cuda"
n=128
X = torch.rand(n, 2, device=dev); Y = torch.sin(X.sum(1, keepdim=True)); m = cigp(kernel.ARDKernel(2), 1.0).to(dev)
def step():
    for p in m.parameters(): p.grad=None
    (-m.negative_log_likelihood(X, Y)).backward()
for _ in range(20): step()
torch.cuda.synchronize()
pr=cProfile.Profile(); pr.enable()
for _ in range(1000): step()
torch.cuda.synchronize()
pr.disable()
st=pstats.Stats(pr); st.sort_stats("tottime").print_stats(28)
