"""The reference's own single-output demo (GaussianProcess/cigp_v10.py:73-101), with the import switched.
python examples/cigp_demo.py        tensors and model stay on the CPU, as the reference writes them: every call copies its
                                    inputs / parameters to the MI355X and the results back
python examples/cigp_demo.py cuda   model.to("cuda") and device tensors: no transfers
(16 training points: ~0.9 ms per Adam step either way -- launch and Python overhead; the first step pays ~0.45 s of set-up)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from fidelityfusion_amd import kernel                     # reference: import kernel
from fidelityfusion_amd.cigp_v10 import cigp              # reference: from cigp_v10 import cigp

torch.manual_seed(1)
xte = torch.linspace(0, 6, 100).view(-1, 1)
yte = torch.sin(xte) + 10
xtr = torch.rand(16, 1) * 6
ytr = torch.sin(xtr) + torch.randn(16, 1) * 0.5 + 10

kernel1 = kernel.SumKernel(kernel.LinearKernel(1), kernel.MaternKernel(1))   # two descriptors, one tile pass (csrc/pair.hip)
model = cigp(kernel=kernel1, log_beta=1.0)
if len(sys.argv) > 1 and sys.argv[1] == "cuda":
    model = model.to("cuda")
    xtr, ytr, xte, yte = (t.to("cuda") for t in (xtr, ytr, xte, yte))
optimizer = torch.optim.Adam(model.parameters(), lr=1e-1)
t0 = time.time()
for i in range(100):
    if i == 1:
        t1 = time.time()        # (the first step loads the code objects and allocates the handle's workspaces)
    optimizer.zero_grad()
    loss = -model.negative_log_likelihood(xtr, ytr)
    loss.backward()
    optimizer.step()
    if i % 20 == 0 or i == 99:
        print("iter", i, "nll:{:.5f}".format(loss.item()))
print("first step {:.3f} s, then {:.2f} ms per step".format(t1 - t0, (time.time() - t1) / 99 * 1e3))
with torch.no_grad():
    ypred, ypred_var = model.forward(xtr, ytr, xte)
rmse = float((ypred - yte).pow(2).mean().sqrt())
print("prediction at 100 points: rmse {:.3f}, mean predictive sd {:.3f}".format(rmse, float(ypred_var.diag().clamp_min(0).sqrt().mean())))
