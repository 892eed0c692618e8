"""A three-fidelity residual GP of our own making, trained the way FidelityFusion_Models/ResGP.py:78-112 trains one -- a cigp per
fidelity on the residual targets, 200 Adam iterations each -- in two ways on the MI355X: the reference's per-iteration loop through the
drop-in modules, and `cigp_v10.train_many`, which runs the same iterations of ALL fidelities in one library call per model (likelihood,
closed-form gradients and Adam's update on the device; the fidelities train side by side).  Same losses, same parameters.

python examples/resgp_train_many.py        (needs an MI355X: the fused training call has no CPU path)
"""
import copy
import math
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from fidelityfusion_amd import kernel
from fidelityfusion_amd.cigp_v10 import cigp, train_many

torch.set_default_dtype(torch.float64)
dev = torch.device("cuda", 0)
gen = torch.Generator().manual_seed(7)


def truth(x, level):                              # three levels of one 2-D function: each adds finer structure to the one below
    base = torch.sin(3.0 * x[:, :1]) * torch.cos(2.0 * x[:, 1:2])
    mid = 0.4 * torch.sin(9.0 * x[:, :1] + 1.0)
    fine = 0.15 * torch.cos(17.0 * x[:, 1:2]) * x[:, :1]
    return base + (mid if level >= 1 else 0.0) + (fine if level >= 2 else 0.0)


sizes = (300, 300, 250)                           # ragged, as in the reference's own demo (ResGP.py:121-136)
xs = [torch.rand(n, 2, generator=gen) for n in sizes]
# residual targets of a subset design: fidelity f learns what fidelity f - 1 leaves over (here: against the known lower level)
ys = [truth(xs[0], 0)] + [truth(xs[f], f) - truth(xs[f], f - 1) for f in (1, 2)]
ys = [y + 0.02 * torch.randn(y.shape, generator=gen) for y in ys]
xs, ys = [x.to(dev) for x in xs], [y.to(dev) for y in ys]
models = [cigp(kernel.ARDKernel(2), 1.0).to(dev) for _ in sizes]
twins = [copy.deepcopy(m) for m in models]
steps, lr = 200, 1e-2

train_many([copy.deepcopy(m) for m in models], xs, ys, 3, lr=lr)      # (warm-up: workspaces, streams)
torch.cuda.synchronize()
t0 = time.perf_counter()
trace, _ = train_many(models, xs, ys, steps, lr=lr)
torch.cuda.synchronize()
t_many = time.perf_counter() - t0

t0 = time.perf_counter()
ref = torch.zeros(len(twins), steps)
for f, (m, x, y) in enumerate(zip(twins, xs, ys)):            # the reference's loop, fidelity after fidelity
    opt = torch.optim.Adam(m.parameters(), lr=lr)
    for i in range(steps):
        opt.zero_grad()
        loss = -m.negative_log_likelihood(x, y)
        loss.backward()
        opt.step()
        ref[f, i] = float(loss.detach())
torch.cuda.synchronize()
t_loop = time.perf_counter() - t0

print("3 fidelities (%s points), %d Adam steps each" % (" / ".join(map(str, sizes)), steps))
print("  train_many          %7.1f ms    final losses %s" % (t_many * 1e3, ["%.4f" % v for v in trace[:, -1].tolist()]))
print("  per-iteration loop  %7.1f ms    final losses %s" % (t_loop * 1e3, ["%.4f" % v for v in ref[:, -1].tolist()]))
print("  largest relative difference of the loss traces: %.1e" % float(((trace.cpu() - ref).abs() / ref.abs().clamp_min(1e-300)).max()))
for f, (m, t) in enumerate(zip(models, twins)):
    d = max(float((a.detach() - b.detach()).abs().max()) for a, b in zip(m.parameters(), t.parameters()))
    print("  fidelity %d: noise 1/beta = %.4f, length scales %s, largest parameter difference %.1e"
          % (f, math.exp(-float(m.log_beta.detach())), ["%.3f" % abs(float(v)) for v in m.kernel.length_scales.detach()], d))
