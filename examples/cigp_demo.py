"""A small drop-in tour: the reference's `cigp` module and kernel classes, imported from this package, on a 2-D test surface of
our own (a damped ripple) -- fit the hyper-parameters by maximum likelihood, report held-out error and the calibration of the
predictive variance.

python examples/cigp_demo.py [cuda]      `cuda`: model and tensors live on the MI355X (no per-call transfers)
"""
import math
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from fidelityfusion_amd import kernel
from fidelityfusion_amd.cigp_v10 import cigp

where = "cuda" if sys.argv[1:] == ["cuda"] else "cpu"
gen = torch.Generator().manual_seed(2026)


def ripple(x):                                   # the surface: exp(-r) cos(3 pi r) around (0.3, 0.6), shifted off zero
    r = (x - torch.tensor([0.3, 0.6])).norm(dim=1, keepdim=True)
    return torch.exp(-r) * torch.cos(3.0 * math.pi * r) + 2.0


x_fit = torch.rand(160, 2, generator=gen)
y_fit = ripple(x_fit) + 0.05 * torch.randn(160, 1, generator=gen)
x_new = torch.rand(400, 2, generator=gen)
y_new = ripple(x_new)

gp = cigp(kernel=kernel.ARDKernel(2), log_beta=2.0).to(where)
x_fit, y_fit, x_new, y_new = (t.to(where) for t in (x_fit, y_fit, x_new, y_new))
opt = torch.optim.Adam(gp.parameters(), lr=5e-2)
history, clock = [], time.perf_counter()
for it in range(150):
    opt.zero_grad()
    nll = -gp.negative_log_likelihood(x_fit, y_fit)     # (the reference's function returns +LL: callers negate)
    nll.backward()
    opt.step()
    history.append(float(nll.detach()))
elapsed = time.perf_counter() - clock
print("150 likelihood + gradient steps on %s: %.2f s;  nll %.3f -> %.3f" % (where, elapsed, history[0], history[-1]))
print("length scales", [round(float(v), 3) for v in gp.kernel.length_scales.detach().abs()], " noise sd %.3f" % math.exp(-0.5 * float(gp.log_beta.detach())))

with torch.no_grad():
    mean, cov = gp.forward(x_fit, y_fit, x_new)
    sd = cov.diagonal().clamp_min(0).sqrt().unsqueeze(1)
    z = (y_new - mean) / sd
print("held-out rmse %.4f;  |error| <= 2 sd on %.0f %% of 400 points" % (float((mean - y_new).pow(2).mean().sqrt()), 100.0 * float((z.abs() <= 2).float().mean())))
