"""Near-singular Sigma (tiny noise): the blocked factorisation against LAPACK through the oracle."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from fidelityfusion_amd import kernel
from fidelityfusion_amd.cigp_v10 import cigp
from oracle import gp_oracle as O

torch.set_default_dtype(torch.float64)
rng = np.random.default_rng(0)
for n, lb in [(700, 8.0), (2000, 12.0), (2000, 16.0), (3000, 20.0)]:
    X = rng.random((n, 3))
    Y = np.sin(4 * X.sum(1, keepdims=True)) + 0.01 * rng.standard_normal((n, 1))
    ls = np.array([0.7, 0.9, 1.1])
    k = kernel.ARDKernel(3)
    with torch.no_grad():
        k.length_scales.copy_(torch.tensor(ls))
    m = cigp(k, lb).to("cuda")
    Yt = torch.tensor(Y, device="cuda", requires_grad=True)
    try:
        ll = m.negative_log_likelihood(torch.tensor(X, device="cuda"), Yt)
        ll.backward()
        got = (float(ll), float(m.log_beta.grad), k.length_scales.grad.cpu().numpy())
    except torch.linalg.LinAlgError as e:
        got = "LinAlgError: " + str(e)[:60]
    try:
        ll_ref, gr = O.cigp_ll_and_grads(X, Y, ls, [1.0], [lb])
        S = O.sigma_cigp(O.ard_kernel(X, X, ls, [1.0]), [lb])
        cond = np.linalg.cond(S)
        ref = (ll_ref, float(gr["log_beta"]), gr["length_scales"])
    except Exception as e:  # noqa
        ref, cond = "oracle failed: " + repr(e)[:60], float("nan")
    if isinstance(got, tuple) and isinstance(ref, tuple):
        print("n=%d log_beta=%.0f cond(Sigma)=%.1e: LL rel diff %.2e, dLL/dlog_beta rel diff %.2e, dLL/dls rel diff %.2e" % (
            n, lb, cond, abs(got[0] - ref[0]) / abs(ref[0]), abs(got[1] - ref[1]) / abs(ref[1]),
            np.abs(got[2] - ref[2]).max() / np.abs(ref[2]).max()))
    else:
        print("n=%d log_beta=%.0f cond=%.1e: hip -> %s ; oracle -> %s" % (n, lb, cond, got if not isinstance(got, tuple) else "ok", ref if not isinstance(ref, tuple) else "ok"))
