"""Thin harness reproducing the CALL PATTERN of the reference's multi-fidelity trainers on the drop-in GP blocks
(SURVEY section 8a row X1).  The model logic of the reference (data managers, non-subset fill, Tensor_linear,
Matrix_Mapping) is out of scope; what is mirrored here is exactly how those models drive the hot path, so that
parity and benchmarks exercise it the way the reference does:

  * `train_gp_blocks`  -- the per-fidelity Adam loops of `train_ResGP` / `train_AR` / `train_CIGAR`
                          (FidelityFusion_Models/ResGP.py:67-112): `loss = -gpr.negative_log_likelihood(x, y | [y, y_var])`,
                          `loss.backward()`, `optimizer.step()`, a fresh Adam over ALL parameters per fidelity;
  * `resgp_predict`    -- `ResGP.forward` (ResGP.py:31-65): sum of per-fidelity posterior means and covariances;
  * `ResGP2023`        -- the 2023 joint loss `loss = sum_f cigp_list[f].compute_loss(x, res_f)` with the fixed-rho
                          residual chain (MFGP_ver2023May/ResGP.py:200-246, multiscale_coupling/Residual.py:9-33) and
                          its `forward` (:145-171), aligned / subset regime (one shared x).
"""
import torch

from .mfgp2023 import CIGP


def train_gp_blocks(gpr_list, data, max_iter=100, lr_init=1e-2, callback=None):
    """gpr_list[f]: module with negative_log_likelihood(x, y); data[f] = (x, y) or (x, [y, y_var])."""
    params = [p for m in gpr_list for p in m.parameters()]
    trace = []
    for f, gpr in enumerate(gpr_list):
        optimizer = torch.optim.Adam(params, lr=lr_init)
        x, y = data[f]
        for i in range(max_iter):
            optimizer.zero_grad()
            loss = -gpr.negative_log_likelihood(x, y)
            loss.backward()
            optimizer.step()
            trace.append(float(loss.detach()))
            if callback is not None:
                callback(f, i, trace[-1])
    return trace


@torch.no_grad()
def resgp_predict(gpr_list, data, x_test):
    mean = cov = None
    for gpr, (x, y) in zip(gpr_list, data):
        m, c = gpr(x, y, x_test)
        mean = m if mean is None else mean + m
        cov = c if cov is None else cov + c
    return mean, cov


class ResGP2023(torch.nn.Module):
    def __init__(self, fidelity_num, cigp_config=None, rho_init=1.0):
        super().__init__()
        self.fidelity_num = fidelity_num
        self.cigp_list = torch.nn.ModuleList([CIGP(cigp_config) for _ in range(fidelity_num)])
        # ResGP keeps rho fixed (Residual 'trainable': False, MFGP_ver2023May/ResGP.py:17,47)
        self.rho = [torch.nn.Parameter(torch.tensor(rho_init, dtype=torch.float32), requires_grad=False)
                    for _ in range(fidelity_num - 1)]
        self.residual_rho = torch.nn.ParameterList(self.rho)

    def compute_loss(self, x, y_list, to_fidelity_n=-1):
        if to_fidelity_n < 0:
            to_fidelity_n = self.fidelity_num + to_fidelity_n
        loss = 0.0
        for f in range(to_fidelity_n + 1):
            if f == 0:
                loss = loss + self.cigp_list[0].compute_loss(x, y_list[0])
            else:
                res = y_list[f] - y_list[f - 1] * self.residual_rho[f - 1]
                loss = loss + self.cigp_list[f].compute_loss(x, res, update_data=True)
        return loss

    def forward(self, x, x_var=0.0, to_fidelity_n=-1):
        if to_fidelity_n < 0:
            to_fidelity_n = self.fidelity_num + to_fidelity_n
        mean = var = None
        for f in range(to_fidelity_n + 1):
            if f == 0:
                mean, var = self.cigp_list[0].forward(x, x_var)
            else:
                rm, rv = self.cigp_list[f].forward(x, x_var)
                mean = mean * self.residual_rho[f - 1] + rm
                var = var * self.residual_rho[f - 1] + rv
        return mean, var
