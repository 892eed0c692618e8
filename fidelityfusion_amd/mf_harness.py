"""Thin harness reproducing the CALL PATTERN of the reference's multi-fidelity trainers on the drop-in GP blocks
(SURVEY section 8a row X1).  The model logic of the reference (data managers, non-subset fill, Tensor_linear,
Matrix_Mapping) is out of scope; what is mirrored here is exactly how those models drive the hot path, so that
parity and benchmarks exercise it the way the reference does:

  * `train_gp_blocks`  -- the per-fidelity Adam loops of `train_ResGP` / `train_AR` / `train_CIGAR`
                          (FidelityFusion_Models/ResGP.py:67-112): `loss = -gpr.negative_log_likelihood(x, y | [y, y_var])`,
                          `loss.backward()`, `optimizer.step()`, a fresh Adam over ALL parameters per fidelity;
  * `resgp_predict`    -- `ResGP.forward` (ResGP.py:31-65): sum of per-fidelity posterior means and covariances;
  * `CIGAR` / `train_cigar` -- `CIGAR.forward` and `train_CIGAR` (FidelityFusion_Models/CIGAR.py:40-134): residual
                          blocks behind the learnable `Tensor_linear` fidelity map, y given as [mean, variance];
  * `ResGP2023`        -- the 2023 joint loss `loss = sum_f cigp_list[f].compute_loss(x, res_f)` with the fixed-rho
                          residual chain (MFGP_ver2023May/ResGP.py:200-246, multiscale_coupling/Residual.py:9-33) and
                          its `forward` (:145-171), aligned / subset regime (one shared x).
"""
import torch

from .cigp_v10 import cigp
from .gp_computation_pack import Tensor_linear
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


class CIGAR(torch.nn.Module):
    """`FidelityFusion_Models/CIGAR.py:14-82` on the drop-in blocks.  The data manager is replaced by explicit data:
    `data[0] = (x, y)` (normalised fidelity-0 set), `data[i] = (x, [res_mean, res_var])` (the 'res-i' sets that
    `train_cigar` produces)."""

    def __init__(self, fidelity_num, kernel_list, data_shape_list):
        super().__init__()
        self.fidelity_num = fidelity_num
        self.gpr_list = torch.nn.ModuleList([cigp(kernel=kernel_list[i], log_beta=1.0) for i in range(fidelity_num)])
        self.Tensor_linear_list = torch.nn.ModuleList(
            [Tensor_linear(data_shape_list[i], data_shape_list[i + 1]) for i in range(fidelity_num - 1)])

    def forward(self, data, x_test, to_fidelity=None):
        level = to_fidelity if to_fidelity is not None else self.fidelity_num - 1
        mean_high = var_high = mean_low = var_low = None
        for i in range(level + 1):
            x_train, y_train = data[i]
            if i == 0:
                mean_low, var_low = self.gpr_list[0].forward(x_train, y_train, x_test)
                if mean_low.dim() == 0:
                    mean_low = mean_low.reshape(1).unsqueeze(0)
                if mean_low.dim() == 1:
                    mean_low = mean_low.unsqueeze(1)
                var_low = var_low.diag().unsqueeze(1).expand_as(mean_low)
                if level == 0:
                    mean_high, var_high = mean_low, var_low
            else:
                mean_res, _ = self.gpr_list[i].forward(x_train, y_train, x_test)
                if mean_res.dim() == 1:
                    mean_res = mean_res.unsqueeze(1)
                var_res = var_low.diag().unsqueeze(1).expand_as(mean_res)   # sic: the low-fidelity variance (:74)
                mean_high = self.Tensor_linear_list[i - 1](mean_low) + mean_res
                var_high = self.Tensor_linear_list[i - 1](var_low) + var_res
                mean_low, var_low = mean_high, var_high
        return mean_high, var_high


def train_cigar(model, data0, fills, max_iter=100, lr_init=1e-1):
    """`train_CIGAR` (CIGAR.py:84-134), non-subset mode.  data0 = (x, y) of fidelity 0; fills[i-1] =
    (x, [y_low_mean, y_low_var], [y_high_mean, y_high_var]) -- what the data manager's fill step hands the loop.
    Returns (LL trace, data list for `CIGAR.forward`)."""
    trace, data = [], [data0]
    for f in range(model.fidelity_num):
        optimizer = torch.optim.Adam(model.parameters(), lr=lr_init)
        if f == 0:
            x, y = data0
            for _ in range(max_iter):
                optimizer.zero_grad()
                ll = model.gpr_list[0].negative_log_likelihood(x, y)
                trace.append(float(ll.detach()))
                (-ll).backward()
                optimizer.step()
        else:
            x, y_low, y_high = fills[f - 1]
            for i in range(max_iter):
                optimizer.zero_grad()
                res_mean = y_high[0] - model.Tensor_linear_list[f - 1](y_low[0])
                res_var = (y_high[1] - y_low[1]).abs()
                if i == max_iter - 1:
                    data.append((x.detach(), [res_mean.detach(), res_var.detach()]))
                ll = model.gpr_list[f].negative_log_likelihood(x, [res_mean, res_var])
                trace.append(float(ll.detach()))
                (-ll).backward()
                optimizer.step()
    return trace, data
