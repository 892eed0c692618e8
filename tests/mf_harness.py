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
  * `AR` / `train_ar`, `NAR` / `train_nar` -- FidelityFusion_Models/AR_autoRegression.py:11-133 (residual chain with a
                          learnable rho per fidelity, trained through dNLL/dY and dNLL/dy_var) and NAR.py:11-113
                          (the low-fidelity prediction concatenated to the inputs);
  * `GAR` / `train_gar`   -- `GAR.forward` and `train_GAR` (FidelityFusion_Models/GAR.py:14-127): the same residual chain
                          with `HOGP_simple` blocks on tensor-valued outputs;
  * `fidelity_kernel_MCMC`, `ContinuousAutoRegression`, `train_car` -- CAR (FidelityFusion_Models/
                          CAR_ContinuousAutoRegression.py:14-133): `GP_basic` blocks whose residual kernels are a base
                          kernel times a Monte-Carlo fidelity integral;
  * `ResGP2023`        -- the 2023 joint loss `loss = sum_f cigp_list[f].compute_loss(x, res_f)` with the fixed-rho
                          residual chain (MFGP_ver2023May/ResGP.py:200-246, multiscale_coupling/Residual.py:9-33) and
                          its `forward` (:145-171), aligned / subset regime (one shared x).
"""
import torch

from fidelityfusion_amd.cigp_v10 import cigp
from fidelityfusion_amd.gp_basic import GP_basic
from fidelityfusion_amd.hogp_simple import HOGP_simple
from fidelityfusion_amd.gp_computation_pack import Tensor_linear
from fidelityfusion_amd.mfgp2023 import CIGP


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


class fidelity_kernel_MCMC(torch.nn.Module):
    """CAR's fidelity kernel (CAR_ContinuousAutoRegression.py:14-67): kernel1(x1, x2) times
    |signal_variance| * MC-integral over the fidelity indicator, 100 uniform samples drawn after
    `torch.manual_seed(105)` on every call (the reference reseeds the global generator there; kept).  The scalar is
    plain torch (autograd reaches b, length_scales, signal_variance); it is folded into the base kernel's amplitude, so
    a stationary kernel1 stays on the fused path (`effective()` / `kfun()`)."""
    _ffgp_device_aware = True

    def __init__(self, input_dim, kernel1, lf, hf, b, initial_length_scale=1.0, initial_signal_variance=1.0, eps=1e-3):
        super().__init__()
        self.kernel1 = kernel1
        self.b = b
        self.lf = lf
        self.hf = hf
        self.length_scales = torch.nn.Parameter(torch.ones(input_dim) * initial_length_scale)
        self.signal_variance = torch.nn.Parameter(torch.tensor([initial_signal_variance]))
        self.eps = eps
        self.seed = 105

    def scale(self):
        length_scales = torch.abs(self.length_scales) + self.eps
        N = 100
        torch.manual_seed(self.seed)
        z1 = torch.rand(N) * (self.hf - self.lf) + self.lf
        z2 = torch.rand(N) * (self.hf - self.lf) + self.lf
        ls = length_scales.cpu()
        dist_z = (z1 / ls - z2 / ls) ** 2
        b = self.b.cpu()
        z_part = (-b * (z1 - self.hf) - b * (z2 - self.hf) - 0.5 * dist_z).exp()
        z_part_mc = z_part.mean() * (self.hf - self.lf) * (self.hf - self.lf)
        return self.signal_variance.abs().cpu() * z_part_mc

    @property
    def effective(self):
        # hasattr(kernel, "effective") is how the GP modules pick the fused path: only a stationary base kernel has it
        if not hasattr(self.kernel1, "effective"):
            raise AttributeError("effective")
        return self._effective

    def _effective(self):
        w, amp, clamp = self.kernel1.effective()
        return w, amp * self.scale().to(amp.device), clamp

    def kfun(self):
        return self.kernel1.kfun() if hasattr(self.kernel1, "kfun") else (0, 1.0)

    def forward(self, x1, x2):
        K = self.kernel1(x1, x2)
        return self.scale().to(K.device) * K


class ContinuousAutoRegression(torch.nn.Module):
    """CAR (:69-114) on the drop-in blocks; `data[i] = (x, y)` replaces the data manager (fidelity 0, then the
    residual sets `train_car` produces)."""

    def __init__(self, fidelity_num, kernel_list, b_init=1.0):
        super().__init__()
        self.fidelity_num = fidelity_num
        self.b = torch.nn.Parameter(torch.tensor(b_init))
        blocks = [GP_basic(kernel=kernel_list[0], noise_variance=1.0)]
        for f in range(fidelity_num - 1):
            input_dim = kernel_list[0].length_scales.shape[0]
            blocks.append(GP_basic(kernel=fidelity_kernel_MCMC(input_dim, kernel_list[f + 1], f, f + 1, self.b),
                                   noise_variance=1.0))
        self.cigp_list = torch.nn.ModuleList(blocks)

    def forward(self, data, x_test):
        y_high = cov_high = y_low = cov_low = None
        for i in range(self.fidelity_num):
            x_train, y_train = data[i]
            if i == 0:
                y_low, cov_low = self.cigp_list[0](x_train, y_train, x_test)
                if self.fidelity_num == 1:
                    y_high, cov_high = y_low, cov_low
            else:
                y_res, cov_res = self.cigp_list[i](x_train, y_train, x_test)
                y_high = y_low + self.b * y_res
                cov_high = cov_low + (self.b ** 2) * cov_res
                y_low, cov_low = y_high, cov_high
        return y_high, cov_high


def train_car(model, data0, overlaps, max_iter=100, lr_init=1e-1):
    """`train_CAR` (:116-148).  overlaps[i-1] = (y_low, subset_x, y_high) on the shared inputs of fidelities i-1 and i.
    Returns (LL trace, data list for `ContinuousAutoRegression.forward`)."""
    trace, data = [], [data0]
    for f in range(model.fidelity_num):
        optimizer = torch.optim.Adam(model.parameters(), lr=lr_init)
        if f == 0:
            x, y = data0
            for _ in range(max_iter):
                optimizer.zero_grad()
                ll = model.cigp_list[0].log_likelihood(x, y)
                trace.append(float(ll.detach()))
                (-ll).sum().backward()
                optimizer.step()
        else:
            y_low, x, y_high = overlaps[f - 1]
            for i in range(max_iter):
                optimizer.zero_grad()
                y_res = y_high - model.b.exp() * y_low
                if i == max_iter - 1:
                    data.append((x.detach(), y_res.detach()))
                ll = model.cigp_list[f].log_likelihood(x, y_res)
                trace.append(float(ll.detach()))
                (-ll).sum().backward()
                optimizer.step()
    return trace, data


class GAR(torch.nn.Module):
    """`FidelityFusion_Models/GAR.py:14-72` on the drop-in blocks; `xs[i]` = training inputs of block i (fidelity 0,
    then the residual sets), which is all `forward` reads from the data manager (the blocks cache K, A, g)."""

    def __init__(self, fidelity_num, kernel_list, data_shape_list):
        super().__init__()
        self.fidelity_num = fidelity_num
        blocks = []
        for i in range(fidelity_num):
            k = i + 1 if i < len(data_shape_list) - 1 else len(data_shape_list) - 1
            blocks.append(HOGP_simple(kernel=kernel_list[i], noise_variance=1.0, output_shape=data_shape_list[k]))
        self.hogp_list = torch.nn.ModuleList(blocks)
        self.Tensor_linear_list = torch.nn.ModuleList(
            [Tensor_linear(data_shape_list[i], data_shape_list[i + 1]) for i in range(fidelity_num - 1)])

    def forward(self, xs, x_test, to_fidelity=None):
        level = to_fidelity if to_fidelity is not None else self.fidelity_num - 1
        mean_high = var_high = mean_low = var_low = None
        for i in range(level + 1):
            if i == 0:
                mean_low, var_low = self.hogp_list[0].forward(xs[0], x_test)
                if level == 0:
                    mean_high, var_high = mean_low, var_low
            else:
                mean_res, var_res = self.hogp_list[i].forward(xs[i], x_test)
                mean_high = self.Tensor_linear_list[i - 1](mean_low) + mean_res
                var_high = self.Tensor_linear_list[i - 1](var_low) + var_res
                mean_low, var_low = mean_high, var_high
        return mean_high, var_high


def train_gar(model, data0, fills, max_iter=100, lr_init=1e-1):
    """`train_GAR` (GAR.py:74-127), non-subset mode; same conventions as `train_cigar`.  Returns (loss trace, xs)."""
    trace, xs = [], [data0[0]]
    for f in range(model.fidelity_num):
        optimizer = torch.optim.Adam(model.parameters(), lr=lr_init)
        if f == 0:
            x, y = data0
            for _ in range(max_iter):
                optimizer.zero_grad()
                loss = model.hogp_list[0].log_likelihood(x, y)
                trace.append(float(loss.detach()))
                loss.backward()
                optimizer.step()
        else:
            x, y_low, y_high = fills[f - 1]
            xs.append(x.detach())
            for _ in range(max_iter):
                optimizer.zero_grad()
                res_mean = y_high[0] - model.Tensor_linear_list[f - 1](y_low[0])
                res_var = (y_high[1] - y_low[1]).abs()
                loss = model.hogp_list[f].log_likelihood(x, [res_mean, res_var])
                trace.append(float(loss.detach()))
                loss.backward()
                optimizer.step()
    return trace, xs


class AR(torch.nn.Module):
    """`AR_autoRegression.py:11-80` on the drop-in blocks; `data[0] = (x, y)`, `data[i] = (x, [res_mean, res_var])`."""

    def __init__(self, fidelity_num, kernel_list, rho_init=1.0):
        super().__init__()
        self.fidelity_num = fidelity_num
        self.gpr_list = torch.nn.ModuleList([cigp(kernel=kernel_list[i], log_beta=1.0) for i in range(fidelity_num)])
        self.rho_list = torch.nn.ParameterList([torch.nn.Parameter(torch.tensor(rho_init)) for _ in range(fidelity_num - 1)])

    def forward(self, data, x_test, to_fidelity=None):
        level = to_fidelity if to_fidelity is not None else self.fidelity_num - 1
        y_high = cov_high = y_low = cov_low = None
        for i in range(level + 1):
            x_train, y_train = data[i]
            if i == 0:
                y_low, cov_low = self.gpr_list[0](x_train, y_train, x_test)
                if level == 0:
                    y_high, cov_high = y_low, cov_low
            else:
                y_res, cov_res = self.gpr_list[i](x_train, y_train, x_test)
                y_high = y_low + self.rho_list[i - 1] * y_res
                cov_high = cov_low + (self.rho_list[i - 1] ** 2) * cov_res
                y_low, cov_low = y_high, cov_high
        return y_high, cov_high


def train_ar(model, data0, fills, max_iter=100, lr_init=1e-1):
    """`train_AR` (:82-133), non-subset mode; conventions as `train_cigar`."""
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
            rho = model.rho_list[f - 1]
            for i in range(max_iter):
                optimizer.zero_grad()
                res_mean = y_high[0] - rho * y_low[0]
                res_var = (y_high[1] - rho * y_low[1]).abs()
                if i == max_iter - 1:
                    data.append((x.detach(), [res_mean.detach(), res_var.detach()]))
                ll = model.gpr_list[f].negative_log_likelihood(x, [res_mean, res_var])
                trace.append(float(ll.detach()))
                (-ll).backward()
                optimizer.step()
    return trace, data


class NAR(torch.nn.Module):
    """`NAR.py:11-58`: block i > 0 is a GP on [x, prediction of block i-1]."""

    def __init__(self, fidelity_num, kernel_list):
        super().__init__()
        self.fidelity_num = fidelity_num
        self.gpr_list = torch.nn.ModuleList([cigp(kernel=kernel_list[i], log_beta=1.0) for i in range(fidelity_num)])

    def forward(self, data, x_test, to_fidelity=None):
        level = to_fidelity if to_fidelity is not None else self.fidelity_num - 1
        y_high = cov_high = y_low = None
        for i in range(level + 1):
            x_train, y_train = data[i]
            if i == 0:
                y_low, cov_low = self.gpr_list[0](x_train, y_train, x_test)
                if level == 0:
                    y_high, cov_high = y_low, cov_low
            else:
                concat_input = torch.cat([x_test, y_low.reshape(-1, 1)], dim=-1)
                y_high, cov_high = self.gpr_list[i](x_train, y_train, concat_input)
                y_low = y_high
        return y_high, cov_high


def train_nar(model, data0, fills, max_iter=100, lr_init=1e-1):
    """`train_NAR` (:60-113), non-subset mode: fills[i-1] as in `train_cigar`; the block trains on [x, y_low_mean]."""
    trace, data = [], [data0]
    for f in range(model.fidelity_num):
        optimizer = torch.optim.Adam(model.parameters(), lr=lr_init)
        if f == 0:
            x, y = data0
        else:
            sx, y_low, y_high = fills[f - 1]
            x = torch.cat([sx, y_low[0]], dim=-1)
            y = [y_high[0], y_high[1]]
            data.append((x.detach(), [y[0].detach(), y[1].detach()]))
        for _ in range(max_iter):
            optimizer.zero_grad()
            ll = model.gpr_list[f].negative_log_likelihood(x, y)
            trace.append(float(ll.detach()))
            (-ll).backward()
            optimizer.step()
    return trace, data


def overlap_and_unique(x1, y1, x2, y2):
    """The data manager's subset bookkeeping (`get_overlap_input_data` / `get_unique_input_data`, MF_data.py:176-252,
    un-normalised): rows of (x1, y1) / (x2, y2) whose inputs occur in both sets, and the rows that do not.  The two
    masks come from the device hash join instead of the reference's N1 x N2 x D broadcast comparison."""
    from fidelityfusion_amd import functional as F
    m1, m2 = F.rows_in(x1, x2), F.rows_in(x2, x1)
    return (x1[m1], y1[m1], x2[m2], y2[m2]), (x1[~m1], y1[~m1], x2[~m2], y2[~m2])


class MeanResidualGP(torch.nn.Module):
    """A caller that assembles its own covariance and subtracts a learnable mean before calling the gp_pack functions --
    the call shape of GaussianProcess/cigp_withMean.py:44-62 (pinned by the `gp_withmean_multitask` fixture): the GP acts on
    the residual y - m(x), the prediction adds m(x*) back.  `mean_func` is any module; the fixture's is a 2-layer MLP."""

    def __init__(self, kernel, noise_variance, mean_func):
        super().__init__()
        self.kernel, self.mean_func = kernel, mean_func
        self.noise_variance = torch.nn.Parameter(torch.tensor([noise_variance]))

    def covariance(self, x):
        from fidelityfusion_amd import functional as F
        return F.add_diagonal(F.kernel_on_device(self.kernel, x, x), self.noise_variance.pow(2))

    def forward(self, x_train, y_train, x_test):
        from fidelityfusion_amd import functional as F
        from fidelityfusion_amd import gp_computation_pack as gp_pack
        cross, prior = F.kernel_on_device(self.kernel, x_train, x_test), F.kernel_on_device(self.kernel, x_test, x_test)
        mu, cov = gp_pack.conditional_Gaussian(y_train - self.mean_func(x_train), self.covariance(x_train), cross, prior)
        return mu + self.mean_func(x_test).to(mu.device), cov

    def log_likelihood(self, x_train, y_train):
        from fidelityfusion_amd import gp_computation_pack as gp_pack
        return gp_pack.Gaussian_log_likelihood(y_train - self.mean_func(x_train), self.covariance(x_train).to(y_train.device))
