"""2023-API single-fidelity CIGP on the HIP path.

Reference: MFGP_ver2023May/base_gp/cigp.py:19-136 (stateful train_x/train_y, `compute_loss` returns +nll with
pi = 3.1415, `forward` runs under no_grad and returns the diagonal variance expanded to [Nt, d]),
kernel/SE_kernel.py:4-44, utils/gp_noise.py:9-24, kernel/kernel_utils.py:20-31 (whose `create_kernel` hands the
whole config dict to SE_kernel as `noise_exp_format`, so `noise_exp_format is True` is False and the kernel
always runs in linear format with length_scale = scale = 1 -- reproduced here).
"""
import copy

import torch

from .. import functional as F

JITTER = 1e-6
EPS = 1e-10
PI = 3.1415

default_config = {
    "noise": {"init_value": 1.0, "format": "exp"},
    "kernel": {"SE": {"noise_exp_format": True, "length_scale": 1.0, "scale": 1.0}},
}


def _merge(default, override):
    out = copy.deepcopy(default)
    if override:
        for k, v in override.items():
            out[k] = _merge(out[k], v) if isinstance(v, dict) and isinstance(out.get(k), dict) else v
    return out


class GP_noise_box(torch.nn.Module):
    """utils/gp_noise.py:9-24 -- a float32 parameter, 'exp' or 'linear' format."""

    def __init__(self, noise_config):
        super().__init__()
        assert noise_config["format"] in ["exp", "linear"], "noise format should be 'exp' or 'linear'"
        self.config = noise_config
        self.format = noise_config["format"]
        v = torch.tensor(noise_config["init_value"], dtype=torch.float32)
        self.value = torch.nn.Parameter(torch.log(v) if self.format == "exp" else v)

    def get(self):
        return torch.exp(self.value) if self.format == "exp" else self.value


class SE_kernel(torch.nn.Module):
    """kernel/SE_kernel.py:4-44: K = scale * exp(-1/2 * ||x/l - x'/l||^2), scalar l; inputs with more than two
    dimensions are flattened."""

    def __init__(self, noise_exp_format, length_scale=1.0, scale=1.0):
        super().__init__()
        self.noise_exp_format = noise_exp_format
        length_scale = torch.tensor(length_scale)
        scale = torch.tensor(scale)
        if noise_exp_format is True:
            self.length_scale = torch.nn.Parameter(torch.log(length_scale))
            self.scale = torch.nn.Parameter(torch.log(scale))
        else:
            self.length_scale = torch.nn.Parameter(length_scale)
            self.scale = torch.nn.Parameter(scale)

    def effective(self):
        if self.noise_exp_format is True:
            return torch.exp(-self.length_scale).reshape(1), torch.exp(self.scale).reshape(1), F.NEG_INF
        return (1.0 / self.length_scale).reshape(1), self.scale.reshape(1), F.NEG_INF

    def forward(self, X, X2):
        w, amp, clamp = self.effective()
        return F.kernel_matrix(X, X2, w, amp, clamp)


def create_kernel(kernel_config):
    if isinstance(kernel_config, list) and len(kernel_config) == 1:
        kernel_config = kernel_config[0]
    for name, cfg in kernel_config.items():
        if name == "SE":
            return SE_kernel(cfg)  # the dict lands in `noise_exp_format`, as in the reference (kernel_utils.py:24)
        raise NotImplementedError
    raise NotImplementedError


def _single(t):
    if isinstance(t, list):
        assert len(t) == 1, "CIGP model only support one input"
        t = t[0]
    return t


def _flat(x):
    return x.reshape(x.shape[0], -1) if x.dim() > 2 else x


class CIGP(F.PosteriorCacheMixin, torch.nn.Module):
    def __init__(self, gp_model_config=None):
        super().__init__()
        self.gp_model_config = _merge(default_config, gp_model_config)
        self.noise_box = GP_noise_box(self.gp_model_config["noise"])
        self.kernel = create_kernel(self.gp_model_config["kernel"])
        self.train_x = None
        self.train_y = None
        self._pcache = F.PosteriorCache()   # the factor of (train_x, train_y, parameters) between predictions

    def forward(self, x, x_var=0.0):
        x = _single(x)
        if self.train_x is None:
            print("gp model model hasn't been trained. predict failed")
            return None
        with torch.no_grad():
            inv_noise = self.noise_box.get().pow(-1).double()  # fp32 parameter arithmetic, then promoted (cigp.py:81)
            xq = _flat(x)

            def build():   # first query rides in the factorisation; the factor stays for the next prediction
                w, amp, clamp = self.kernel.effective()
                return F.Posterior(_flat(self.train_x), self.train_y, w, amp, inv_noise + JITTER, clamp=clamp, first_query=xq)
            post, fresh = self._pcache.get([self.train_x, self.train_y] + list(self.parameters()), build)
            if fresh:
                u, vd = post.first[0], post.first[1].diagonal() + float(inv_noise)
            else:
                u, vd = post.predict(xq, full_cov=False, var_add_all=float(inv_noise))
            u = u.to(device=self.train_y.device, dtype=self.train_y.dtype if self.train_y.dtype.is_floating_point else torch.float64)
            var_diag = vd.to(device=u.device, dtype=u.dtype).reshape(-1, 1).expand_as(u) + x_var
        return u, var_diag

    def compute_loss(self, x, y, x_var=0.0, y_var=0.0, update_data=False):
        x = _single(x)
        y = _single(y)
        assert y.ndim == 2, "y should be 2d tensor"
        if self.train_x is None or update_data:
            self.train_x = x
            self.train_y = y
        w, amp, clamp = self.kernel.effective()
        diag_add = self.noise_box.get().pow(-1).double() + JITTER
        return F.nlml(_flat(x), y, w, amp, diag_add=diag_add, add_all=float(y_var), clamp=clamp, variant=F.FFGP_LL_V1,
                      pi_const=PI, **F._slot_args())
