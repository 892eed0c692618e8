"""2023-API `HOGP` (reference: MFGP_ver2023May/base_gp/hogp.py:20-233) on the device -- the stateful variant of
`fidelityfusion_amd.hogp_simple.HOGP_simple`: one kernel PER mode (`create_kernels` over the repeated config), the
noise box, `y_var` added to A, `train_x` / `train_y` kept from the first `compute_loss` (or when `update_data`), and a
`forward` whose "variance" is diag(K) + A x_0 (K* K_x + 1e-6 eye^2)  x_m U_m^2 exactly as written (:226-229, no
inverse).  Covariances from the library's assembly, mode products on the fp64 GEMM, eigendecompositions by the library's own
solvers (LDS Jacobi up to 64 x 64, the two-stage `ffgp_syevd` above; see hogp_simple.py)."""
import math

import torch

from ..hogp_simple import _outer, kron_nll, mode_dot, multi_mode_dot
from .. import functional as F
from .cigp import GP_noise_box, SE_kernel, _merge, _single

JITTER = 1e-6

default_config = {
    "noise": {"init_value": 1.0, "format": "linear"},
    "kernel": [{"SE": {"noise_exp_format": True, "length_scale": 1.0, "scale": 1.0}}],
    "learnable_grid": False,
    "learnable_mapping": False,
    "fidelity_shapes": None,
}


def create_kernels(kernel_configs):
    """kernel/kernel_utils.py:5-16 -- like `create_kernel`, the config dict lands in SE_kernel's `noise_exp_format`"""
    out = []
    for cfg in kernel_configs:
        for name, kc in cfg.items():
            if name != "SE":
                raise NotImplementedError
            out.append(SE_kernel(kc))
    return torch.nn.ModuleList(out)


class HOGP(torch.nn.Module):
    def __init__(self, gp_model_config=None):
        super().__init__()
        self.gp_model_config = _merge(default_config, gp_model_config)
        y_shape = self.gp_model_config["fidelity_shapes"]
        if y_shape is None:
            raise ValueError("y_shape must be set as list")
        if isinstance(y_shape[0], (list, torch.Size)):
            y_shape = y_shape[0]
        self.noise_box = GP_noise_box(self.gp_model_config["noise"])
        self.train_x = None
        self.train_y = None
        self.n_dim = len(y_shape)
        self.kernel_list = create_kernels(self.gp_model_config["kernel"] * (self.n_dim + 1))
        self.grid = torch.nn.ParameterList([torch.nn.Parameter(torch.tensor(range(v)).reshape(-1, 1).float()) for v in y_shape])
        if self.gp_model_config["learnable_grid"] is False:
            for p in self.grid:
                p.requires_grad = False
        self.mapping_vector = torch.nn.ParameterList([torch.nn.Parameter(torch.eye(v)) for v in y_shape])
        if self.gp_model_config["learnable_mapping"] is False:
            for p in self.mapping_vector:
                p.requires_grad = False

    def check_single_tensor(self, t):
        return _single(t)

    def _dev(self):
        return F._device_of(*list(self.parameters()))

    def compute_kernel_cache(self):
        dev = self._dev()
        ks = [F.kernel_on_device(self.kernel_list[0], self.train_x, self.train_x)]
        for i in range(self.n_dim):
            _in = mode_dot(self.grid[i].to(device=dev, dtype=torch.float64),
                           self.mapping_vector[i].to(device=dev, dtype=torch.float64), 0)
            ks.append(F.kernel_on_device(self.kernel_list[i + 1], _in, _in))
        self.k_result_cache = ks   # the eigen pairs (`eigen_cache`) come out of the likelihood evaluation

    def compute_loss(self, x, y, x_var=0.0, y_var=0.0, update_data=False):
        x, y = _single(x), _single(y)
        if self.train_x is None or update_data:
            self.train_x = x
            self.train_y = y
        self.compute_kernel_cache()
        dev = self._dev()
        tau = self.noise_box.get().to(dev).pow(-1)
        tau = tau + (y_var.to(device=dev, dtype=torch.float64) if isinstance(y_var, torch.Tensor) else y_var)
        loss, cache = kron_nll(self.train_y.to(device=dev, dtype=torch.float64), tau, self.k_result_cache)
        self.eigen_cache = cache["eigen"]
        self.A = cache["A"]
        self.g = cache["g"]
        odt = y.dtype if y.dtype.is_floating_point else torch.float64
        return loss.to(device=y.device, dtype=odt)

    def forward(self, x, x_vars=0.0):
        x = _single(x)
        with torch.no_grad():
            K_star = F.kernel_on_device(self.kernel_list[0], x, self.train_x)
            predict_u = multi_mode_dot(self.g, [K_star] + self.k_result_cache[1:])
            diag_K_dims = _outer([K.diag() for K in self.k_result_cache[1:]]).unsqueeze(0)
            diag_K_x = F.kernel_on_device(self.kernel_list[0], x, x).diag()
            for _ in range(self.n_dim):
                diag_K_x = diag_K_x.unsqueeze(-1)
            diag_K = diag_K_x * diag_K_dims
            S_2 = (self.A * self.A.pow(-1 / 2)).pow(2)
            Kx = self.k_result_cache[0]
            ev_x = F.matmul_nt(K_star, Kx.T.contiguous()) + JITTER * torch.eye(K_star.shape[0], Kx.shape[0], device=Kx.device,
                                                                                 dtype=torch.float64).pow(2)
            evs = [ev_x] + [self.eigen_cache[i + 1].vector.pow(2) for i in range(self.n_dim)]
            var_diag = diag_K + multi_mode_dot(S_2, evs)
            odt = x.dtype if x.dtype.is_floating_point else torch.float64
        return predict_u.to(device=x.device, dtype=odt), var_diag.to(device=x.device, dtype=odt)
