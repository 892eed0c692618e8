"""2023-API look-alikes (reference: MFGP_ver2023May/base_gp/cigp.py, kernel/SE_kernel.py, utils/gp_noise.py)."""
from .cigp import CIGP, GP_noise_box, SE_kernel, create_kernel  # noqa: F401
