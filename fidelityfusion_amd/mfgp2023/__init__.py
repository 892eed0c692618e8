"""2023-API look-alikes (reference: MFGP_ver2023May/base_gp/{cigp,hogp}.py, kernel/SE_kernel.py, utils/gp_noise.py)."""
from .cigp import CIGP, GP_noise_box, SE_kernel, create_kernel  # noqa: F401
from .hogp import HOGP, create_kernels  # noqa: F401
