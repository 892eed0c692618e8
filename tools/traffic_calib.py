"""Calibration target for the roofline kernel's FETCH_SIZE reading: the lower-trapezoid 128-tile SYRK update
C -= P P^T at m = 16384 with k = 16 (traffic = the C tiles, read once and written once: known bytes) and with k = 512 (the
factorisation's own shape).  Run under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum:
    rocprofv3 --pmc FETCH_SIZE -d out -o p --output-format csv -- python3 tools/traffic_calib.py
The two launches per k appear as kernel ffgp_gemm_f64<0, 0, 1, 0, 128, 128> in dispatch order: k = 16, 16, 512, 512."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from fidelityfusion_amd import _lib

m = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
h = _lib.handle(0)
_lib.bind_stream(h, 0)
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
A = torch.rand((m, 512), generator=g, device=dev, dtype=torch.float64) - 0.5
Cm = torch.zeros((m, m), device=dev, dtype=torch.float64)
torch.cuda.synchronize()
p = lambda t: C.c_void_p(t.data_ptr())
for k in (16, 16, 512, 512):
    rc = _lib.lib.ffgp_gemm(h, 0, 0, 1, 0, p(A), 512, p(A), 512, p(Cm), m, m, m, k, -1.0, 1.0)
    assert rc == 0, rc
    torch.cuda.synchronize()
tiles = (m // 128) * (m // 128 + 1) // 2
print("m = %d: %d lower tiles; C bytes read = written = %.1f MB; operand bytes per launch at k: 16 -> %.1f MB, 512 -> %.1f MB (every tile's own A and B panels, before any cache)"
      % (m, tiles, tiles * 128 * 128 * 8 / 1e6, tiles * 2 * 128 * 16 * 8 / 1e6, tiles * 2 * 128 * 512 * 8 / 1e6))
