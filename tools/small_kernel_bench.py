"""The one-kernel NLML of csrc/small.hip against the blocked path, one library call, n = 16 ... 128."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from fidelityfusion_amd import _lib, functional as F
torch.set_default_dtype(torch.float64)
dev = torch.device("cuda", 0)
h = _lib.handle(0); _lib.bind_stream(h, 0)
def timed(fn, reps=300):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
for n in (16, 32, 48, 64, 96, 128):
    X = torch.rand(n, 2, device=dev); Y = torch.sin(X.sum(1, keepdim=True))
    w = torch.ones(2, device=dev); amp = torch.ones(1, device=dev); dadd = torch.tensor([0.37], device=dev)
    keep = []
    p, _ = F._problem(dev, X, Y, w, amp, dadd, None, None, 0.0, 0.0, 1e-30, 1, 3.1415, keep, (0, 1.0))
    out = torch.empty((), device=dev)
    g = _lib.Grads(); gw, ga, gd = torch.empty(2, device=dev), torch.empty(1, device=dev), torch.empty(1, device=dev)
    g.g_w_dev, g.g_amp_dev, g.g_diag_add_dev = gw.data_ptr(), ga.data_ptr(), gd.data_ptr()
    res = []
    _lib.set_option("small_max_n", 128)     # (the library's own threshold is n <= 40)
    for fused in (1, 0):
        _lib.set_option("small_fused", fused)
        res.append((timed(lambda: _lib.lib.ffgp_nlml_fused(h, C.byref(p), out.data_ptr(), C.byref(g))),
                    timed(lambda: _lib.lib.ffgp_nlml_fused(h, C.byref(p), out.data_ptr(), None)), float(out), gw.clone()))
    print("n=%3d  one kernel: fwd+grad %.3f ms, fwd %.3f | blocked path: fwd+grad %.3f, fwd %.3f | same value %s grad diff %.1e"
          % (n, res[0][0], res[0][1], res[1][0], res[1][1], abs(res[0][2] - res[1][2]) < 1e-9 * abs(res[1][2]), float((res[0][3] - res[1][3]).abs().max())), flush=True)
_lib.set_option("small_fused", 1)
_lib.set_option("small_max_n", 0)
