"""Phase timeline of the pipelined diagonal-block kernel (ffgp_potrf_diag128_v2): builds fidelityfusion_amd/libffgp_dtrace.so
with -DFFGP_DIAG_TRACE (wave 0 and helper 0 stamp wall_clock64 -- 100 MHz -- at their phase boundaries) and prints the
stamps of one 128 x 128 factorisation in microseconds."""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "fidelityfusion_amd", "csrc")
SO = os.path.join(ROOT, "fidelityfusion_amd", "libffgp_dtrace.so")


def build():
    import re
    names = re.search(r"^SRCS = (.*)$", open(os.path.join(CSRC, "Makefile")).read(), re.M).group(1).split()   # the library's own list
    srcs = [os.path.join(CSRC, f) for f in names]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-DFFGP_DIAG_TRACE", "-shared",
                           "-Wno-unused-value", "-Wno-unused-result", "-o", SO] + srcs + ["-ldl"])


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "build":
        build()
        sys.exit(0)
    import torch
    lib = C.CDLL(SO)
    h = C.c_void_p()
    assert lib.ffgp_create(0, C.byref(h)) == 0
    dev = torch.device("cuda:0")
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 128           # n > 128: the launch at row `at` of a whole factorisation (in situ)
    at = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    g = torch.Generator(device=dev).manual_seed(0)
    X = torch.rand((n, 8), generator=g, device=dev, dtype=torch.float64)
    A = torch.exp(-0.5 * torch.cdist(X, X) ** 2) + 0.37 * torch.eye(n, device=dev, dtype=torch.float64)
    buf = torch.zeros(128, dtype=torch.int64, device=dev)
    mode = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    lib.ffgp_set_option(h, b"diag_v2", C.c_double(mode))
    for rep in range(3):
        W = A.clone()
        buf.zero_()
        torch.cuda.synchronize()
        assert lib.ffgp_debug_set_diag_trace(C.c_void_p(buf.data_ptr())) == 0
        assert lib.ffgp_debug_set_diag_trace_row(at if n > 128 else -1) == 0
        assert lib.ffgp_potrf(h, C.c_void_p(W.data_ptr()), n, n) == 0
        torch.cuda.synchronize()
        t = buf.cpu().numpy().astype("float64") / 100.0     # us
    t0 = t[0]
    raw = buf.cpu().numpy()
    if raw[121] > raw[120] and t[23] > t[0]:
        print("shader clock during the kernel: %.0f MHz" % ((raw[121] - raw[120]) / (t[23] - t[0])))
    print("wave 0 : start 0.00 | load+roles %.2f" % (t[1] - t0))
    for jj in range(8):
        f, w_, g_ = t[2 + 3 * jj] - t0, t[3 + 3 * jj] - t0, t[4 + 3 * jj] - t0
        print("  jj=%d  F done %.2f   doneU seen %.2f   G done %.2f" % (jj, f, w_ if jj < 7 else float("nan"), g_ if jj < 7 else float("nan")))
    if mode == 4:
        print("helper0: prologue (blocks loaded) %.2f" % (t[31] - t0))
    if mode == 4:
        print("wave 0, round-4 stamps: pivots done | operands seen | seqF set | G starts | rows announced | G done")
        for jj in range(8):
            print("  jj=%d  %6.2f %6.2f %6.2f %6.2f %6.2f %6.2f" % (jj, t[64 + jj] - t0, t[3 + 3 * jj] - t0, t[2 + 3 * jj] - t0, t[72 + jj] - t0 if jj < 7 else float("nan"),
                                                                  t[80 + jj] - t0 if jj < 7 else float("nan"), t[4 + 3 * jj] - t0 if jj < 7 else float("nan")))
    print("helper0:")
    if mode == 4:        # round-4 kernel: stage entered, (urgent work ...) inverse row started, stage complete
        for jj in range(8):
            a, c_, d = t[32 + 4 * jj] - t0, t[34 + 4 * jj] - t0, t[35 + 4 * jj] - t0
            print("  s=%d  inv(L_s) seen %.2f   solves + updates done %.2f   inverse row done %.2f" % (jj, a, c_ if jj >= 1 else float("nan"), d))
        sys.exit(0)
    for jj in range(8):
        a, b, c_, d = (t[32 + 4 * jj + k] - t0 for k in range(4))
        print("  jj=%d  seqF seen %.2f   A1/A2 done %.2f   B1/B2 done %.2f   B3 done %.2f" % (jj, a, b, c_ if jj < 7 else float("nan"), d if jj < 7 else float("nan")))
