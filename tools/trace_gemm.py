"""Per-workgroup phase timeline of the fp64 GEMM kernel (development tool, not part of the product).

Builds fidelityfusion_amd/libffgp_trace.so with -DFFGP_GEMM_TRACE (every workgroup stamps s_memtime at: tile start,
first operand tile in LDS, end of the k loop, end of the epilogue), runs one launch and prints where a tile's
wall time goes.      python tools/trace_gemm.py build      (here, cross-compile)
                     python tools/trace_gemm.py run [syrk|gemm] [m] [k] [slots]   (on the GPU box)
"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "fidelityfusion_amd", "csrc")
SO = os.path.join(ROOT, "fidelityfusion_amd", "libffgp_trace.so")


def build():
    srcs = ["gemm.hip", "potrf.hip", "assemble.hip", "solve.hip", "grad.hip", "join.hip", "eig.hip", "api.hip"]
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-DFFGP_GEMM_TRACE", "-shared",
           "-Wno-unused-value", "-Wno-unused-result", "-I" + os.path.join(ROOT, "include"), "-o", SO] + [os.path.join(CSRC, f) for f in srcs]
    subprocess.check_call(cmd)
    print("built", SO)


def run(kind="syrk", m=16384, k=512, slots=0):
    import numpy as np
    import torch
    lib = C.CDLL(SO)
    h = C.c_void_p()
    assert lib.ffgp_create(0, C.byref(h)) == 0
    lib.ffgp_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_double]
    lib.ffgp_gemm.argtypes = [C.c_void_p] + [C.c_int] * 4 + [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                                             C.c_int, C.c_int, C.c_int, C.c_double, C.c_double]
    dev = "cuda:0"
    A = torch.rand((m, k), device=dev, dtype=torch.float64) - 0.5
    Cm = torch.zeros((m, m), device=dev, dtype=torch.float64)
    tiles = (m // 128) * (m // 128 + 1) // 2 if kind == "syrk" else (m // 128) ** 2
    buf = torch.zeros((tiles, 8), device=dev, dtype=torch.int64)
    lower = 1 if kind == "syrk" else 0

    def launch():
        return lib.ffgp_gemm(h, 0, 0, lower, 0, A.data_ptr(), k, A.data_ptr(), k, Cm.data_ptr(), m, m, m, k, -1.0, 1.0)
    for _ in range(3):
        launch()
    torch.cuda.synchronize()
    assert lib.ffgp_debug_set_trace(C.c_void_p(buf.data_ptr())) == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    launch()
    torch.cuda.synchronize()
    lib.ffgp_debug_set_trace(None)
    t = buf.cpu().numpy()
    clk = t[:, :4].astype(np.float64)
    wall = t[:, 7].astype(np.float64) / 100.0    # 100 MHz constant clock -> us
    hw = t[:, 6]
    cu = ((hw >> 32) & 0xf) * 1000 + ((hw >> 13) & 0x7) * 100 + ((hw >> 8) & 0xf)   # xcc, se, cu
    # s_memtime ticks: calibrate against the wall clock over the whole launch
    span_wall = wall.max() - wall.min()
    # s_memtime ticks at the shader clock; its origin differs per XCD, so only differences taken on one CU are
    # meaningful.  Calibrate on XCD 0 against the 100 MHz wall clock.
    x0 = ((hw >> 32) & 0xf) == ((hw >> 32) & 0xf)[0]
    tick_us = (wall[x0].max() - wall[x0].min()) / max(1.0, clk[x0, 0].max() - clk[x0, 0].min())
    d = np.diff(clk, axis=1) * tick_us
    print("kind=%s m=%d k=%d slots=%d tiles=%d ; s_memtime tick = %.4f ns ; launch span %.1f us" %
          (kind, m, k, slots, tiles, tick_us * 1e3, span_wall))
    names = ["prologue (decode + first operand tile -> LDS)", "k loop", "epilogue"]
    for i, nme in enumerate(names):
        print("  %-48s mean %7.2f us  p10 %7.2f  p50 %7.2f  p90 %7.2f" % (nme, d[:, i].mean(), *np.percentile(d[:, i], [10, 50, 90])))
    tot = (clk[:, 3] - clk[:, 0]) * tick_us
    print("  %-48s mean %7.2f us" % ("tile total", tot.mean()))
    # per CU: idle gaps between consecutive tiles on the same CU slot cannot be seen directly; print the busy
    # fraction of each CU over the launch instead
    xcc = (hw >> 32) & 0xf
    org = np.zeros(len(clk))
    for x in np.unique(xcc):
        org[xcc == x] = clk[xcc == x, 0].min()
    start = (clk[:, 0] - org) * tick_us
    end = (clk[:, 3] - org) * tick_us
    kl0 = (clk[:, 1] - org) * tick_us
    kl1 = (clk[:, 2] - org) * tick_us
    ucu = np.unique(cu)
    print("  distinct CUs seen: %d" % len(ucu))
    fr = []
    for c in ucu[:: max(1, len(ucu) // 16)]:
        sel = cu == c
        # time with at least one workgroup inside its k loop on this CU
        ev = sorted([(a, 1) for a in kl0[sel]] + [(b, -1) for b in kl1[sel]])
        busy, depth, last, both = 0.0, 0, 0.0, 0.0
        for tt, dlt in ev:
            if depth > 0:
                busy += tt - last
            if depth > 1:
                both += tt - last
            depth += dlt
            last = tt
        fr.append((c, sel.sum(), busy, both, end[sel].max() - start[sel].min()))
    for c, nt, busy, both, span in fr:
        print("  cu %5d: %3d tiles, span %8.1f us, >=1 WG in k loop %8.1f us (%.1f%%), 2 WGs in k loop %8.1f us (%.1f%%)" %
              (c, nt, span, busy, 100 * busy / span, both, 100 * both / span))
    # first tiles of one CU as a timeline
    c = ucu[len(ucu) // 2]
    sel = np.where(cu == c)[0]
    sel = sel[np.argsort(start[sel])][:10]
    print("  timeline of cu %d (us): start, k-loop start, k-loop end, end" % c)
    for i in sel:
        print("    tile %5d: %8.1f %8.1f %8.1f %8.1f" % (i, start[i], kl0[i], kl1[i], end[i]))


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build()
    else:
        a = sys.argv[2:]
        run(a[0] if a else "syrk", int(a[1]) if len(a) > 1 else 16384, int(a[2]) if len(a) > 2 else 512,
            int(a[3]) if len(a) > 3 else 0)
