"""Which piece of a HOGP block changes its result when two host threads drive the GPU at once (development probe for
functional.threaded_blocks)?  Every piece is run on the same inputs one after another and from `nslots` threads, and compared bitwise.
usage: python tools/thread_probe.py [n=8192] [nslots=2] [blocks=4]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _devlib  # noqa: F401,E402  (development build: these switches are not in the shipped library)
import torch
from fidelityfusion_amd import eigh as E
from fidelityfusion_amd import functional as F
from fidelityfusion_amd import kernel as K_

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
nslots = int(sys.argv[2]) if len(sys.argv) > 2 else 2
nb = int(sys.argv[3]) if len(sys.argv) > 3 else 4
dev = torch.device("cuda:0")
Xs, Ks, Ys = [], [], []
kern = K_.ARDKernel(8).double().to(dev)
with torch.no_grad():
    for f in range(nb):
        g = torch.Generator(device=dev).manual_seed(f)
        X = torch.rand((n, 8), generator=g, device=dev, dtype=torch.float64)
        Xs.append(X)
        Ks.append(kern(X, X).clone())
        Ys.append(torch.randn((n, 512), generator=g, device=dev, dtype=torch.float64))
torch.cuda.synchronize()


def compare(name, fn):
    with torch.no_grad():
        ref = [fn(f) for f in range(nb)]
        torch.cuda.synchronize()
        got = F.threaded_blocks([(lambda f=f: fn(f)) for f in range(nb)], nslots=nslots)
        torch.cuda.synchronize()
    bad = []
    for f in range(nb):
        r, g_ = ref[f], got[f]
        r = r if isinstance(r, (tuple, list)) else (r,)
        g_ = g_ if isinstance(g_, (tuple, list)) else (g_,)
        for j, (a, b) in enumerate(zip(r, g_)):
            if not torch.equal(a, b):
                bad.append("block %d output %d: max |diff| %.3e (scale %.3e), %d of %d entries differ" %
                           (f, j, float((a - b).abs().max()), float(a.abs().max()), int((a != b).sum()), a.numel()))
    print("%-28s %s" % (name, "identical" if not bad else "DIFFERENT\n    " + "\n    ".join(bad)), flush=True)


compare("kernel assembly", lambda f: kern(Xs[f], Xs[f]))
compare("matmul_nt (K Y)", lambda f: F.matmul_nt(Ys[f].T.contiguous(), Ks[f]))
compare("sy2sb", lambda f: E.sy2sb(Ks[f]))
AB = [E.sy2sb(Ks[f])[0] for f in range(nb)]
compare("sb2st", lambda f: E.sb2st(AB[f]))
de = [E.sb2st(AB[f])[:2] for f in range(nb)]
compare("stedc", lambda f: E.stedc(de[f][0], de[f][1]))
compare("eigh", lambda f: E.eigh(Ks[f]))

# --- is it the handle or the concurrency? --------------------------------------------------------------------------------------
import threading
from fidelityfusion_amd import _lib
with torch.no_grad():
    ref = [E.sy2sb(Ks[f]) for f in range(nb)]
    torch.cuda.synchronize()
    for slot in (1, 2):
        with _lib.thread_slot(slot):
            alt = [E.sy2sb(Ks[f]) for f in range(nb)]
        torch.cuda.synchronize()
        print("sy2sb one after another on handle slot %d: %s" % (slot, ["same" if all(torch.equal(a, b) for a, b in zip(ref[f], alt[f])) else "DIFFERENT" for f in range(nb)]), flush=True)
    lock = threading.Lock()

    def locked(f):
        with lock:
            out = E.sy2sb(Ks[f])
            torch.cuda.synchronize()
            return out
    got = F.threaded_blocks([(lambda f=f: locked(f)) for f in range(nb)], nslots=nslots)
    torch.cuda.synchronize()
    print("sy2sb from threads, one call at a time (lock + synchronize): %s" % ["same" if all(torch.equal(a, b) for a, b in zip(ref[f], got[f])) else "DIFFERENT" for f in range(nb)], flush=True)
    for rep in range(2):
        got = F.threaded_blocks([(lambda f=f: E.sy2sb(Ks[f])) for f in range(nb)], nslots=nslots)
        torch.cuda.synchronize()
        print("sy2sb from threads, concurrent (repeat %d): %s" % (rep, ["same" if all(torch.equal(a, b) for a, b in zip(ref[f], got[f])) else "DIFFERENT" for f in range(nb)]), flush=True)

# --- which kernel?  toggle the band reduction's alternative kernels on every handle in use ------------------------------------------
def set_all(key, val):
    for s_ in range(0, nslots + 1):
        _lib.set_option_handle(_lib.handle(0, s_), key, val)


def concurrent_ok(label, reps=3):
    with torch.no_grad():
        ref_ = [E.sy2sb(Ks[f]) for f in range(nb)]
        torch.cuda.synchronize()
        out = []
        for _ in range(reps):
            got_ = F.threaded_blocks([(lambda f=f: E.sy2sb(Ks[f])) for f in range(nb)], nslots=nslots)
            torch.cuda.synchronize()
            out.append("".join("." if all(torch.equal(a, b) for a, b in zip(ref_[f], got_[f])) else "X" for f in range(nb)))
    print("%-40s %s   (. = block identical to the sequential run, X = different)" % (label, " ".join(out)), flush=True)


concurrent_ok("defaults")
set_all("sb_av_gemm", 1)
concurrent_ok("A Y on the general GEMM (sb_av_gemm=1)")
set_all("sb_av_gemm", 0)
set_all("sb_qr4", 1)
concurrent_ok("256-thread leaf QR (sb_qr4=1)")
set_all("sb_av_gemm", 1)
concurrent_ok("both")
set_all("sb_av_gemm", 0)
set_all("sb_qr4", 0)

# --- where does a concurrent band reduction first leave the sequential one? -----------------------------------------------------------
with torch.no_grad():
    ref_ = [E.sy2sb(Ks[f]) for f in range(nb)]
    torch.cuda.synchronize()
    shown = 0
    for rep in range(6):
        got_ = F.threaded_blocks([(lambda f=f: E.sy2sb(Ks[f])) for f in range(nb)], nslots=nslots)
        torch.cuda.synchronize()
        for f in range(nb):
            Yr, Yg = ref_[f][1], got_[f][1]
            if torch.equal(Yr, Yg):
                continue
            colbad = (Yr != Yg).any(0)                       # columns of Y that differ
            c0 = int(torch.nonzero(colbad)[0])
            p0 = c0 // 32
            blk = (Yr[:, 32 * p0:32 * p0 + 32] != Yg[:, 32 * p0:32 * p0 + 32])
            rows = torch.nonzero(blk.any(1)).flatten()
            r0 = 32 * p0 + 32                                # first row of the panel below the band
            rel = (rows - r0).cpu().numpy()
            d = (Yr[:, 32 * p0:32 * p0 + 32] - Yg[:, 32 * p0:32 * p0 + 32]).abs()
            ABr, ABg = ref_[f][0], got_[f][0]
            abbad = torch.nonzero((ABr != ABg).any(1)).flatten()
            import numpy as np
            print("repeat %d block %d: first differing panel %d (m = %d rows, %d leaves); rows differing in it: %d, from %d to %d (relative to the panel); "
                  "per 512-row leaf: %s; max |dY| there %.2e; columns of the panel differing: %d; first differing band column %s (panel %s)"
                  % (rep, f, p0, n - r0, (n - r0 + 511) // 512, rows.numel(), rel.min(), rel.max(),
                     np.bincount(rel // 512, minlength=(n - r0 + 511) // 512).tolist(), float(d.max()), int(blk.any(0).sum()),
                     int(abbad[0]) if abbad.numel() else None, int(abbad[0]) // 32 if abbad.numel() else None), flush=True)
            shown += 1
        if shown >= 6:
            break
