"""Is the tests' comparator safe beside other GPU work?  The pieces of test_cigp_fp32_log_beta_keeps_the_jitter -- this library's
likelihood and assembled covariance, torch.linalg.cholesky / solve_triangular on it -- computed alone, then ITER times while another
host thread keeps the GPU busy on its own stream, every piece compared bitwise with its solo value.
usage: [NOISE_KIND=mix|eigh|gemm|nlml|torch] python tools/vendor_cholesky_probe.py [ard|sum] [iterations=300]
  torch = plain torch.matmul in the other thread: no kernel of this library runs there at all.
Result on this image (PyTorch 2.10.0+rocm7.0, docs/experiments.md): the library's pieces never differ; torch.linalg.cholesky does, in 1-34 %
of the iterations, by up to 4e-2 relative, with every kind of load -- sometimes it raises instead."""
import os
import sys
import threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from fidelityfusion_amd import _lib, kernel, eigh as E, functional as F
from fidelityfusion_amd.cigp_v10 import cigp

DEV = "cuda:0"
kind = sys.argv[1] if len(sys.argv) > 1 else "sum"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 300
noise_kind = os.environ.get("NOISE_KIND", "mix")
torch.set_default_dtype(torch.float64)
rng = np.random.default_rng(4)
n, D = 1500, 3
X = torch.tensor(rng.uniform(0, 1, (n, D)), device=DEV)
Y = torch.tensor(np.sin(3 * rng.uniform(0, 1, (n, 1))), device=DEV)
k = kernel.ARDKernel(D) if kind == "ard" else kernel.SumKernel(kernel.LinearKernel(D), kernel.MaternKernel(D))
m = cigp(k, 2.0).to(DEV).float()


def pieces():
    with torch.no_grad():
        got = m.negative_log_likelihood(X, Y)
        K = m.kernel(X, X)
        eye = torch.eye(n, device=DEV, dtype=torch.float32)
        Sigma = K + m.log_beta.exp().pow(-1) * eye + 1e-6 * eye
        L = torch.linalg.cholesky(Sigma)
        Gamma = torch.linalg.solve_triangular(L, Y, upper=False)
        want = -(0.5 * (Gamma ** 2).sum() + L.diagonal().log().sum() + 0.5 * n * np.log(2 * 3.1415))
    return {"library likelihood": got.clone(), "library K": K.clone(), "Sigma": Sigma.clone(), "torch cholesky": L.clone(),
            "torch solve_triangular": Gamma.clone(), "reference likelihood": want.clone()}


ref = pieces()
torch.cuda.synchronize()
again = pieces()
torch.cuda.synchronize()
print("alone, twice: every piece identical: %s;  rel(library, reference) = %.1e"
      % (all(torch.equal(ref[q], again[q]) for q in ref),
         float(((ref["library likelihood"] - ref["reference likelihood"]) / ref["reference likelihood"]).abs())), flush=True)
stop, rounds = threading.Event(), [0]


def load():
    torch.cuda.set_device(0)
    dev = torch.device(DEV)
    st = torch.cuda.Stream(0)
    with torch.cuda.stream(st), _lib.thread_slot(7), torch.no_grad():
        g = torch.Generator(device=dev).manual_seed(99)
        Xn = torch.rand((2048, 6), generator=g, device=dev)
        d = torch.cdist(Xn, Xn)
        Kn = torch.exp(-0.5 * d * d)
        B = torch.randn((3072, 2048), generator=g, device=dev)
        Yn = torch.randn((2048, 3), generator=g, device=dev)
        w, amp, dadd = torch.ones(6, device=dev), torch.ones(1, device=dev), torch.full((1,), 0.05, device=dev)
        while not stop.is_set():
            what = {"mix": rounds[0] % 3, "eigh": 0, "gemm": 1, "nlml": 2, "torch": 3}[noise_kind]
            if what == 0:
                E.eigh(Kn)
            elif what == 1:
                for _ in range(6):
                    F.matmul_nt(B, B)
            elif what == 2:
                for _ in range(8):
                    F.nlml(Xn, Yn, w, amp, diag_add=dadd, clamp=1e-30)
            else:
                for _ in range(6):
                    torch.matmul(B, B.T)
            st.synchronize()
            rounds[0] += 1


t = threading.Thread(target=load, daemon=True)
t.start()
bad = {q: 0 for q in ref}
worst = {q: 0.0 for q in ref}
raised = 0
for it in range(iters):
    try:
        cur = pieces()
    except Exception as e:   # noqa: BLE001  (torch.linalg.cholesky sometimes reports a non-PD matrix under load)
        raised += 1
        continue
    torch.cuda.synchronize()
    for q in ref:
        if not torch.equal(ref[q], cur[q]):
            bad[q] += 1
            worst[q] = max(worst[q], float((ref[q] - cur[q]).abs().max() / ref[q].abs().max()))
stop.set()
t.join(timeout=60)
print("beside a '%s' load, %d iterations (%d rounds of the load, %d iterations raised):" % (noise_kind, iters, rounds[0], raised))
for q in ref:
    print("   %-24s differed %4d times   worst relative difference %.2e" % (q, bad[q], worst[q]))
