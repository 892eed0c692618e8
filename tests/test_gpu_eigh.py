"""The round-2 block Jacobi (fidelityfusion_amd/eigh.py, the slow independent cross-check) against LAPACK, and the HOGP block
 with every eigensolver: the default two-stage ffgp_syevd, the block Jacobi, and rocSOLVER as the comparator."""
import numpy as np
import pytest
import torch

pytestmark = [pytest.mark.gpu, pytest.mark.noisy]   # (noisy: beside a background load by default, tests/conftest.py)
DEV = "cuda:0"


def _kernel_matrix(n, D, ls, seed=0, kind="se"):
    g = torch.Generator(device=DEV).manual_seed(seed)
    X = torch.rand((n, D), generator=g, device=DEV, dtype=torch.float64)
    d = torch.cdist(X / ls, X / ls)
    return torch.exp(-0.5 * d * d) if kind == "se" else torch.exp(-d)


def _check(K, ev, U, tol_rec, tol_orth=2e-10):
    n = K.shape[0]
    lam_ref = torch.linalg.eigvalsh(K)
    scale = float(lam_ref.abs().max())
    assert ev.shape == (n,) and U.shape == (n, n)
    assert bool((ev[1:] >= ev[:-1]).all())                                          # ascending, as torch.linalg.eigh
    assert float((U.T @ U - torch.eye(n, device=DEV, dtype=torch.float64)).abs().max()) < tol_orth
    rec = (U * ev) @ U.T
    assert float(torch.linalg.matrix_norm(rec - K)) <= tol_rec * float(torch.linalg.matrix_norm(K))
    assert float((ev - lam_ref).abs().max()) <= tol_rec * scale


@pytest.mark.parametrize("n", [70, 128, 257, 600])
def test_block_jacobi_dense_core_vs_lapack(n):
    from fidelityfusion_amd.eigh import jacobi_eigh
    g = torch.Generator(device=DEV).manual_seed(n)
    M = torch.randn((n, n), generator=g, device=DEV, dtype=torch.float64)
    for A in (0.5 * (M + M.T), _kernel_matrix(n, 2, 0.7, seed=n)):
        ev, U = jacobi_eigh(A)
        _check(A, ev, U, 5e-12)


def test_hogp_block_same_with_both_eigensolvers():
    """HOGP_simple.log_likelihood / forward at N = 500, d = 6 x 5 with the library's two-stage solver (the default), its block
    Jacobi and rocSOLVER (comparator): loss, every gradient, cached g, posterior -- the quantities are basis-independent, so they
    must agree although the solvers return different bases of the near-null space"""
    from fidelityfusion_amd import hogp_simple, kernel
    n, d1, d2 = 500, 6, 5
    g = torch.Generator(device=DEV).manual_seed(5)
    X = torch.rand((n, 3), generator=g, device=DEV, dtype=torch.float64)
    Y = torch.randn((n, d1, d2), generator=g, device=DEV, dtype=torch.float64)
    Xt = torch.rand((9, 3), generator=g, device=DEV, dtype=torch.float64)
    res = {}
    assert hogp_simple.EIGENSOLVER == "ffgp"          # the library's own solver is the default
    assert "rocsolver" not in open(hogp_simple.__file__).read().split('"""', 2)[2]      # no vendor route in the product module

    class _VendorPairs:                               # the comparator lives here, in the test: torch.linalg.eigh = rocSOLVER
        def __init__(self, matrix):
            self.value, self.vector = torch.linalg.eigh(matrix.detach(), UPLO="U")

    for solver in ("ffgp", "jacobi", "rocsolver"):
        own_pairs = hogp_simple.eigen_pairs
        if solver == "rocsolver":
            hogp_simple.eigen_pairs = _VendorPairs
        else:
            hogp_simple.EIGENSOLVER = solver
        try:
            m = hogp_simple.HOGP_simple(kernel.ARDKernel(3), 0.7, [d1, d2], variance_mode="eigen").double().to(DEV)
            Yr = Y.clone().requires_grad_(True)
            loss = m.log_likelihood(X, Yr)
            loss.backward()
            with torch.no_grad():
                mu, var = m.forward(X, Xt)
            res[solver] = (float(loss), Yr.grad.clone(), m.noise_variance.grad.clone(), m.kernel_list[0].length_scales.grad.clone(),
                           m.g.clone(), mu.clone(), var.clone())
        finally:
            hogp_simple.EIGENSOLVER = "ffgp"
            hogp_simple.eigen_pairs = own_pairs
    rel = lambda x, y: float((x - y).abs().max() / y.abs().max())
    b = res["rocsolver"]
    for own in ("ffgp", "jacobi"):
        a = res[own]
        assert abs(a[0] - b[0]) <= 1e-10 * abs(b[0]), own
        assert rel(a[1], b[1]) < 1e-8 and rel(a[2], b[2]) < 1e-8 and rel(a[3], b[3]) < 1e-7, own
        assert rel(a[4], b[4]) < 1e-8 and rel(a[5], b[5]) < 1e-8, own
    # (the reference's "variance" expression divides by the eigenvalues of the jitter-free K_x -- ~1e-16 here -- and is a
    #  different O(1e5) number for every basis of the near-null space: nothing to compare)


def test_hogp_blocks_from_host_threads_match_one_after_another():
    """functional.threaded_blocks: HOGP blocks driven from two host threads (own handle slot, own stream each -- ffgp_syevd waits
    for its bulge chasing on the host, so one thread cannot overlap two blocks) return exactly what the same blocks return one
    after another, forward values and the cached posterior pieces; the default handle is untouched by the workers' slots."""
    from fidelityfusion_amd import _lib
    from fidelityfusion_amd import functional as F
    from fidelityfusion_amd import kernel as K_
    from fidelityfusion_amd.hogp_simple import HOGP_simple
    n, D, modes = 448, 3, (6, 5)
    blocks = []
    for f in range(5):
        g = torch.Generator(device=DEV).manual_seed(100 + f)
        X = torch.rand((n, D), generator=g, device=DEV, dtype=torch.float64)
        Y = torch.randn((n,) + modes, generator=g, device=DEV, dtype=torch.float64)
        torch.manual_seed(f)
        blocks.append((HOGP_simple(K_.ARDKernel(D), 1.0, list(modes)).double().to(DEV), X, Y))
    with torch.no_grad():
        ref = [m.log_likelihood(X, Y).clone() for m, X, Y in blocks]
        ref_g = [m.g.clone() for m, _, _ in blocks]
        seen = []

        def run(i):
            m, X, Y = blocks[i]
            seen.append(getattr(_lib._tls, "slot", 0))
            return m.log_likelihood(X, Y)
        got = F.threaded_blocks([(lambda i=i: run(i)) for i in range(5)], nslots=2)
    torch.cuda.synchronize()
    assert sorted(set(seen)) == [1, 2] and getattr(_lib._tls, "slot", 0) == 0
    for f in range(5):
        assert torch.equal(got[f], ref[f])
        assert torch.equal(blocks[f][0].g, ref_g[f])


def test_threaded_blocks_raise_in_the_caller_and_keep_grad_mode():
    from fidelityfusion_amd import functional as F
    X = torch.rand((64, 2), device=DEV, dtype=torch.float64)
    w = torch.ones(2, device=DEV, dtype=torch.float64)
    amp = torch.ones(1, device=DEV, dtype=torch.float64)
    Y = torch.ones((64, 1), device=DEV, dtype=torch.float64)

    def bad():   # a covariance that is not positive definite: the raise must arrive in the calling thread
        return F.nlml(X, Y, w, amp, diag_add=torch.tensor([-5.0], device=DEV, dtype=torch.float64), clamp=1e-30)

    def good():
        assert not torch.is_grad_enabled()
        return F.nlml(X, Y, w, amp, diag_add=torch.tensor([0.1], device=DEV, dtype=torch.float64), clamp=1e-30)
    with torch.no_grad():
        with pytest.raises(torch.linalg.LinAlgError):
            F.threaded_blocks([good, bad, good], nslots=2)
        vals = F.threaded_blocks([good, good, good], nslots=3)
        one = good()
        # a call from inside a worker runs inline on that worker's slot (the worker slots are not re-entrant)
        from fidelityfusion_amd import _lib
        inner = F.threaded_blocks([lambda: (_lib.current_slot(), F.threaded_blocks([good, lambda: _lib.current_slot()], nslots=2)),
                                   lambda: (_lib.current_slot(), None)], nslots=2)
    assert all(torch.equal(v, one) for v in vals)
    assert inner[0][0] == 1 and inner[1][0] == 2 and inner[0][1][1] == 1 and torch.equal(inner[0][1][0], one)


def test_threaded_blocks_forward_with_gradients_then_backward_in_the_caller():
    """Likelihood blocks evaluated with gradients enabled in worker threads, backward() called afterwards from the calling thread
    (the documented order): every loss and every parameter gradient equals the one-after-another run.  The slot a block ran under is
    resolved in its worker thread and carried into backward, which runs on autograd's thread."""
    from oracle import gp_oracle as O
    from fidelityfusion_amd import functional as F
    from fidelityfusion_amd import kernel
    from fidelityfusion_amd.cigp_v10 import cigp
    data = []
    for f, (n, D, d) in enumerate([(900, 4, 3), (1300, 6, 1), (700, 3, 8), (1100, 5, 2), (300, 2, 1)]):
        X, Y = O.synthetic_xy(n, D, d, seed=30 + f)
        data.append((torch.tensor(X, device=DEV), torch.tensor(Y, device=DEV), D))

    def run(threaded):
        models = [cigp(kernel.ARDKernel(D), 0.7).to(DEV) for (_, _, D) in data]
        fns = [(lambda m=m, X=X, Y=Y: -m.negative_log_likelihood(X, Y)) for m, (X, Y, _) in zip(models, data)]
        losses = F.threaded_blocks(fns, nslots=3) if threaded else [fn() for fn in fns]
        torch.stack([l.reshape(()) for l in losses]).sum().backward()
        return ([l.detach().clone() for l in losses], [m.kernel.length_scales.grad.clone() for m in models],
                [m.log_beta.grad.clone() for m in models])
    l0, g0, b0 = run(False)
    for _ in range(4):
        l1, g1, b1 = run(True)
        for a, b in zip(l0 + g0 + b0, l1 + g1 + b1):
            assert float((a - b).abs().max()) <= 1e-12 * float(a.abs().max()), (a, b)
