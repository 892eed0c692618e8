"""The hand-written two-stage symmetric eigensolver (ffgp_syevd) through the C ABI: every stage
against the properties that define it and against the CPU restatement's conventions (oracle/eigh_twostage.py), the whole solver
against LAPACK (numpy.linalg.eigh) -- the routine the reference calls through torch.linalg.eigh
(two_fidelity_models/hogp_simple.py:15-19,97-100)."""
import numpy as np
import pytest
import torch

pytestmark = [pytest.mark.gpu, pytest.mark.noisy]   # (noisy: beside a background load by default, tests/conftest.py)
DEV = "cuda:0"


def _kernel_matrix(n, D, ls, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    X = torch.rand((n, D), generator=g, dtype=torch.float64)
    d = torch.cdist(X / ls, X / ls)
    return torch.exp(-0.5 * d * d)


def _random_sym(n, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    M = torch.randn((n, n), generator=g, dtype=torch.float64)
    return M + M.T


def _band_dense(AB, b=32):
    """dense symmetric matrix from the band storage AB[c, r - c]"""
    AB = AB.cpu().numpy()
    n = AB.shape[0]
    B = np.zeros((n, n))
    for k in range(b + 1):
        v = AB[:n - k, k]
        B[np.arange(k, n), np.arange(0, n - k)] = v
        B[np.arange(0, n - k), np.arange(k, n)] = v
    return B


@pytest.mark.parametrize("n,kind", [(64, "rand"), (128, "rand"), (576, "kern"), (1088, "rand"), (1600, "kern")])
def test_stage1_band_reduction(n, kind):
    """sy2sb: the band has bandwidth 32, the eigenvalues of A, and A = Q1 B Q1^T with an orthogonal Q1 (through ormq1)"""
    from fidelityfusion_amd import eigh as E
    A = (_random_sym(n, n) if kind == "rand" else _kernel_matrix(n, 3, 0.6, n))
    AB, Y = E.sy2sb(A.to(DEV))
    assert float(AB[:, 33:].abs().max()) == 0.0
    B = _band_dense(AB)
    ref = np.linalg.eigvalsh(A.numpy())
    scale = np.abs(ref).max()
    assert np.abs(np.linalg.eigvalsh(B) - ref).max() <= 5e-14 * scale
    Q1 = E.ormq1(Y, torch.eye(n, dtype=torch.float64, device=DEV)).cpu().numpy()
    assert np.abs(Q1.T @ Q1 - np.eye(n)).max() <= 1e-13
    assert np.abs(Q1 @ B @ Q1.T - A.numpy()).max() <= 2e-13 * scale


@pytest.mark.parametrize("opts", [{"sb_av_gemm": 1}, {"sb_qr4": 1}, {"sb_lookahead": 1}, {"sb_qr4": 1, "sb_lookahead": 1}],
                         ids=lambda o: "+".join(sorted(o)))
def test_stage1_band_reduction_alternative_kernels(opts):
    from conftest import need_dev_options
    need_dev_options()
    """the band reduction's off-by-default forms (A Y on the general GEMM; leaf QRs on 256-thread workgroups; the next panel's QR
    chain on the side stream) meet the same properties -- four leaves per panel at first, the last one ragged"""
    from fidelityfusion_amd import _lib
    from fidelityfusion_amd import eigh as E
    n = 1600
    A = _kernel_matrix(n, 3, 0.6, n)
    for k, v in opts.items():
        _lib.set_option(k, v)
    try:
        AB, Y = E.sy2sb(A.to(DEV))
        Q1 = E.ormq1(Y, torch.eye(n, dtype=torch.float64, device=DEV)).cpu().numpy()
    finally:
        for k in opts:
            _lib.set_option(k, 0)
    assert float(AB[:, 33:].abs().max()) == 0.0
    B = _band_dense(AB)
    ref = np.linalg.eigvalsh(A.numpy())
    scale = np.abs(ref).max()
    assert np.abs(np.linalg.eigvalsh(B) - ref).max() <= 5e-14 * scale
    assert np.abs(Q1.T @ Q1 - np.eye(n)).max() <= 1e-13
    assert np.abs(Q1 @ B @ Q1.T - A.numpy()).max() <= 2e-13 * scale


@pytest.mark.parametrize("n,kind", [(64, "rand"), (128, "rand"), (192, "kern"), (576, "kern"), (1088, "rand"), (1600, "kern"), (2304, "rand")])
def test_stage1_band_reduction_on_the_lower_triangle(n, kind):
    """the default form for n >= 6144 (`sb_lower`: the trailing update writes, and A22 Y reads, the lower triangle only --
    sy2sb_av_sym), forced at small sizes: same properties as the full form, its band within rounding of the full form's, the whole
    solver on it; row blocks of 128 and k chunks at every alignment (n - 32 (p + 1) rows trail panel p: ragged last row blocks, chunks
    cut by the diagonal, one-chunk and one-block panels)"""
    from fidelityfusion_amd import _lib
    from fidelityfusion_amd import eigh as E
    A = (_random_sym(n, n) if kind == "rand" else _kernel_matrix(n, 3, 0.6, n))
    AB_full, _ = E.sy2sb(A.to(DEV))
    _lib.set_option("sb_lower_min_n", 64)
    try:
        AB, Y = E.sy2sb(A.to(DEV))
        Q1 = E.ormq1(Y, torch.eye(n, dtype=torch.float64, device=DEV)).cpu().numpy()
        lam, U = E.eigh(A.to(DEV))
    finally:
        _lib.set_option("sb_lower_min_n", 6144)
    assert float(AB[:, 33:].abs().max()) == 0.0
    B = _band_dense(AB)
    ref = np.linalg.eigvalsh(A.numpy())
    scale = np.abs(ref).max()
    assert np.abs(np.linalg.eigvalsh(B) - ref).max() <= 5e-14 * scale
    assert np.abs(Q1.T @ Q1 - np.eye(n)).max() <= 1e-13
    assert np.abs(Q1 @ B @ Q1.T - A.numpy()).max() <= 2e-13 * scale
    assert float((AB - AB_full).abs().max()) <= 1e-12 * scale        # same reflectors up to rounding: the two forms sum in different orders
    lam, U = lam.cpu().numpy(), U.cpu().numpy()
    assert np.abs(lam - ref).max() <= 5e-14 * scale
    assert np.abs(U.T @ U - np.eye(n)).max() <= 2e-13
    assert np.abs((U * lam) @ U.T - A.numpy()).max() <= 5e-13 * scale


def test_syevd_at_the_size_where_the_lower_triangle_form_is_the_default():
    """n = 6144 with the library's defaults (`sb_lower_min_n`): residual, orthogonality, and the spectrum against rocSOLVER's"""
    from fidelityfusion_amd import eigh as E
    n = 6144
    g = torch.Generator(device=DEV).manual_seed(6144)
    X = torch.rand((n, 6), generator=g, device=DEV, dtype=torch.float64)
    d = torch.cdist(X, X)
    K = torch.exp(-0.5 * d * d / 0.49) + 1e-3 * torch.eye(n, device=DEV, dtype=torch.float64)
    del d
    lam, U = E.eigh(K)
    scale = float(lam.abs().max())
    assert float((U.T @ U - torch.eye(n, device=DEV, dtype=torch.float64)).abs().max()) <= 2e-13
    assert float(((U * lam) @ U.T - K).abs().max()) <= 1e-12 * scale
    ref = torch.linalg.eigvalsh(K)
    assert float((lam - ref).abs().max()) <= 1e-12 * scale


@pytest.mark.parametrize("n", [64, 192, 640, 1216])
def test_stage2_bulge_chasing(n):
    """sb2st: tridiagonal with the band's eigenvalues; B = Q2 T Q2^T with an orthogonal Q2 (through ormq2)"""
    from fidelityfusion_amd import eigh as E
    g = np.random.default_rng(n)
    AB = np.zeros((n, 64))
    for k in range(33):
        AB[:n - k, k] = g.standard_normal(n - k)
    B = _band_dense(torch.from_numpy(AB))
    d, e, refl = E.sb2st(torch.from_numpy(AB).to(DEV))
    d, e = d.cpu().numpy(), e.cpu().numpy()
    T = np.diag(d) + np.diag(e[:-1], 1) + np.diag(e[:-1], -1)
    ref = np.linalg.eigvalsh(B)
    scale = np.abs(ref).max()
    assert np.abs(np.linalg.eigvalsh(T) - ref).max() <= 5e-14 * scale
    Q2 = E.ormq2(refl, torch.eye(n, dtype=torch.float64, device=DEV)).cpu().numpy()
    assert np.abs(Q2.T @ Q2 - np.eye(n)).max() <= 1e-13
    assert np.abs(Q2 @ T @ Q2.T - B).max() <= 2e-13 * scale


@pytest.mark.parametrize("n,kind", [(64, "rand"), (128, "rand"), (192, "lap"), (704, "rand"), (1024, "clustered"), (1344, "graded")])
def test_stage3_divide_and_conquer(n, kind):
    from fidelityfusion_amd import eigh as E
    g = np.random.default_rng(n)
    if kind == "rand":
        d, e = g.standard_normal(n), g.standard_normal(n)
    elif kind == "lap":
        d, e = 2.0 * np.ones(n), -np.ones(n)
    elif kind == "clustered":          # many equal eigenvalues: heavy deflation
        d, e = np.ones(n), 1e-14 * g.standard_normal(n)
        d[::7] = 2.0
    else:                              # graded like a kernel matrix's tridiagonal form
        d = np.exp(-np.arange(n) / 20.0) + 1e-17
        e = 0.3 * np.exp(-np.arange(n) / 20.0)
    W, Z = E.stedc(torch.from_numpy(d).to(DEV), torch.from_numpy(e).to(DEV))
    W, Z = W.cpu().numpy(), Z.cpu().numpy()
    T = np.diag(d) + np.diag(e[:-1], 1) + np.diag(e[:-1], -1)
    ref = np.linalg.eigvalsh(T)
    scale = np.abs(ref).max()
    assert np.all(np.diff(W) >= 0)
    assert np.abs(W - ref).max() <= 1e-14 * scale * max(1.0, np.sqrt(n) / 8)
    assert np.abs(Z.T @ Z - np.eye(n)).max() <= 2e-13
    assert np.abs((Z * W) @ Z.T - T).max() <= 1e-13 * scale


@pytest.mark.parametrize("n,kind", [(1, "rand"), (7, "rand"), (64, "kern"), (65, "rand"), (130, "kern"), (257, "kern1"), (600, "rand"),
                                    (1100, "kern"), (2048, "kern8")])
def test_syevd_vs_lapack(n, kind):
    """the whole solver: eigenvalues against LAPACK to 1e-13 ||A||, orthogonality, reconstruction, ascending order -- generic
    matrices, kernel matrices with numerically singular spectra (D = 1, 3, 8), sizes that need padding"""
    from fidelityfusion_amd import eigh as E
    if kind == "rand":
        A = _random_sym(n, n)
    elif kind == "kern1":
        A = _kernel_matrix(n, 1, 0.5, n)
    elif kind == "kern8":
        A = _kernel_matrix(n, 8, 1.0, n)
    else:
        A = _kernel_matrix(n, 3, 0.7, n)
    W, Z = E.eigh(A.to(DEV))
    W, Z = W.cpu().numpy(), Z.cpu().numpy()
    ref = np.linalg.eigvalsh(A.numpy())
    scale = max(np.abs(ref).max(), 1e-300)
    assert W.shape == (n,) and Z.shape == (n, n)
    assert np.all(np.diff(W) >= 0)
    assert np.abs(W - ref).max() <= 1e-13 * scale
    assert np.abs(Z.T @ Z - np.eye(n)).max() <= 5e-13
    assert np.linalg.norm((Z * W) @ Z.T - A.numpy()) <= 1e-13 * np.linalg.norm(A.numpy()) * max(1.0, np.sqrt(n) / 8)


def test_syevd_reads_the_lower_triangle_and_keeps_its_input():
    from fidelityfusion_amd import eigh as E
    n = 200
    A = _random_sym(n, 3).to(DEV)
    junk = A.clone()
    junk[np.triu_indices(n, 1)] = 7.0
    keep = junk.clone()
    W, _ = E.eigh(junk)
    assert torch.equal(junk, keep)
    ref = torch.linalg.eigvalsh(A.cpu())
    assert float((W.cpu() - ref).abs().max()) <= 1e-13 * float(ref.abs().max())


@pytest.mark.parametrize("qr4", [0, 1])
def test_syevd_three_level_tsqr_above_8192(qr4):
    """n > 8224: the panels have more than 16 leaves of 512 rows, so the TSQR gets its middle level (the leaf kernel on stacks of 16 R
    factors; qr4: the 256-thread leaf kernel in both places).  Comparator at this size: rocSOLVER on the same device (LAPACK on the
    host would take minutes)"""
    from fidelityfusion_amd import _lib
    from fidelityfusion_amd import eigh as E
    n = 8500
    g = torch.Generator(device=DEV).manual_seed(1)
    X = torch.rand((n, 6), generator=g, device=DEV, dtype=torch.float64)
    d = torch.cdist(X, X)
    K = torch.exp(-0.5 * d * d)
    del d
    if qr4:
        from conftest import need_dev_options
        need_dev_options()
        _lib.set_option("sb_qr4", qr4)
    try:
        W, Z = E.eigh(K)
    finally:
        if qr4:
            _lib.set_option("sb_qr4", 0)
    ref = torch.linalg.eigvalsh(K)
    scale = float(ref.abs().max())
    assert float((W - ref).abs().max()) <= 1e-13 * scale
    assert float((Z.T @ Z - torch.eye(n, device=DEV, dtype=torch.float64)).abs().max()) <= 5e-13
    assert float(torch.linalg.matrix_norm((Z * W) @ Z.T - K)) <= 2e-13 * float(torch.linalg.matrix_norm(K))


def test_solver_is_unchanged_by_a_second_solver_running_beside_it():
    """Two host threads, each with its own handle and stream (functional.threaded_blocks), run band reductions and whole eigh calls at a
    size where the kernels of the two really share the chip: every result must equal, bit for bit, what the same call returns alone.
    (A missing barrier in sy2sb_form_y passed every single-stream test of the round and failed here in most repeats: a wave delayed by
    the other solver's waves read LDS operands that a faster wave had already refilled.)"""
    from fidelityfusion_amd import eigh as E
    from fidelityfusion_amd import functional as F
    n, nb = 8192, 4
    Ks = []
    for f in range(nb):
        g = torch.Generator(device=DEV).manual_seed(40 + f)
        X = torch.rand((n, 8), generator=g, device=DEV, dtype=torch.float64)
        d = torch.cdist(X, X)
        Ks.append(torch.exp(-0.5 * d * d))
        del d
    with torch.no_grad():
        for name, fn, reps in (("sy2sb", E.sy2sb, 4), ("eigh", E.eigh, 2)):
            ref = [fn(K) for K in Ks]
            torch.cuda.synchronize()
            for rep in range(reps):
                got = F.threaded_blocks([(lambda K=K: fn(K)) for K in Ks], nslots=2)
                torch.cuda.synchronize()
                for f in range(nb):
                    for j, (a, b) in enumerate(zip(ref[f], got[f])):
                        assert torch.equal(a, b), "%s, repeat %d, matrix %d, output %d: %d entries differ, max %.2e" % (
                            name, rep, f, j, int((a != b).sum()), float((a - b).abs().max()))
                del got
            del ref


def test_eigensolver_and_likelihood_chains_side_by_side():
    """Different kinds of work sharing the chip: one host thread runs eigh at N = 4096 and 8192, the other a stream of NLML forwards --
    blocked factorisations at N = 4096 and 1500, the finishing-kernel size 100, the one-kernel size 32, a Sum(Linear, Matern) kernel
    -- each on its own handle and stream.  Every result must equal the solo result bit for bit, repeat after repeat."""
    from fidelityfusion_amd import eigh as E
    from fidelityfusion_amd import functional as F
    from fidelityfusion_amd import kernel as K_
    from fidelityfusion_amd.cigp_v10 import cigp
    Ks = []
    for n in (8192, 4096):
        g = torch.Generator(device=DEV).manual_seed(n)
        X = torch.rand((n, 8), generator=g, device=DEV, dtype=torch.float64)
        d = torch.cdist(X, X)
        Ks.append(torch.exp(-0.5 * d * d))
        del d
    gp = []
    for f, (n, D, d) in enumerate([(4096, 8, 1), (1500, 5, 3), (100, 3, 2), (32, 2, 1), (2048, 4, 2)]):
        g = torch.Generator(device=DEV).manual_seed(7 + f)
        X = torch.rand((n, D), generator=g, device=DEV, dtype=torch.float64)
        Y = torch.randn((n, d), generator=g, device=DEV, dtype=torch.float64)
        torch.manual_seed(f)
        kern = K_.SumKernel(K_.LinearKernel(D), K_.MaternKernel(D)) if f == 4 else K_.ARDKernel(D)
        gp.append((cigp(kern, 0.5).double().to(DEV), X, Y))

    def solver():
        return [E.eigh(K) for K in Ks]

    def likelihoods():
        out = []
        for _ in range(6):
            out += [m.negative_log_likelihood(X, Y) for m, X, Y in gp]
        return torch.stack([o.reshape(()) for o in out])
    from fidelityfusion_amd import _lib
    with torch.no_grad():
        # the solo results through the same code path as the workers: under their handle slots (off slot 0 the likelihood modules do not
        # take the raw-parameter route, which agrees with this one to rounding only), one after the other
        with _lib.thread_slot(1):
            ref_e = solver()
        torch.cuda.synchronize()
        with _lib.thread_slot(2):
            ref_l = likelihoods()
        torch.cuda.synchronize()
        for rep in range(3):
            got_e, got_l = F.threaded_blocks([solver, likelihoods], nslots=2)
            torch.cuda.synchronize()
            assert torch.equal(got_l, ref_l), "repeat %d: likelihood values differ by up to %.2e" % (rep, float((got_l - ref_l).abs().max()))
            for (w0, z0), (w1, z1) in zip(ref_e, got_e):
                assert torch.equal(w0, w1) and torch.equal(z0, z1), "repeat %d: eigenpairs differ" % rep


@pytest.mark.noisy
@pytest.mark.parametrize("n", [256, 1024, 2048])
def test_xcd_local_chase_is_the_chip_wide_chase_bit_for_bit(n):
    """round 5: bands of up to 2048 columns run their bulge chase with every working wave on ONE XCD (the kernel checks HW_REG_XCC_ID,
    sweeps are handed out by a ticket) and hand the band over through that XCD's L2 (plain stores, L1-bypassing loads).  Only the cache
    policy differs from the chip-wide form: eigenvalues and eigenvectors must be identical to the last bit, also beside a background load"""
    from fidelityfusion_amd import _lib
    from fidelityfusion_amd import eigh as E
    g = torch.Generator(device=DEV).manual_seed(n)
    X = torch.rand((n, 5), generator=g, device=DEV, dtype=torch.float64)
    d = torch.cdist(X, X)
    K = torch.exp(-0.5 * d * d / 0.36)
    res = {}
    for xl in (1, 0, 1):
        _lib.set_option("chase_xl", xl, 0)
        try:
            W, Z = E.eigh(K)
        finally:
            _lib.set_option("chase_xl", 1, 0)
        res.setdefault(xl, []).append((W.clone(), Z.clone()))
    assert torch.equal(res[1][0][0], res[0][0][0]) and torch.equal(res[1][0][1], res[0][0][1])
    assert torch.equal(res[1][0][0], res[1][1][0]) and torch.equal(res[1][0][1], res[1][1][1])
    lam = torch.linalg.eigvalsh(K)
    assert float((res[1][0][0] - lam).abs().max()) <= 1e-11 * float(lam.abs().max())
    # no wave of the XCD-local launch lands on the chosen XCD (partition modes, a CU-masked stream, a part with fewer XCDs; here: an id
    # the chip does not have): the ticket never moves and the chip-wide launch behind it chases every sweep -- same eigenpairs, not an
    # un-reduced band returned with FFGP_OK
    _lib.set_option("chase_xcc", 12, 0)
    try:
        W, Z = E.eigh(K)
    finally:
        _lib.set_option("chase_xcc", 0, 0)
    assert torch.equal(W, res[0][0][0]) and torch.equal(Z, res[0][0][1])


@pytest.mark.noisy
@pytest.mark.parametrize("n", [300, 1024, 1500])
def test_back_transformation_on_eight_waves_is_the_four_wave_form_bit_for_bit(n):
    """round 5: from 8192 columns on, Z <- Q2 Z runs 32-column slabs on EIGHT waves (waves 0-3 the first 16-column tile, waves 4-7 the second:
    q2_apply_wave4<2, true>); per tile the arithmetic is the four-wave kernel's, so eigenvalues and eigenvectors must be identical to the last
    bit -- forced here at small sizes (ragged last slab, matrices that are not a multiple of 32 columns) through option q2_split_min_cols"""
    from fidelityfusion_amd import _lib
    from fidelityfusion_amd import eigh as E
    g = torch.Generator(device=DEV).manual_seed(n + 1)
    X = torch.rand((n, 4), generator=g, device=DEV, dtype=torch.float64)
    d = torch.cdist(X, X)
    K = torch.exp(-0.5 * d * d / 0.25)
    res = {}
    for split_from in (0, 1 << 30, 0):
        _lib.set_option("q2_split_min_cols", split_from, 0)
        try:
            W, Z = E.eigh(K)
        finally:
            _lib.set_option("q2_split_min_cols", 8192, 0)
        res.setdefault(split_from, []).append((W.clone(), Z.clone()))
    assert torch.equal(res[0][0][0], res[1 << 30][0][0]) and torch.equal(res[0][0][1], res[1 << 30][0][1])
    assert torch.equal(res[0][0][0], res[0][1][0]) and torch.equal(res[0][0][1], res[0][1][1])
    lam = torch.linalg.eigvalsh(K)
    assert float((res[0][0][0] - lam).abs().max()) <= 1e-11 * float(lam.abs().max())
    Z = res[0][0][1]
    assert float((Z.T @ Z - torch.eye(n, device=DEV, dtype=torch.float64)).abs().max()) < 1e-11

