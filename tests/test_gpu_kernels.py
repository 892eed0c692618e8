"""GPU: each HIP kernel family against numpy on the same seeded inputs, called through the C ABI (ctypes).
These are the per-kernel tests (layer 2 of the pyramid in SURVEY.md section 4); end-to-end parity with the
oracle and the golden fixtures is in test_gpu_parity.py."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def ff():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from fidelityfusion_amd import _lib
    h = _lib.handle(0)
    _lib.bind_stream(h, 0)
    return _lib, h


def dev(a):
    return torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device="cuda:0")


def ptr(t):
    return C.c_void_p(t.data_ptr())


def relerr(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


# ------------------------------------------------------------------------------------------------ GEMM
def run_gemm(ff, opa, opb, lower, tri, A, B, C0, alpha, beta, pad=(0, 0, 0)):
    """A is op(A) [m,k]; B is op(B) [k,n]; returns C.  Storage follows the op flags; pads add to the lds."""
    _lib, h = ff
    m, k = A.shape
    n = B.shape[1]
    As = A if opa == 0 else A.T
    Bs = B.T if opb == 0 else B
    def padded(M, extra):
        buf = np.full((M.shape[0], M.shape[1] + extra), np.nan)
        buf[:, :M.shape[1]] = M
        return buf
    Ab, Bb, Cb = padded(As, pad[0]), padded(Bs, pad[1]), padded(C0, pad[2])
    Ad, Bd, Cd = dev(Ab), dev(Bb), dev(Cb)
    rc = _lib.lib.ffgp_gemm(h, opa, opb, lower, tri, ptr(Ad), Ab.shape[1], ptr(Bd), Bb.shape[1], ptr(Cd), Cb.shape[1],
                            m, n, k, alpha, beta)
    assert rc == 0, rc
    torch.cuda.synchronize()
    out = Cd.cpu().numpy()
    assert np.isnan(out[:, C0.shape[1]:]).all() or pad[2] == 0, "gemm wrote into the ldc padding"
    return out[:, :C0.shape[1]]


@pytest.fixture(params=[32, 64, 128])
def tile(request, ff):
    """run the test once per GEMM tile shape (the launcher picks automatically in production)"""
    _lib, h = ff
    assert _lib.lib.ffgp_set_option(h, b"gemm_tile", float(request.param)) == 0
    yield request.param
    _lib.lib.ffgp_set_option(h, b"gemm_tile", 0.0)


@pytest.mark.parametrize("opa,opb", [(0, 0), (0, 1), (1, 0), (1, 1)])
@pytest.mark.parametrize("m,n,k", [(128, 128, 16), (256, 128, 64), (100, 77, 21), (1, 1, 1), (300, 200, 130), (129, 257, 48)])
def test_gemm_layouts(ff, tile, opa, opb, m, n, k):
    rng = np.random.default_rng(m * 7 + n * 3 + k + opa * 2 + opb)
    A = rng.standard_normal((m, k))
    B = rng.standard_normal((k, n))      # asymmetric operands: a swapped row/col map cannot hide
    C0 = rng.standard_normal((m, n))
    out = run_gemm(ff, opa, opb, 0, 0, A, B, C0, -1.0, 1.0, pad=(2, 4, 6))
    assert relerr(out, C0 - A @ B) < 1e-13, (opa, opb, m, n, k)
    out = run_gemm(ff, opa, opb, 0, 0, A, B, np.full((m, n), np.nan), 2.0, 0.0, pad=(0, 0, 0))
    assert relerr(out, 2.0 * A @ B) < 1e-13, "beta == 0 must not read C"


def test_gemm_identity_asymmetric(ff):
    """A = I with an asymmetric B: catches a transposed accumulator map (cdna guide, section 3)."""
    B = np.arange(128 * 128, dtype=np.float64).reshape(128, 128)
    out = run_gemm(ff, 0, 1, 0, 0, np.eye(128), B, np.zeros((128, 128)), 1.0, 0.0)
    assert np.array_equal(out, B)
    out = run_gemm(ff, 0, 0, 0, 0, np.eye(128), B, np.zeros((128, 128)), 1.0, 0.0)
    assert np.array_equal(out, B)


def test_gemm_unaligned_operands(ff):
    """odd leading dimensions fall back to 8-byte loads"""
    rng = np.random.default_rng(5)
    A, B = rng.standard_normal((70, 33)), rng.standard_normal((33, 5))
    for opa in (0, 1):
        for opb in (0, 1):
            out = run_gemm(ff, opa, opb, 0, 0, A, B, np.zeros((70, 5)), 1.0, 0.0, pad=(1, 1, 0) if (opa + opb) % 2 else (0, 0, 0))
            assert relerr(out, A @ B) < 1e-13


@pytest.mark.parametrize("m,n,k", [(128, 128, 32), (300, 300, 64), (700, 300, 128), (1300, 1300, 16), (1100, 129, 40)])
@pytest.mark.parametrize("ops", [(0, 0), (1, 1)])
def test_gemm_lower_trapezoid(ff, tile, m, n, k, ops):
    rng = np.random.default_rng(m + n + k)
    A = rng.standard_normal((m, k))
    B = rng.standard_normal((k, n))
    C0 = rng.standard_normal((m, n))
    out = run_gemm(ff, ops[0], ops[1], 1, 0, A, B, C0, -1.0, 1.0, pad=(0, 0, 2))
    full = C0 - A @ B
    mask = np.tril(np.ones((m, n), dtype=bool))
    assert relerr(out[mask], full[mask]) < 1e-13
    assert np.array_equal(out[~mask], C0[~mask]), "strictly-upper part must not be written"


@pytest.mark.parametrize("case", [
    ("full", 0, 0, 2176, 2176, 40, 0), ("full", 0, 1, 2200, 2150, 24, 0), ("full", 1, 0, 2090, 2300, 33, 0), ("full", 1, 1, 2176, 2304, 16, 0),
    ("lower", 0, 0, 2944, 2944, 48, 0), ("lower", 0, 0, 3000, 3000, 20, 0), ("lower", 1, 1, 3500, 1300, 36, 0),
    ("lower", 1, 1, 2944, 2944, 2944, 1), ("full", 0, 1, 2300, 2176, 2300, 4)])
def test_gemm_split_tail(ff, case):
    """the split tail of the 128-tile launches (last tiles mod 256 handed out as 64 x 64 quarters): same bits as the
    unsplit launch, nothing written outside the region, ragged edges and triangular k ranges included"""
    _lib, h = ff
    mode, opa, opb, m, n, k, tri = case
    rng = np.random.default_rng(m + 3 * n + k)
    if tri == 1:     # LAUUM shape: X^T X for lower-triangular X, both MN-major
        X = np.tril(rng.standard_normal((m, m)))
        A, B = X.T.copy(), X
    elif tri == 4:   # A lower-triangular (K-major)
        A, B = np.tril(rng.standard_normal((m, m))), rng.standard_normal((m, n))
    else:
        A, B = rng.standard_normal((m, k)), rng.standard_normal((k, n))
    C0 = rng.standard_normal((m, n))
    outs = []
    for rem_max in (255.0, 0.0):
        assert _lib.lib.ffgp_set_option(h, b"split_rem_max", rem_max) == 0
        outs.append(run_gemm(ff, opa, opb, 1 if mode == "lower" else 0, tri, A, B, C0, -1.0, 1.0, pad=(0, 0, 2)))
    _lib.lib.ffgp_set_option(h, b"split_rem_max", 180.0)
    assert np.array_equal(outs[0], outs[1]), "split and unsplit launches must agree bit for bit"
    full = C0 - A @ B
    mask = np.tril(np.ones((m, n), dtype=bool)) if mode == "lower" else np.ones((m, n), dtype=bool)
    assert relerr(outs[0][mask], full[mask]) < 1e-12
    assert np.array_equal(outs[0][~mask], C0[~mask])


@pytest.mark.parametrize("opa,opb", [(0, 0), (0, 1), (1, 0), (1, 1)])
@pytest.mark.parametrize("m,n,k", [(1, 4096, 4096), (1, 1000, 333), (3, 517, 2100), (8, 129, 64), (5, 9, 7), (2, 70, 5000)])
def test_gemm_few_output_rows(ff, opa, opb, m, n, k):
    """C[m x n] with m <= 8 (A^T = Gamma^T L^-1 of a single-output GP) runs as the transposed matrix-vector product: every
    operand layout, alpha / beta, padded leading dimensions left untouched, beta == 0 never reads C"""
    rng = np.random.default_rng(m + 5 * n + k + opa * 2 + opb)
    A = rng.standard_normal((m, k))
    B = rng.standard_normal((k, n))
    C0 = rng.standard_normal((m, n))
    scale = (np.abs(A) @ np.abs(B)).max()
    out = run_gemm(ff, opa, opb, 0, 0, A, B, C0, -1.0, 1.0, pad=(2, 4, 6))
    assert np.abs(out - (C0 - A @ B)).max() <= 1e-14 * scale * np.sqrt(k)
    out = run_gemm(ff, opa, opb, 0, 0, A, B, np.full((m, n), np.nan), 2.0, 0.0)
    assert np.abs(out - 2.0 * A @ B).max() <= 2e-14 * scale * np.sqrt(k)
    out2 = run_gemm(ff, opa, opb, 0, 0, A, B, np.full((m, n), np.nan), 2.0, 0.0)
    assert np.array_equal(out, out2)
    # TRI_LO_J: op(B)(k, j) = 0 for k < j (the inverse factor L^-1 of the likelihood's backward): the k loop may start at j
    Bl = np.tril(B)
    out = run_gemm(ff, opa, opb, 0, 2, A, Bl, np.full((m, n), np.nan), 1.0, 0.0, pad=(0, 2, 0))
    assert np.abs(out - A @ Bl).max() <= 1e-14 * scale * np.sqrt(k)


@pytest.mark.parametrize("opa,opb", [(0, 0), (0, 1), (1, 0), (1, 1)])
@pytest.mark.parametrize("m,n,k", [(256, 1, 16384), (256, 256, 8192), (70, 33, 5000), (64, 64, 2048 + 17), (1, 1, 4096), (130, 512, 3000)])
def test_gemm_split_k(ff, opa, opb, m, n, k):
    """thin products with a long k are cut along k (batched partial products + a fixed-order reduction): values, beta,
    untouched ldc padding, and bit-for-bit repeatability"""
    rng = np.random.default_rng(m + 5 * n + k + opa * 2 + opb)
    A = rng.standard_normal((m, k))
    B = rng.standard_normal((k, n))
    C0 = rng.standard_normal((m, n))
    ref = C0 - A @ B
    scale = np.abs(A) @ np.abs(B)
    out = run_gemm(ff, opa, opb, 0, 0, A, B, C0, -1.0, 1.0, pad=(2, 4, 6))
    assert np.abs(out - ref).max() <= 1e-14 * scale.max() * np.sqrt(k)
    out2 = run_gemm(ff, opa, opb, 0, 0, A, B, C0, -1.0, 1.0, pad=(2, 4, 6))
    assert np.array_equal(out, out2), "the reduction order is fixed: repeated launches agree bit for bit"
    out = run_gemm(ff, opa, opb, 0, 0, A, B, np.full((m, n), np.nan), 2.0, 0.0)
    assert np.abs(out - 2.0 * A @ B).max() <= 2e-14 * scale.max() * np.sqrt(k), "beta == 0 must not read C"


@pytest.mark.parametrize("opa,opb", [(0, 0), (0, 1), (1, 0), (1, 1)])
@pytest.mark.parametrize("m,n,k", [(1, 1, 1), (63, 1, 7), (64, 2, 128), (300, 3, 1000), (5000, 1, 1024), (4500, 5, 4099), (129, 8, 515),
                                   (1024, 1, 1024)])
def test_gemm_skinny(ff, opa, opb, m, n, k):
    """products with at most 8 output columns run on the matrix-vector kernels: every layout, ragged sizes, odd leading
    dimensions, beta, untouched padding"""
    rng = np.random.default_rng(m + 7 * n + k + opa * 2 + opb)
    A = rng.standard_normal((m, k))
    B = rng.standard_normal((k, n))
    C0 = rng.standard_normal((m, n))
    scale = (np.abs(A) @ np.abs(B)).max() + 1.0
    for pad in ((0, 0, 0), (1, 3, 5)):
        out = run_gemm(ff, opa, opb, 0, 0, A, B, C0, -1.0, 1.0, pad=pad)
        assert np.abs(out - (C0 - A @ B)).max() <= 1e-14 * scale * np.sqrt(k + 1.0)
    out = run_gemm(ff, opa, opb, 0, 0, A, B, np.full((m, n), np.nan), 0.5, 0.0)
    assert np.abs(out - 0.5 * A @ B).max() <= 1e-14 * scale * np.sqrt(k + 1.0), "beta == 0 must not read C"


def test_gemm_triangular_k_ranges(ff, tile):
    rng = np.random.default_rng(9)
    n = 520
    Lt = np.tril(rng.standard_normal((n, n)))
    M = rng.standard_normal((n, 200))
    # hi_i: A lower-triangular (K-major), C = A @ M
    out = run_gemm(ff, 0, 1, 0, 4, Lt, M, np.zeros((n, 200)), 1.0, 0.0)
    assert relerr(out, Lt @ M) < 1e-13
    # lo_j: B lower-triangular stored k x n, C = M^T-ish @ Lt
    M2 = rng.standard_normal((150, n))
    out = run_gemm(ff, 0, 1, 0, 2, M2, Lt, np.zeros((150, n)), 1.0, 0.0)
    assert relerr(out, M2 @ Lt) < 1e-13
    # lo_i with both MN-major, lower tiles: S = X^T X for lower-triangular X (LAUUM)
    out = run_gemm(ff, 1, 1, 1, 1, Lt.T, Lt, np.zeros((n, n)), 1.0, 0.0)
    ref = Lt.T @ Lt
    mask = np.tril(np.ones((n, n), dtype=bool))
    assert relerr(out[mask], ref[mask]) < 1e-13


# ------------------------------------------------------------------------------------------------ Cholesky
def spd(n, rng, cond=1e3):
    Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
    ev = np.logspace(0, np.log10(cond), n)
    return (Q * ev) @ Q.T


def potrf(ff, S, rows=None, naive=False, nb_outer=None):
    _lib, h = ff
    n = S.shape[0]
    m = 0 if rows is None else rows.shape[0]
    ld = (n + 1) // 2 * 2 + 2
    W = np.full((n + m, ld), np.nan)
    W[:n, :n] = np.tril(S) + np.triu(np.full((n, n), 777.0), 1)   # poison above the diagonal: must not be read
    if m:
        W[n:, :n] = rows
    Wd = dev(W)
    _lib.lib.ffgp_set_option(h, b"naive", 1.0 if naive else 0.0)
    if nb_outer:
        _lib.lib.ffgp_set_option(h, b"nb_outer", float(nb_outer))
    rc = _lib.lib.ffgp_potrf_rows(h, ptr(Wd), n, n + m, ld)
    _lib.lib.ffgp_set_option(h, b"naive", 0.0)
    _lib.lib.ffgp_set_option(h, b"nb_outer", 512.0)
    out = Wd.cpu().numpy()
    return rc, out, Wd, ld


@pytest.mark.parametrize("n", [1, 16, 17, 100, 128, 129, 300, 511, 640, 1000, 1537])
def test_potrf_matches_lapack(ff, tile, n):
    rng = np.random.default_rng(n)
    S = spd(n, rng)
    rc, out, _, _ = potrf(ff, S)
    assert rc == 0
    L = np.linalg.cholesky(S)
    got = np.tril(out[:n, :n])
    assert relerr(got, L) < 1e-11, n
    up = out[:n, :n][np.triu_indices(n, 1)]
    assert (up == 777.0).all(), "strictly-upper triangle was written"


@pytest.mark.noisy
@pytest.mark.parametrize("mode", [0, 1, 3, 4, 40])
def test_potrf_diag_kernel_variants(ff, mode):
    """the diagonal-block kernel in its forms -- 0: barrier version, 3: round-3 pipeline, 1: the same with the DP-ALU DPP pivot
    step (v_mov_b64_dpp / v_fmac_f64_dpp), 4: the default, round 6's ffgp_potrf_diag128_v4 (two barriers per 16-column stage, the inverse's
    rows in the shadow of the next block's pivots), 40: round 4's flag-driven pipeline v3 (option diag_v4 = 0) -- against
    LAPACK at sizes with full, partial and single blocks; the factor and the cached block inverses must be bit-stable over repeated
    calls while a background load shares the GPU (LDS flag protocols, no barrier after the role hand-out)"""
    import ctypes as C
    from conftest import need_dev_options
    from fidelityfusion_amd import _lib
    h = _lib.handle(0)
    if mode in (1, 3):                # the round-3 pipelines live in the development build only
        assert _lib.has_dev_options() or _lib.lib.ffgp_set_option(h, b"diag_v2", C.c_double(mode)) < 0    # (the shipped library refuses the key)
        need_dev_options()
    assert _lib.lib.ffgp_set_option(h, b"diag_v2", C.c_double(4 if mode == 40 else mode)) == 0
    assert _lib.lib.ffgp_set_option(h, b"diag_v4", C.c_double(0 if mode == 40 else 1)) == 0
    try:
        for n in (128, 100, 16, 129, 640, 1000, 1537, 4300):      # (1537, 4300: with the look-ahead's side stream and its hand-offs)
            rng = np.random.default_rng(10 * n + mode)
            S = spd(n, rng)
            ref = np.linalg.cholesky(S)
            first = None
            for rep in range(3):
                rc, out, _, _ = potrf(ff, S)
                assert rc == 0
                got = np.tril(out[:n, :n])
                assert relerr(got, ref) < 1e-11, (mode, n)
                assert (out[:n, :n][np.triu_indices(n, 1)] == 777.0).all(), "strictly-upper triangle was written"
                if first is None:
                    first = got
                else:
                    assert (got == first).all(), "factor changed between identical calls (mode %d, n %d)" % (mode, n)
        # a non-positive pivot inside a diagonal block is reported with its 1-based index by every variant
        S = spd(300, np.random.default_rng(7))
        S[200, 200] = -5.0
        rc, _, _, _ = potrf(ff, S)
        assert rc == 201, (mode, rc)
    finally:
        assert _lib.lib.ffgp_set_option(h, b"diag_v2", C.c_double(4)) == 0
        assert _lib.lib.ffgp_set_option(h, b"diag_v4", C.c_double(1)) == 0


@pytest.mark.parametrize("n", [4000, 6200])
def test_potrf_large_matches_lapack(ff, n):
    """many workgroups per launch, look-ahead stream active: catches cross-workgroup races the small cases cannot"""
    rng = np.random.default_rng(n)
    B = rng.standard_normal((n, 64))
    S = B @ B.T + np.diag(rng.random(n) + 0.5)
    for _ in range(2):
        rc, out, _, _ = potrf(ff, S)
        assert rc == 0
        assert relerr(np.tril(out[:n, :n]), np.linalg.cholesky(S)) < 1e-11


@pytest.mark.parametrize("n,nb", [(700, 128), (700, 256), (1300, 384)])
def test_potrf_outer_block_sizes(ff, n, nb):
    rng = np.random.default_rng(n + nb)
    S = spd(n, rng)
    rc, out, _, _ = potrf(ff, S, nb_outer=nb)
    assert rc == 0
    assert relerr(np.tril(out[:n, :n]), np.linalg.cholesky(S)) < 1e-11


@pytest.mark.parametrize("carry,lookahead", [(0, 1), (1, 1), (1, 0)])
@pytest.mark.parametrize("n,m,nb", [(1450, 5, 512), (2600, 0, 512), (1921, 70, 256), (640, 3, 128)])
def test_potrf_lookahead_forms(ff, n, m, nb, carry, lookahead):
    """both look-ahead forms of the blocked driver -- the carry form (a panel's updates also cover the next panel's first
    block; the main stream's S_bz strip gates the side stream's first update) and the S_a / S_b / S_ii form -- and the plain
    in-order driver, ragged sizes, passenger rows, panel widths down to one block: factor and rows against LAPACK, twice (a
    missed cross-stream dependency is a race, not a deterministic error)"""
    import scipy.linalg as sla
    _lib, h = ff
    rng = np.random.default_rng(n + m + nb)
    B = rng.standard_normal((n, 48))
    S = B @ B.T + np.diag(rng.random(n) + 0.5)
    R = rng.standard_normal((m, n)) if m else None
    L = np.linalg.cholesky(S)
    _lib.lib.ffgp_set_option(h, b"la_carry", float(carry))
    _lib.lib.ffgp_set_option(h, b"lookahead", float(lookahead))
    _lib.lib.ffgp_set_option(h, b"la_min_n", 0.0)          # (by default blocks this small are factored in order)
    try:
        for _ in range(2):
            rc, out, _, _ = potrf(ff, S, R, nb_outer=nb)
            assert rc == 0
            assert relerr(np.tril(out[:n, :n]), L) < 1e-11
            if m:
                assert relerr(out[n:, :n], sla.solve_triangular(L, R.T, lower=True).T) < 1e-10
    finally:
        _lib.lib.ffgp_set_option(h, b"la_carry", 2.0)
        _lib.lib.ffgp_set_option(h, b"lookahead", 1.0)
        _lib.lib.ffgp_set_option(h, b"la_min_n", 1024.0)


@pytest.mark.parametrize("n", [1024, 1025, 1100, 3585])
def test_potrf_around_the_lookahead_threshold(ff, n):
    """default options on both sides of la_min_n: 1024 rows are factored in order on one stream, more with the side stream (1025: a one-row
    second panel; 1100: a partial one)"""
    rng = np.random.default_rng(n)
    B = rng.standard_normal((n, 40))
    S = B @ B.T + np.diag(rng.random(n) + 0.5)
    rc, out, _, _ = potrf(ff, S, rng.standard_normal((3, n)))
    assert rc == 0
    assert relerr(np.tril(out[:n, :n]), np.linalg.cholesky(S)) < 1e-11


@pytest.mark.parametrize("n,m", [(129, 0), (300, 0), (700, 45), (1537, 3), (4000, 130)])
def test_potrf_block_trsm_kernel_is_the_general_gemm_bit_for_bit(ff, n, m):
    """the chain's own TRSM kernel (ffgp_trsm128_kernel: the triangle's zero k-steps skipped, operands re-dealt across lanes) against the
    same factorisation with the TRSM on the general GEMM (option trsm128 = 0): every element of the factor and of the passenger rows
    identical, ragged row tails and a partial last block (which stays on the general GEMM) included"""
    _lib, h = ff
    rng = np.random.default_rng(n + m)
    B = rng.standard_normal((n, 56))
    S = B @ B.T + np.diag(rng.random(n) + 0.5)
    R = rng.standard_normal((m, n)) if m else None
    outs = {}
    try:
        for on in (1, 0):
            assert _lib.lib.ffgp_set_option(h, b"trsm128", float(on)) == 0
            rc, out, _, _ = potrf(ff, S, R)
            assert rc == 0
            outs[on] = out
    finally:
        _lib.lib.ffgp_set_option(h, b"trsm128", 1.0)
    low = np.tril_indices(n)
    assert np.array_equal(outs[1][:n, :n][low], outs[0][:n, :n][low])
    assert np.array_equal(outs[1][n:, :n], outs[0][n:, :n])
    assert relerr(np.tril(outs[1][:n, :n]), np.linalg.cholesky(S)) < 1e-11


def test_potrf_polite_64_tile_updates_change_nothing_but_the_occupancy(ff):
    """the carry-form look-ahead launches its 64-tile trailing updates with 60 KiB of unused LDS (two workgroups per CU instead of four, so
    the chain's kernels find room: option polite64_pad_kb): same tiles, same arithmetic -- the factor and the passenger rows must be
    identical to the last bit with and without it"""
    _lib, h = ff
    n, m = 4700, 40
    rng = np.random.default_rng(n)
    B = rng.standard_normal((n, 48))
    S = B @ B.T + np.diag(rng.random(n) + 0.5)
    R = rng.standard_normal((m, n))
    outs = {}
    try:
        for pad in (60, 0):
            assert _lib.lib.ffgp_set_option(h, b"polite64_pad_kb", float(pad)) == 0
            rc, out, _, _ = potrf(ff, S, R)
            assert rc == 0
            outs[pad] = out
    finally:
        _lib.lib.ffgp_set_option(h, b"polite64_pad_kb", 60.0)
    low = np.tril_indices(n)
    assert np.array_equal(outs[60][:n, :n][low], outs[0][:n, :n][low])
    assert np.array_equal(outs[60][n:, :n], outs[0][n:, :n])
    assert relerr(np.tril(outs[60][:n, :n]), np.linalg.cholesky(S)) < 1e-11
    assert _lib.lib.ffgp_set_option(h, b"polite64_pad_kb", 65.0) != 0      # (more than 64 KiB of dynamic LDS is not requested)


def test_potrf_value_handoffs_are_the_event_handoffs_bit_for_bit(ff):
    """the look-ahead's cross-stream hand-offs as values in device memory (hipStreamWriteValue32 / hipStreamWaitValue32, option ho_values),
    the "panel complete" word written by the next diagonal-block kernel (ho_defer), and the event pairs they replace: the same launches in
    the same order on the same data -- only the waiting differs -- so the factor and the passenger rows are identical to the last bit;
    repeated calls reuse the words with growing sequence numbers"""
    _lib, h = ff
    n, m = 4700, 40
    rng = np.random.default_rng(n + 1)
    B = rng.standard_normal((n, 48))
    S = B @ B.T + np.diag(rng.random(n) + 0.5)
    R = rng.standard_normal((m, n))
    outs = {}
    try:
        for key, (hv, hd) in {"events": (0, 0), "values": (1, 0), "deferred by the diagonal block": (1, 1), "deferred": (1, 2), "deferred again": (1, 2)}.items():
            assert _lib.lib.ffgp_set_option(h, b"ho_values", float(hv)) == 0
            assert _lib.lib.ffgp_set_option(h, b"ho_defer", float(hd)) == 0
            rc, out, _, _ = potrf(ff, S, R)
            assert rc == 0
            outs[key] = out
    finally:
        _lib.lib.ffgp_set_option(h, b"ho_values", 1.0)
        _lib.lib.ffgp_set_option(h, b"ho_defer", 2.0)
    low = np.tril_indices(n)
    for key in ("values", "deferred by the diagonal block", "deferred", "deferred again"):
        assert np.array_equal(outs[key][:n, :n][low], outs["events"][:n, :n][low]), key
        assert np.array_equal(outs[key][n:, :n], outs["events"][n:, :n]), key
    assert relerr(np.tril(outs["deferred"][:n, :n]), np.linalg.cholesky(S)) < 1e-11


def test_potrf_lookahead_form_changes_over_inside_one_factorisation(ff):
    """la_carry_n / la_carry_rows: a block of more than la_carry_n rows starts with round-1 iterations (S_a on the chain's stream) and changes over to carry
    iterations once its trailing matrix is small enough; the first carry iteration meets a panel that carried nothing.  Exercised at a
    small size by lowering the threshold: every choice of the change-over point (none, first iteration, in the middle, always carry) gives
    LAPACK's factor, and the same passenger rows to rounding"""
    _lib, h = ff
    n, m = 5000, 24
    rng = np.random.default_rng(n + 2)
    B = rng.standard_normal((n, 40))
    S = B @ B.T + np.diag(rng.random(n) + 0.5)
    R = rng.standard_normal((m, n))
    L = np.linalg.cholesky(S)
    want_rows = np.linalg.solve(L, R.T).T
    outs = {}
    try:
        assert _lib.lib.ffgp_set_option(h, b"la_carry_n", 0.0) == 0          # (no block is small enough to be carry form throughout)
        for rows_ in (0, 2048, 3000, 4488, 12288):
            assert _lib.lib.ffgp_set_option(h, b"la_carry_rows", float(rows_)) == 0
            rc, out, _, _ = potrf(ff, S, R)
            assert rc == 0, rows_
            outs[rows_] = out
            assert relerr(np.tril(out[:n, :n]), L) < 1e-11, rows_
            assert relerr(out[n:, :n], want_rows) < 1e-9, rows_
    finally:
        _lib.lib.ffgp_set_option(h, b"la_carry_rows", 8192.0)
        _lib.lib.ffgp_set_option(h, b"la_carry_n", 12288.0)
    assert _lib.lib.ffgp_set_option(h, b"la_carry_rows", -1.0) != 0


def test_potrf_naive_kernels_agree(ff):
    rng = np.random.default_rng(2)
    S = spd(200, rng)
    rows = rng.standard_normal((3, 200))
    rc1, o1, _, _ = potrf(ff, S, rows, naive=False)
    rc2, o2, _, _ = potrf(ff, S, rows, naive=True)
    assert rc1 == 0 and rc2 == 0
    assert relerr(np.tril(o1[:200, :200]), np.tril(o2[:200, :200])) < 1e-11
    assert relerr(o1[200:, :200], o2[200:, :200]) < 1e-10


@pytest.mark.parametrize("n,m", [(100, 1), (257, 7), (640, 130), (1000, 300)])
def test_potrf_passenger_rows(ff, n, m):
    """rows below Sigma come out as rows @ L^-T  (Gamma^T, V^T of the fused paths)"""
    import scipy.linalg as sla
    rng = np.random.default_rng(n + m)
    S = spd(n, rng)
    R = rng.standard_normal((m, n))
    rc, out, _, _ = potrf(ff, S, R)
    assert rc == 0
    L = np.linalg.cholesky(S)
    ref = sla.solve_triangular(L, R.T, lower=True).T
    assert relerr(out[n:, :n], ref) < 1e-10


@pytest.mark.parametrize("n,bad", [(50, 1), (300, 130), (300, 300), (640, 513)])
def test_potrf_reports_first_bad_pivot(ff, n, bad):
    rng = np.random.default_rng(n + bad)
    S = spd(n, rng, cond=10.0)
    # make the leading minor of order `bad` singular/indefinite, earlier minors stay PD
    L = np.linalg.cholesky(S)
    L[bad - 1, bad - 1] = 0.0
    S2 = L @ L.T
    S2[bad - 1, bad - 1] -= 1.0
    rc, _, _, _ = potrf(ff, S2)
    assert rc == bad, (rc, bad)


def _device_spd(n, seed):
    """a well-conditioned SPD matrix built on the device (R R^T / 64 + 2 I: every pivot is >= 2)"""
    g = torch.Generator(device="cuda:0").manual_seed(seed)
    R = torch.randn(n, 64, dtype=torch.float64, device="cuda:0", generator=g)
    S = R @ R.T / 64.0
    S.diagonal().add_(2.0)
    return S


def _potrf_inplace(ff, S):
    """ffgp_potrf_rows on a copy of S (default options: look-ahead with value hand-offs from 1024 rows on) -> (rc, factor)"""
    _lib, h = ff
    n = S.shape[0]
    W = S.clone()
    rc = _lib.lib.ffgp_potrf_rows(h, ptr(W), n, n, n)
    torch.cuda.synchronize()
    return rc, W


@pytest.mark.noisy
@pytest.mark.parametrize("n", [1536, 4700, 9000])
def test_potrf_reports_first_bad_pivot_on_the_lookahead_path(ff, n):
    """the reference's `torch.linalg.cholesky` raises on the first non-positive pivot (GaussianProcess/cigp_v10.py:61); above la_min_n = 1024
    rows the factorisation is the look-ahead form with value hand-offs between its streams -- a failing pivot in the FIRST, a MIDDLE and
    the LAST outer panel must come back as its 1-based index, the call must return (no gate left waiting), and the handle must factor the
    next matrix correctly.  The pivot is made to fail by the UPDATED value (diagonal entry lowered by the pivot's own square + 1/2), so the
    trailing updates that feed it are part of what is tested"""
    _lib, h = ff
    S = _device_spd(n, n)
    rc, L = _potrf_inplace(ff, S)
    assert rc == 0
    piv = L.diagonal().clone()
    npan = (n + 511) // 512
    cases = sorted({7, 300, 512 * (npan // 2) + 129, 512 * (npan // 2) + 512, 512 * (npan - 1) + 1, n - 1, n})
    for bad in cases:
        S2 = S.clone()
        S2[bad - 1, bad - 1] -= float(piv[bad - 1]) ** 2 + 0.5
        rc, _ = _potrf_inplace(ff, S2)
        assert rc == bad, (n, bad, rc)
        rc, L2 = _potrf_inplace(ff, S)            # the handle is as good as new: same factor, bit for bit
        assert rc == 0 and torch.equal(L2.tril(), L.tril()), (n, bad)


def test_a_lost_handoff_is_an_error_within_a_second_not_a_hang(ff):
    """the look-ahead's streams hand over through words of device memory that a one-wave gate kernel polls (potrf.hip, ffgp_handoff_gate).
    A publication that never happens -- forced here by the test hook `ho_withhold`; in the field: a tool that serialises this process's
    kernels and was not recognised when the handle was created -- must end in FFGP_ERR_HANDOFF after `ho_timeout_ms`, never in a hung
    queue; the handle then factors the next matrix normally"""
    import time
    _lib, h = ff
    n = 3000
    S = _device_spd(n, 5)
    rc, L = _potrf_inplace(ff, S)
    assert rc == 0
    assert _lib.lib.ffgp_set_option(h, b"ho_timeout_ms", 150.0) == 0
    seen = set()
    try:
        for k in range(1, 7):
            assert _lib.lib.ffgp_set_option(h, b"ho_withhold", float(k)) == 0
            t0 = time.perf_counter()
            rc, L2 = _potrf_inplace(ff, S)
            dt = time.perf_counter() - t0
            _lib.lib.ffgp_set_option(h, b"ho_withhold", 0.0)
            assert rc in (0, -5) and dt < 1.0, (k, rc, dt)
            seen.add(rc)
            if rc == 0:      # (a publication nobody was waiting for)
                assert torch.equal(L2.tril(), L.tril()), k
            rc, L3 = _potrf_inplace(ff, S)
            assert rc == 0 and torch.equal(L3.tril(), L.tril()), k
    finally:
        _lib.lib.ffgp_set_option(h, b"ho_withhold", 0.0)
        _lib.lib.ffgp_set_option(h, b"ho_timeout_ms", 2000.0)
    assert -5 in seen, seen


def test_trsm_and_potrs(ff):
    import scipy.linalg as sla
    _lib, h = ff
    rng = np.random.default_rng(11)
    for n, nrhs in [(300, 1), (300, 5), (700, 130)]:
        S = spd(n, rng)
        rc, out, Wd, ld = potrf(ff, S)
        assert rc == 0
        L = np.linalg.cholesky(S)
        B = rng.standard_normal((n, nrhs))
        Bd = dev(B)
        assert _lib.lib.ffgp_trsm_lower(h, ptr(Wd), n, ld, ptr(Bd), nrhs, nrhs) == 0
        torch.cuda.synchronize()
        assert relerr(Bd.cpu().numpy(), sla.solve_triangular(L, B, lower=True)) < 1e-10
        Bd = dev(B)
        assert _lib.lib.ffgp_trsm_lower_t(h, ptr(Wd), n, ld, ptr(Bd), nrhs, nrhs) == 0
        torch.cuda.synchronize()
        assert relerr(Bd.cpu().numpy(), sla.solve_triangular(L.T, B, lower=False)) < 1e-10
        Bd = dev(B)
        assert _lib.lib.ffgp_potrs(h, ptr(Wd), n, ld, ptr(Bd), nrhs, nrhs) == 0
        torch.cuda.synchronize()
        assert relerr(Bd.cpu().numpy(), np.linalg.solve(S, B)) < 1e-9


@pytest.mark.parametrize("S,cases", [(256, [(300, 7), (640, 130), (1000, 301), (1537, 1), (257, 3)]),
                                     (512, [(1100, 33), (2049, 2)]), (1024, [(2500, 33), (3072, 1)])])
def test_trsm_super_block_sweeps(ff, S, cases):
    """the super-block sweeps (inverted S x S diagonal super-blocks, 2 n/S launches instead of 2 n/128): forward, transposed
    and potrs against scipy, ragged last super-block and odd right-hand-side counts included; the cached inverses follow
    the factor (a second factorisation in the same buffer must not reuse them)"""
    import scipy.linalg as sla
    _lib, h = ff
    rng = np.random.default_rng(S)
    assert _lib.lib.ffgp_set_option(h, b"super_block", float(S)) == 0
    assert _lib.lib.ffgp_set_option(h, b"super_min_n", 1.0) == 0
    try:
        for n, nrhs in cases:
            for rep in range(2):     # rep 1: same size, new matrix -> very likely the same device buffer
                A = spd(n, rng)
                rc, out, Wd, ld = potrf(ff, A)
                assert rc == 0
                L = np.linalg.cholesky(A)
                B = rng.standard_normal((n, nrhs))
                for fn, ref in ((_lib.lib.ffgp_trsm_lower, sla.solve_triangular(L, B, lower=True)),
                                (_lib.lib.ffgp_trsm_lower_t, sla.solve_triangular(L.T, B, lower=False)),
                                (_lib.lib.ffgp_potrs, sla.cho_solve((L, True), B))):
                    Bp = np.full((n, nrhs + 3), np.nan)      # padded leading dimension: nothing may land in the padding
                    Bp[:, :nrhs] = B
                    Bd = dev(Bp)
                    assert fn(h, ptr(Wd), n, ld, ptr(Bd), nrhs, nrhs + 3) == 0
                    torch.cuda.synchronize()
                    got = Bd.cpu().numpy()
                    assert np.isnan(got[:, nrhs:]).all()
                    assert relerr(got[:, :nrhs], ref) < 1e-9, (S, n, nrhs, fn.__name__)
    finally:
        _lib.lib.ffgp_set_option(h, b"super_block", 1024.0)
        _lib.lib.ffgp_set_option(h, b"super_min_n", 2048.0)


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6])
def test_solves_fuzz_default_paths(ff, seed):
    """random sizes around the switch points of the solve paths (128-block / super-block sweeps, MFMA tiles / split-K /
    matrix-vector kernels) with the library's default settings: trsm, trsm^T and potrs residuals against the factor itself
    (L X = B, L^T X = B, L L^T X = B evaluated in fp64 on the device)"""
    _lib, h = ff
    rng = np.random.default_rng(100 + seed)
    n = int(rng.choice([rng.integers(1500, 2048), rng.integers(2048, 2300), rng.integers(2300, 5200), 1024 * rng.integers(2, 5)]))
    nrhs = int(rng.choice([1, 2, rng.integers(3, 9), rng.integers(9, 70), rng.integers(200, 320)]))
    g = torch.Generator(device="cuda:0").manual_seed(seed)
    X = torch.rand((n, 5), generator=g, device="cuda:0", dtype=torch.float64)
    K = torch.exp(-0.5 * torch.cdist(X, X) ** 2) + 0.3 * torch.eye(n, device="cuda:0", dtype=torch.float64)
    ld = n + (n & 1) + 2
    W = torch.zeros((n, ld), device="cuda:0", dtype=torch.float64)
    W[:, :n] = torch.tril(K)
    assert _lib.lib.ffgp_potrf(h, ptr(W), n, ld) == 0
    L = torch.tril(W[:, :n])
    B = torch.randn((n, nrhs), generator=g, device="cuda:0", dtype=torch.float64)
    scale = float(B.abs().max())
    for fn, back in ((_lib.lib.ffgp_trsm_lower, lambda Z: L @ Z), (_lib.lib.ffgp_trsm_lower_t, lambda Z: L.T @ Z),
                     (_lib.lib.ffgp_potrs, lambda Z: L @ (L.T @ Z))):
        Z = B.clone()
        assert fn(h, ptr(W), n, ld, ptr(Z), nrhs, nrhs) == 0
        torch.cuda.synchronize()
        assert float((back(Z) - B).abs().max()) <= 1e-9 * scale * max(1.0, float(Z.abs().max())), (n, nrhs, fn.__name__)


@pytest.mark.parametrize("n", [100, 128, 300, 700, 1100])
def test_potri(ff, tile, n):
    _lib, h = ff
    rng = np.random.default_rng(n)
    S = spd(n, rng)
    rc, out, Wd, ld = potrf(ff, S)
    assert rc == 0
    assert _lib.lib.ffgp_potri(h, ptr(Wd), n, ld) == 0
    torch.cuda.synchronize()
    got = np.tril(Wd.cpu().numpy()[:n, :n])
    assert relerr(got, np.tril(np.linalg.inv(S))) < 1e-9


# ------------------------------------------------------------------------------------------------ assembly
@pytest.mark.parametrize("n1,n2,D", [(65, 33, 1), (64, 64, 5), (200, 131, 16), (130, 70, 37)])
def test_assemble_rect(ff, n1, n2, D):
    _lib, h = ff
    rng = np.random.default_rng(n1 + n2 + D)
    x1, x2 = rng.random((n1, D)) * 3, rng.random((n2, D)) * 3
    w = rng.random(D) + 0.3
    amp = np.array([1.7])
    K = torch.empty((n1, n2), dtype=torch.float64, device="cuda:0")
    x1d, x2d, wd, ad = dev(x1), dev(x2), dev(w), dev(amp)   # keep the device buffers alive across the launch
    rc = _lib.lib.ffgp_assemble(h, ptr(x1d), n1, ptr(x2d), n2, D, ptr(wd), ptr(ad), 1e-30, None, None, 0,
                                None, 0, 0.0, 0.0, ptr(K), n2, 0, 0, 1.0)
    assert rc == 0
    torch.cuda.synchronize()
    diff = (x1[:, None, :] - x2[None, :, :]) * w
    ref = 1.7 * np.exp(-0.5 * np.maximum((diff ** 2).sum(-1), 1e-30))
    assert relerr(K.cpu().numpy(), ref) < 1e-14


def test_assemble_sigma_extras(ff):
    _lib, h = ff
    rng = np.random.default_rng(3)
    n, D = 150, 4
    x = rng.random((n, D))
    w, amp, dadd = rng.random(D) + 0.5, np.array([0.9]), np.array([0.31])
    yv = rng.random((n, n))
    am = rng.random((n, n))
    xd, wd, ad, dd, yvd, amd = dev(x), dev(w), dev(amp), dev(dadd), dev(yv), dev(am)
    for lower in (0, 1):
        K = torch.full((n, n), 555.0, dtype=torch.float64, device="cuda:0")
        rc = _lib.lib.ffgp_assemble(h, ptr(xd), n, ptr(xd), n, D, ptr(wd), ptr(ad), float("-inf"), ptr(dd),
                                    ptr(yvd), n + 1, ptr(amd), n, 0.25, 1e-6, ptr(K), n, lower, 0, 1.0)
        assert rc == 0
        torch.cuda.synchronize()
        diff = (x[:, None, :] - x[None, :, :]) * w
        K0 = 0.9 * np.exp(-0.5 * (diff ** 2).sum(-1))
        low = np.tril(am) + np.tril(am, -1).T
        ref = K0 + np.diag(0.31 + np.diag(yv)) + low + 0.25 + 1e-6 * K0.mean() * np.eye(n)
        got = K.cpu().numpy()
        mask = np.tril(np.ones((n, n), dtype=bool))
        assert relerr(got[mask], ref[mask]) < 1e-13
        if lower:
            assert (got[~mask] == 555.0).all()
        else:
            assert relerr(got, ref) < 1e-13


def test_mfma_peak_probe(ff):
    _lib, h = ff
    tf = _lib.mfma_f64_peak(0)
    print("measured fp64 MFMA stream: %.1f TFLOP/s" % tf)
    assert 10.0 < tf < 200.0


@pytest.mark.timeout(300)
def test_c_abi_standalone_consumer(tmp_path):
    """examples/nlml_c_abi.cpp: a C++ program that links libffgp.so directly (no Python, no torch in the process),
    runs the fused NLML + gradients and checks one gradient against a finite difference of the value"""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc on this box")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "nlml_c_abi")
    libdir = os.path.join(root, "fidelityfusion_amd")
    subprocess.check_call([hipcc, "-O2", "--offload-arch=gfx950", "-I", os.path.join(root, "include"),
                           os.path.join(root, "examples", "nlml_c_abi.cpp"), "-L", libdir, "-lffgp",
                           "-Wl,-rpath," + libdir, "-o", exe])
    out = subprocess.run([exe, "1500", "6", "3"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "c-abi example ok" in out.stdout


def test_c_abi_rccl_consumer(tmp_path):
    """examples/joint_nll_rccl.cpp: the sharded joint likelihood without Python -- blocks dealt to the node's GPUs, fused
    NLML enqueued per GPU, ONE ffgp_allreduce_sum (RCCL, resolved by the library at run time) of the F-vector; every GPU
    must end with the block-by-block values.  On the 1-GPU box the communicator has one rank; the call path is the same."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc) or not os.path.exists("/opt/rocm/lib/librccl.so"):
        pytest.skip("no hipcc / RCCL development files on this box")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "joint_nll_rccl")
    libdir = os.path.join(root, "fidelityfusion_amd")
    subprocess.check_call([hipcc, "-O2", "--offload-arch=gfx950", "-I", os.path.join(root, "include"), "-I", "/opt/rocm/include",
                           os.path.join(root, "examples", "joint_nll_rccl.cpp"), "-L", libdir, "-lffgp", "-L", "/opt/rocm/lib", "-lrccl",
                           "-Wl,-rpath," + libdir, "-o", exe])
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([exe, "8", "6", "1024", "5", "3", "2"], capture_output=True, text=True, timeout=240, env=env)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "mismatches 0" in out.stdout


@pytest.mark.timeout(120)
@pytest.mark.parametrize("n", [1, 2, 5, 31, 33, 64])
def test_syevj_small_vs_lapack(n):
    """hand-written batched Jacobi eigensolver (csrc/eig.hip) against numpy.linalg.eigh: eigenvalues, orthogonality,
    reconstruction; kernel-like spectra (tiny trailing eigenvalues) and generic symmetric matrices; both orderings"""
    import torch
    from fidelityfusion_amd import functional as F
    rng = np.random.default_rng(n)
    mats = []
    for b in range(5):
        if b % 2 == 0:                      # kernel matrix: fast-decaying spectrum, numerically rank deficient
            X = rng.random((n, 2))
            sq = ((X[:, None, :] - X[None, :, :]) ** 2).sum(-1)
            mats.append(np.exp(-0.5 * sq / 0.5 ** 2))
        else:
            R = rng.standard_normal((n, n))
            mats.append(0.5 * (R + R.T))
    M = np.stack(mats)
    ev, Q = F._syevj_small(torch.tensor(M, device="cuda"))
    ev, Q = ev.cpu().numpy(), Q.cpu().numpy()
    for b in range(len(mats)):
        w = np.linalg.eigvalsh(M[b])
        scale = max(np.abs(w).max(), 1e-300)
        assert np.abs(ev[b] - w).max() <= 1e-13 * scale, (b, np.abs(ev[b] - w).max())
        assert np.abs(Q[b].T @ Q[b] - np.eye(n)).max() < 1e-13
        assert np.abs(Q[b] @ np.diag(ev[b]) @ Q[b].T - M[b]).max() <= 1e-13 * scale
    evd, Qd = F._syevj_small(torch.tensor(M, device="cuda"), descending=True)
    assert np.allclose(evd.cpu().numpy(), ev[:, ::-1], rtol=0, atol=1e-13 * np.abs(ev).max())


@pytest.mark.parametrize("shape,tau_kind", [((37, 5, 4), "scalar"), ((70, 6), "scalar"), ((20, 3, 4, 2), "tensor")])
def test_kron_nll_closed_form_backward(shape, tau_kind):
    """the Kronecker-structured likelihood of HOGP (hogp_simple.py:92-117): value and the closed-form gradients w.r.t. Y,
    tau and every K_m against the dense definition 1/2 log|S| + 1/2 y^T S^-1 y, S = kron(K_m) + tau I, differentiated
    by torch (scalar tau), and against torch autograd through `eigh` for an elementwise tau (which lives in the
    eigenbasis, as the reference's y_var does)"""
    import math
    import torch
    from fidelityfusion_amd.hogp_simple import kron_nll, multi_mode_dot, _outer
    rng = np.random.default_rng(11)
    Ks0 = []
    for dm in shape:
        R = rng.standard_normal((dm, dm))
        Ks0.append(R @ R.T / dm + 0.1 * np.eye(dm))
    Y0 = rng.standard_normal(shape)
    nd = int(np.prod(shape))
    tau0 = np.array([0.7]) if tau_kind == "scalar" else 0.5 + rng.random(shape)

    def leaves():
        return (torch.tensor(Y0, device="cuda", requires_grad=True), torch.tensor(tau0, device="cuda", requires_grad=True),
                [torch.tensor(K, device="cuda", requires_grad=True) for K in Ks0])

    y, tau, Ks = leaves()
    loss, cache = kron_nll(y, tau, Ks)
    loss.backward()
    y2, tau2, Ks2 = leaves()
    if tau_kind == "scalar":
        S = Ks2[0]
        for K in Ks2[1:]:
            S = torch.kron(S, K)
        S = S + tau2 * torch.eye(nd, device="cuda", dtype=torch.float64)
        v = y2.reshape(-1, 1)
        ref = (0.5 * nd * math.log(2 * math.pi) + 0.5 * torch.logdet(S) + 0.5 * (v.T @ torch.linalg.solve(S, v)).sum()) / nd
        g_ref = torch.linalg.solve(S.detach(), v.detach()).reshape(shape)
        assert np.abs((cache["g"] - g_ref).cpu().numpy()).max() <= 1e-10 * float(g_ref.abs().max())
    else:
        es = [torch.linalg.eigh(K, UPLO="U") for K in Ks2]
        A = _outer([e[0] for e in es]) + tau2
        T1 = multi_mode_dot(y2, [e[1].T.contiguous() for e in es])
        ref = (0.5 * nd * math.log(2 * math.pi) + 0.5 * torch.log(A).sum() + 0.5 * (T1 * T1 / A).sum()) / nd
    ref.backward()
    assert abs(float(loss.detach()) - float(ref.detach())) <= 1e-12 * abs(float(ref.detach()))
    sym = lambda G: 0.5 * (G + G.T)
    pairs = [(y.grad, y2.grad), (tau.grad, tau2.grad)] + [(sym(a.grad), sym(b.grad)) for a, b in zip(Ks, Ks2)]
    for a, b in pairs:
        assert a.shape == b.shape
        assert np.abs((a - b).cpu().numpy()).max() <= 1e-9 * max(float(b.abs().max()), 1e-300)


def test_eigh_small_backward_matches_torch():
    """the autograd wrapper of the LDS Jacobi solver: gradients of a scalar function of (eigenvalues, eigenvectors) --
    chosen invariant to the eigenvectors' signs -- against torch.linalg.eigh's"""
    import torch
    from fidelityfusion_amd import functional as F
    rng = np.random.default_rng(4)
    R = rng.standard_normal((23, 23))
    K0 = R @ R.T / 23 + np.diag(rng.random(23))
    W = torch.tensor(rng.standard_normal((23, 23)), device="cuda")
    outs = []
    for fn in (F.eigh_small, lambda K: torch.linalg.eigh(K, UPLO="U")):
        K = torch.tensor(K0, device="cuda", requires_grad=True)
        lam, U = fn(K)
        loss = (lam ** 2 * torch.arange(1, 24, device="cuda")).sum() + ((U * lam.sqrt()) @ (U * lam.sqrt()).T * W).sum() \
            + (U @ torch.diag(1.0 / (1.0 + lam)) @ U.T * W.T).sum()
        loss.backward()
        outs.append((float(loss.detach()), K.grad.cpu().numpy()))
    assert abs(outs[0][0] - outs[1][0]) <= 1e-11 * abs(outs[1][0])
    g0, g1 = outs[0][1], outs[1][1]
    assert np.abs(g0 - 0.5 * (g1 + g1.T)).max() <= 1e-9 * np.abs(g1).max()


def _asm_call(_lib, h, x1d, n1, x2d, n2, D, wd, ad, clamp, dd, dvd, add_all, mj, lower, opts):
    for k, v in opts.items():
        assert _lib.lib.ffgp_set_option(h, k.encode(), float(v)) == 0
    K = torch.full((n1, n2), 555.0, dtype=torch.float64, device="cuda:0")
    try:
        rc = _lib.lib.ffgp_assemble(h, ptr(x1d), n1, ptr(x2d), n2, D, ptr(wd), ptr(ad), clamp, ptr(dd) if dd is not None else None,
                                    ptr(dvd) if dvd is not None else None, 1, None, 0, add_all, mj, ptr(K), n2, lower, 0, 1.0)
        assert rc == 0
        torch.cuda.synchronize()
    finally:
        _lib.lib.ffgp_set_option(h, b"asm_mm", 1.0)
        _lib.lib.ffgp_set_option(h, b"asm_mm_min", 6144.0)
        _lib.lib.ffgp_set_option(h, b"asm_mm_grid", 768.0)
    return K


@pytest.mark.gpu
@pytest.mark.parametrize("n1,n2,D,lower,extras", [(640, 640, 16, 1, True), (640, 640, 16, 0, False), (709, 709, 5, 1, True), (710, 710, 5, 1, True), (1210, 840, 9, 0, False),
                                                   (1000, 1000, 20, 0, True), (448, 448, 37, 1, False), (384, 333, 16, 0, False),
                                                   (517, 1100, 3, 0, False), (2048, 2048, 128, 1, False)])
def test_assemble_matrix_core_path(ff, n1, n2, D, lower, extras):
    """interior squared-exponential tiles through the MFMA chain (norm expansion, operands in lane layout, persistent waves,
    deferred stores) against the difference kernel: the diagonal tiles and every Sigma extra must agree exactly, interior entries
    to the expansion's error eps * (|x_i|^2 + |x_j|^2); lower / full / rectangular sweeps, ragged edges, one to eight k-chunks,
    the mean(K) jitter's tile sums, several persistent grid sizes"""
    _lib, h = ff
    rng = np.random.default_rng(n1 * 3 + n2 + D)
    x1 = rng.random((n1, D)) * 2 + 5.0                      # (an offset: the kernel shifts by the first point)
    sym = (n1 == n2)
    x2 = x1 if sym else rng.random((n2, D)) * 2 + 5.0
    w = (rng.random(D) + 0.3) / np.sqrt(D / 4.0)
    x1d, wd, ad = dev(x1), dev(w), dev(np.array([1.3]))
    x2d = x1d if sym else dev(x2)
    dd = dev(np.array([0.37])) if extras else None
    dvd = dev(rng.random(n1)) if extras else None
    add_all, mj = (0.125, 1e-3) if extras else (0.0, 0.0)
    clamp = 1e-30 if D != 5 else float("-inf")
    ref = _asm_call(_lib, h, x1d, n1, x2d, n2, D, wd, ad, clamp, dd, dvd, add_all, mj, lower, {"asm_mm": 0})
    for grid in (768, 5, 100000):
        got = _asm_call(_lib, h, x1d, n1, x2d, n2, D, wd, ad, clamp, dd, dvd, add_all, mj, lower, {"asm_mm_min": 128, "asm_mm_grid": grid})
        # scaled inputs are O(1) after the shift: the expansion's error in the distance is a few eps * |x|^2 ~ 1e-14
        assert float((got - ref).abs().max()) < 3e-14 * 1.3, grid
        if lower:
            assert bool((got[torch.ones_like(got, dtype=torch.bool).triu(1)] == 555.0).all())
        if sym:
            for t in range(0, n1, 64):       # diagonal tiles: difference form in both launches -- the same bits but for the jitter's sum
                blk = (slice(t, t + 64), slice(t, t + 64))
                tol = 0.0 if mj == 0.0 else 1e-15
                assert float((got[blk] - ref[blk]).abs().max()) <= tol
    # the matrix-core launch really ran (and is not the same arithmetic): some interior entry differs in the last bits --
    # except with an odd leading dimension, where its 16-byte stores are not possible and the difference kernel runs alone
    assert torch.equal(got, ref) == (n2 % 2 == 1)


@pytest.mark.gpu
def test_assemble_matrix_core_near_coincident_points(ff):
    """points that nearly coincide with another far-away row: the expansion would lose their distance (1e-18 against |x|^2 ~ 1);
    the wave flags the tile and the second launch recomputes it in the difference form -- bit-identical to the difference kernel --
    while unflagged tiles keep the matrix-core result"""
    _lib, h = ff
    rng = np.random.default_rng(77)
    n, D = 1024, 8
    x = rng.random((n, D)) * 2
    x[900:910] = x[100:110] + 1e-9 * rng.standard_normal((10, D))       # tile (14, 1)
    x[700] = x[3]                                                       # exact duplicate: tile (10, 0)
    xd, wd, ad = dev(x), dev(np.ones(D)), dev(np.array([1.0]))
    ref = _asm_call(_lib, h, xd, n, xd, n, D, wd, ad, 1e-30, None, None, 0.0, 0.0, 1, {"asm_mm": 0})
    got = _asm_call(_lib, h, xd, n, xd, n, D, wd, ad, 1e-30, None, None, 0.0, 0.0, 1, {"asm_mm_min": 128})
    for (ti, tj) in ((14, 1), (10, 0)):
        blk = (slice(64 * ti, 64 * ti + 64), slice(64 * tj, 64 * tj + 64))
        assert torch.equal(got[blk], ref[blk]), (ti, tj)
    assert float(got[700, 3]) == 1.0 and abs(float(got[905, 105]) - 1.0) < 1e-16
    assert float((got - ref).abs().max()) < 1e-14 and not torch.equal(got, ref)


@pytest.mark.gpu
def test_assembly_exp_range(ff):
    """the assembly's own exp (range reduction + degree-13 polynomial + ldexp) against torch over the whole argument range:
    ~1 ulp where the result is normal, exactly 0 where exp() underflows, finite in the subnormal range"""
    import torch
    from fidelityfusion_amd import functional as F
    dev = "cuda:0"
    # squared distances 0 .. 1600 between points on a line: K = exp(-s / 2) spans 1 .. e^-800
    x = torch.linspace(0.0, 40.0, 4001, dtype=torch.float64, device=dev).reshape(-1, 1)
    one = torch.ones(1, dtype=torch.float64, device=dev)
    K = F.kernel_matrix(x[:1], x, one, one, clamp=1e-30)[0]
    ref = torch.exp(-0.5 * torch.clamp((x[:, 0] - x[0, 0]) ** 2, min=1e-30))
    normal = ref > 1e-300
    assert float(((K - ref).abs() / ref)[normal].max()) < 4e-16
    assert bool((K[ref == 0.0] == 0.0).all()) and int((ref == 0.0).sum()) > 100
    sub = (~normal) & (ref > 0)
    assert bool(torch.isfinite(K[sub]).all()) and float((K[sub] - ref[sub]).abs().max()) < 1e-300
