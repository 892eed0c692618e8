"""CPU: the numpy restatement of the device's two-stage symmetric eigensolver (oracle/eigh_twostage.py) against LAPACK -- the routine
the reference reaches through torch.linalg.eigh (two_fidelity_models/hogp_simple.py:15-19) -- stage by stage and end to end, incl. the
orderings the device kernels rely on (grouped WY blocks of the chase's reflectors, forward and backward; two- and three-level TSQR
with Householder reconstruction on rank-deficient panels)."""
import numpy as np
import pytest

from oracle import eigh_twostage as E


def _kern(rng, n, D, ls):
    X = rng.random((n, D))
    return np.exp(-0.5 * ((X[:, None, :] - X[None, :, :]) ** 2).sum(-1) / ls ** 2)


def test_stages_and_block_orderings():
    rng = np.random.default_rng(0)
    n, b = 192, 16
    M = rng.standard_normal((n, n))
    A = M + M.T
    ref = np.linalg.eigvalsh(A)
    Bd, panels = E.sy2sb(A.copy(), b, leaf=64)
    assert np.abs(np.tril(Bd, -b - 1)).max() == 0.0
    assert np.abs(np.linalg.eigvalsh(Bd) - ref).max() < 1e-12 * np.abs(ref).max()
    for agg in (1, 3):
        Q1 = E.apply_q1(panels, np.eye(n), agg=agg)
        assert np.abs(Q1.T @ Q1 - np.eye(n)).max() < 1e-13 and np.abs(Q1 @ Bd @ Q1.T - A).max() < 1e-12
    d, e, refl = E.sb2st(Bd, b)
    T = np.diag(d) + np.diag(e, 1) + np.diag(e, -1)
    assert np.abs(np.linalg.eigvalsh(T) - ref).max() < 1e-12 * np.abs(ref).max()
    Q2 = E.apply_q2(refl, n, b, np.eye(n), None)
    assert np.abs(Q2 @ T @ Q2.T - Bd).max() < 1e-12
    X = rng.standard_normal((n, 5))
    for g in (4, 16):
        assert np.abs(E.apply_q2(refl, n, b, np.eye(n), g) - Q2).max() < 1e-14        # (group descending, step ascending)
        assert np.abs(E.apply_q2_wavefront(refl, n, b, np.eye(n), g) - Q2).max() < 1e-14   # (four groups per pass, two steps apart)
        assert np.abs(E.apply_q2_t(refl, n, b, X, g) - Q2.T @ X).max() < 1e-13         # (group ascending, step descending, T^T)
    lam, Z = E.stedc(d, e, leaf=16)
    assert np.abs(lam - np.linalg.eigvalsh(T)).max() < 1e-12 * np.abs(ref).max()
    assert np.abs(Z.T @ Z - np.eye(n)).max() < 1e-13


@pytest.mark.parametrize("m,b,leaf,fan", [(1000, 16, 32, 4), (700, 16, 32, 3), (520, 32, 64, 16)])
def test_tsqr_householder_reconstruction(m, b, leaf, fan):
    rng = np.random.default_rng(m)
    P = rng.standard_normal((m, b))
    P[:, 3] *= 1e-14                      # a numerically dependent column: what kernel-matrix panels look like
    P[:, 7] = P[:, 2] + 1e-15 * rng.standard_normal(m)
    Y, T, R = E.tsqr_hr3(P, leaf, fan)
    H = np.eye(m) - Y @ T @ Y.T
    Rf = np.zeros((m, b))
    Rf[:b] = R
    assert np.abs(H.T @ H - np.eye(m)).max() < 1e-13
    assert np.abs(H @ Rf - P).max() < 1e-13 * np.abs(P).max()
    assert np.allclose(np.triu(Y[:b], 1), 0) and np.allclose(np.diag(Y[:b]), 1)


@pytest.mark.parametrize("kind", ["random", "kernel_lowrank", "clusters", "laplacian"])
def test_end_to_end_vs_lapack(kind):
    rng = np.random.default_rng(5)
    if kind == "random":
        M = rng.standard_normal((130, 130))
        A = M + M.T
    elif kind == "kernel_lowrank":
        A = _kern(rng, 300, 1, 0.5)
    elif kind == "clusters":
        Q, _ = np.linalg.qr(rng.standard_normal((200, 200)))
        A = Q @ np.diag(np.r_[np.ones(120), 2 * np.ones(80)]) @ Q.T
    else:
        A = np.diag(2.0 * np.ones(257)) + np.diag(-np.ones(256), 1) + np.diag(-np.ones(256), -1)
    n = A.shape[0]
    lam, Z = E.eigh_twostage(A, b=16, leaf=64, g=16, agg=4)
    ref = np.linalg.eigvalsh(A)
    assert np.abs(lam - ref).max() < 1e-13 * np.abs(ref).max()
    assert np.abs(Z.T @ Z - np.eye(n)).max() < 1e-13
    assert np.linalg.norm((Z * lam) @ Z.T - A) < 1e-13 * np.linalg.norm(A) * 4
