"""Symmetric eigendecomposition on the library's own kernels: `eigh(K)` is the drop-in for the `torch.linalg.eigh(K_x)` of the HOGP
block (FidelityFusion_Models/two_fidelity_models/hogp_simple.py:15-19,97-100; MFGP_ver2023May/base_gp/hogp.py:20-24).

`eigh` -> `ffgp_syevd` (csrc/sy2sb.hip, sb2st.hip, stedc.hip, syevd.hip): dense -> band (TSQR + Householder reconstruction panels,
rank-64 updates on the fp64 matrix cores) -> tridiagonal (bulge chasing) -> divide & conquer -> two back-transformations.  The
stage wrappers (`sy2sb`, `sb2st`, `stedc`, `ormq2`, `ormq1`) expose the pieces for tests and profiling.

`jacobi_eigh` (round 2) stays as the slow, independent hand-written cross-check and as the route for n > 32768: two-sided block
Jacobi, 32-wide blocks paired round-robin, every 64 x 64 pair problem on the LDS Jacobi kernel (ffgp_syevj_small), rotations applied
by batched MFMA GEMMs; eigenvectors of the pair solver re-ordered by centre of mass so that every rotation stays close to the
identity.  5-9 sweeps on generic matrices, 10-19 on kernel matrices; N = 2048 1.2 s, 4096 4.2 s, 8192 21 s.
"""
import ctypes as C

import torch

from . import _lib
from ._lib import check, lib

_ptr = lambda t: C.c_void_p(t.data_ptr())


def _h(dev):
    h = _lib.handle(dev.index)
    _lib.bind_stream(h, dev.index)
    return h


def _gemm(dev, opa, opb, A, B, m, n, k, alpha=1.0, out=None, beta=0.0, lower=0):
    """C[m, n] = alpha op(A) op(B) + beta C on ffgp_gemm (opa = 0: A stored m x k; 1: k x m.  opb = 0: B stored n x k; 1: k x n)"""
    if out is None:
        out = torch.empty((m, n), dtype=torch.float64, device=dev)
    if m and n:
        check(lib.ffgp_gemm(_h(dev), opa, opb, lower, 0, _ptr(A), A.stride(0), _ptr(B), B.stride(0), _ptr(out), out.stride(0), m, n, k,
                            float(alpha), float(beta)), "ffgp_gemm")
    return out


# ----------------------------------------------------------------------------------------------------------------------
# two-stage solver (ffgp_syevd) and its stages
# ----------------------------------------------------------------------------------------------------------------------
SYEVD_MAX_N = 32768


def eigh(K):
    """(eigenvalues ascending [n], eigenvectors in columns [n, n]) of the symmetric matrix K (lower triangle read), fp64, on the
    device; K is not modified.  n <= 32768 runs the two-stage solver, larger matrices the block Jacobi."""
    if K.dim() != 2 or K.shape[0] != K.shape[1]:
        raise ValueError("eigh expects a square matrix, got %s" % (tuple(K.shape),))
    dev = K.device
    n = K.shape[0]
    A = K.detach().to(torch.float64)
    if A.stride(1) != 1 or A.stride(0) < n:
        A = A.contiguous()
    if n > SYEVD_MAX_N:
        return jacobi_eigh(A)
    W = torch.empty(n, dtype=torch.float64, device=dev)
    Z = torch.empty((n, n), dtype=torch.float64, device=dev)
    if n:
        check(lib.ffgp_syevd(_h(dev), _ptr(A), n, A.stride(0), _ptr(W), _ptr(Z), n), "ffgp_syevd")
    return W, Z


def sy2sb(A):
    """stage 1 on a copy of A [n, n] (n a multiple of 64): (AB [n, 64] band storage, Y [n, n] panel reflectors)"""
    dev, n = A.device, A.shape[0]
    Aw = A.detach().to(torch.float64).clone().contiguous()
    AB = torch.empty((n, 64), dtype=torch.float64, device=dev)
    Y = torch.empty((n, n), dtype=torch.float64, device=dev)
    check(lib.ffgp_sy2sb(_h(dev), _ptr(Aw), n, n, _ptr(AB), _ptr(Y), n), "ffgp_sy2sb")
    return AB, Y


def sb2st(AB):
    """stage 2 on a copy of the band: (d [n], e [n], reflector store)"""
    dev, n = AB.device, AB.shape[0]
    ABw = AB.clone().contiguous()
    d = torch.empty(n, dtype=torch.float64, device=dev)
    e = torch.empty(n, dtype=torch.float64, device=dev)
    refl = torch.empty(int(lib.ffgp_sb2st_reflector_doubles(n)), dtype=torch.float64, device=dev)
    check(lib.ffgp_sb2st(_h(dev), _ptr(ABw), n, _ptr(d), _ptr(e), _ptr(refl)), "ffgp_sb2st")
    return d, e, refl


def stedc(d, e):
    """stage 3: eigenpairs of the symmetric tridiagonal matrix (d, e[:n-1])"""
    dev, n = d.device, d.shape[0]
    W = torch.empty(n, dtype=torch.float64, device=dev)
    Z = torch.empty((n, n), dtype=torch.float64, device=dev)
    check(lib.ffgp_stedc(_h(dev), _ptr(d.contiguous()), _ptr(e.contiguous()), n, _ptr(W), _ptr(Z), n), "ffgp_stedc")
    return W, Z


def ormq2(refl, Z):
    """Z <- Q2 Z in place (the chase's reflectors)"""
    n = Z.shape[0]
    check(lib.ffgp_ormq2(_h(Z.device), _ptr(refl), n, _ptr(Z), Z.stride(0), Z.shape[1]), "ffgp_ormq2")
    return Z


def ormq1(Y, Z):
    """Z <- Q1 Z in place (the panels' reflectors)"""
    n = Z.shape[0]
    check(lib.ffgp_ormq1(_h(Z.device), _ptr(Y), Y.stride(0), n, _ptr(Z), Z.stride(0), Z.shape[1]), "ffgp_ormq1")
    return Z


def _syevj_small(M):
    from .functional import _syevj_small as f
    return f(M)


# ----------------------------------------------------------------------------------------------------------------------
# dense core: two-sided block Jacobi
# ----------------------------------------------------------------------------------------------------------------------
def _round_robin(nb):
    """the nb - 1 rounds of a round-robin tournament on nb (even) players: every round pairs all players, every pair
    meets exactly once per cycle"""
    players = list(range(nb))
    rounds = []
    for _ in range(nb - 1):
        rounds.append([(players[i], players[nb - 1 - i]) for i in range(nb // 2)])
        players = [players[0]] + [players[-1]] + players[1:-1]
    return rounds


def _rotate_rows(dev, M, idx, Jt_src, scratch):
    """M[rows of every pair, :] <- J^T M[rows, :]   for all (disjoint) pairs at once.
    idx [npairs * 64] row indices, Jt_src [npairs, 64, 64] with J's columns = eigenvectors."""
    npairs, m = idx.numel() // 64, M.shape[1]
    X, Y = scratch
    torch.index_select(M, 0, idx, out=X)                      # gather: [npairs * 64, m]
    check(lib.ffgp_gemm_batched(_h(dev), 1, 1, 0, _ptr(Jt_src), 64, 64 * 64, _ptr(X), m, 64 * m, _ptr(Y), m, 64 * m, 64, m, 64, 1.0, 0.0,
                                npairs), "ffgp_gemm_batched")
    M.index_copy_(0, idx, Y)


def jacobi_eigh(B, max_sweeps=30, tol=2e-15):
    """(evals ascending [n], evecs [n, n]) of a symmetric B by two-sided block Jacobi on the library's kernels.
    B is not modified."""
    dev = B.device
    n = B.shape[0]
    if n <= 64:
        ev, Q = _syevj_small(B.contiguous()[None])
        return ev[0], Q[0]
    nb = -(-n // 32)
    nb += nb & 1
    m = nb * 32
    A = torch.zeros((m, m), dtype=torch.float64, device=dev)
    A[:n, :n] = B
    scale = float(B.abs().sum(1).max()) or 1.0      # Gershgorin: every eigenvalue of B lies in [-scale, scale]
    if m > n:   # padding: decoupled 1 x 1 blocks with distinct values below the whole spectrum -- they never rotate
        A[n:, n:] = torch.diag(-scale * (2.0 + torch.arange(m - n, dtype=torch.float64, device=dev)))
    Vt = torch.eye(m, dtype=torch.float64, device=dev)                 # V^T: its rows are rotated like A's
    rounds = _round_robin(nb)
    ar = torch.arange(32, device=dev)
    idx_rounds = []
    for pairs in rounds:
        p = torch.tensor([a for a, _ in pairs], device=dev)
        q = torch.tensor([b for _, b in pairs], device=dev)
        idx = torch.cat([(p * 32).unsqueeze(1) + ar, (q * 32).unsqueeze(1) + ar], 1).reshape(-1)      # [npairs * 64]
        idx_rounds.append(idx)
    npairs = nb // 2
    scratch = (torch.empty((npairs * 64, m), dtype=torch.float64, device=dev), torch.empty((npairs * 64, m), dtype=torch.float64, device=dev))
    fro = float(torch.linalg.matrix_norm(A[:n, :n])) or 1.0
    ramp = torch.arange(64, dtype=torch.float64, device=dev).reshape(1, 64, 1)
    prev = float("inf")
    for sweep in range(max_sweeps):
        for idx in idx_rounds:
            blocks = A.index_select(0, idx).reshape(npairs, 64, m).gather(2, idx.reshape(npairs, 1, 64).expand(npairs, 64, 64))
            blocks = 0.5 * (blocks + blocks.transpose(1, 2))
            _, J = _syevj_small(blocks.contiguous())
            # the pair solver returns its eigenvectors sorted by eigenvalue; a block-Jacobi rotation must instead stay close
            # to the identity (sorted columns keep permuting converged diagonal entries between the two blocks and the
            # iteration stalls): order the columns by where their mass sits
            pos = ((J * J) * ramp).sum(1)                                   # [npairs, 64] centre of mass of every eigenvector
            J = J.gather(2, torch.argsort(pos, dim=1).unsqueeze(1).expand(-1, 64, -1)).contiguous()
            _rotate_rows(dev, A, idx, J, scratch)                     # A <- J^T A
            A = A.T.contiguous()
            _rotate_rows(dev, A, idx, J, scratch)                     # A <- J^T (J^T A)^T = J^T A J   (symmetric again)
            _rotate_rows(dev, Vt, idx, J, scratch)                    # V^T <- J^T V^T
        off = float(torch.linalg.matrix_norm(A - torch.diag(A.diagonal())))
        if off <= tol * fro or (off <= 1e-12 * fro and off > 0.25 * prev):   # converged, or at the rounding floor
            break
        prev = off
    ev = A.diagonal()[:m].clone()
    keep = torch.ones(m, dtype=torch.bool, device=dev)
    if m > n:   # drop the padding's eigenpairs: the ones whose vectors live on the padded coordinates
        mass = Vt[:, n:].abs().amax(1)
        keep = mass < 0.5
        if int(keep.sum()) != n:   # should not happen (the padding sits outside B's Gershgorin interval): keep the n rows with least padding mass
            keep = torch.zeros(m, dtype=torch.bool, device=dev)
            keep[torch.argsort(mass)[:n]] = True
    ev, V = ev[keep], Vt[keep][:, :n].T
    order = torch.argsort(ev)
    return ev[order].contiguous(), V[:, order].contiguous()


