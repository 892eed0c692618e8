"""TEST INFRASTRUCTURE -- a CPU (numpy) restatement of the BUILD'S OWN two-stage symmetric eigensolver (csrc/syevd_*.hip), stage by
stage, so that every device stage has something to be compared with at small sizes.  Only tests/ may import it.

The reference itself calls `torch.linalg.eigh` (FidelityFusion_Models/two_fidelity_models/hogp_simple.py:15-19,97-100;
MFGP_ver2023May/base_gp/hogp.py:20-24) -- i.e. LAPACK's syevd, a third-party routine; the end-to-end oracle for the device
solver is therefore LAPACK (numpy.linalg.eigh), and this file only restates the *intermediate* quantities of the route the
device takes, in the same order:

  stage 1  sy2sb    dense -> band (bandwidth b): per panel a TSQR (leaf Householder QRs + one QR of the stacked R factors),
                    Householder reconstruction of the panel's compact-WY form (Ballard, Demmel, Grigori, Jacquelin, Knight,
                    Nguyen 2015: modified LU of  [I;0] - Q1 S), two-sided update  A <- A - V W^T - W V^T
  stage 2  sb2st    band -> tridiagonal by bulge chasing (one reflector of length <= b per (sweep, step)), reflectors kept
  stage 3  stedc    tridiagonal divide & conquer (Cuppen; deflation as LAPACK dlaed2, secular equation solved per root in the
                    shifted variable, Gu-Eisenstat recomputation of z), merges as products with a dense "S" matrix that folds
                    the sorting permutation, the deflation rotations and the secular eigenvectors
  stage 4  back-transformation  Z <- Q1 (Q2 Z): stage-2 reflectors grouped g sweeps at a time into WY blocks applied in the
                    order (sweep group descending, step ascending); stage-1 panels aggregated into wide WY blocks
"""
import numpy as np


# ----------------------------------------------------------------------------------------------------------------------
# Householder helpers
# ----------------------------------------------------------------------------------------------------------------------
def house(x):
    """(v, tau, beta) with v[0] = 1, (I - tau v v^T) x = beta e_1  (LAPACK dlarfg convention; x = 0 or already e_1-like: tau = 0)"""
    x = np.asarray(x, dtype=np.float64)
    v = np.zeros_like(x)
    v[0] = 1.0
    if x.size <= 1:
        return v, 0.0, (x[0] if x.size else 0.0)
    sigma = float(np.dot(x[1:], x[1:]))
    if sigma == 0.0:
        return v, 0.0, float(x[0])
    alpha = float(x[0])
    nrm = np.sqrt(alpha * alpha + sigma)
    beta = -nrm if alpha >= 0 else nrm
    tau = (beta - alpha) / beta
    v[1:] = x[1:] / (alpha - beta)
    return v, tau, beta


def householder_qr_vt(A):
    """unblocked Householder QR of A [m, b] (m >= 1): returns (V unit lower trapezoidal [m, b], T upper [b, b], R upper [b, b]) with
    A = (I - V T V^T) [R; 0].  For m < b the trailing reflectors are identity (tau = 0)."""
    A = np.array(A, dtype=np.float64)
    m, b = A.shape
    V = np.zeros((m, b))
    T = np.zeros((b, b))
    for j in range(b):
        if j >= m:
            continue
        v, tau, beta = house(A[j:, j])
        V[j:, j] = v
        A[j, j] = beta
        A[j + 1:, j] = 0.0
        if tau != 0.0 and j + 1 < b:
            w = v @ A[j:, j + 1:]
            A[j:, j + 1:] -= tau * np.outer(v, w)
        # T[:j, j] = -tau T[:j,:j] (V[:, :j]^T v_j)
        if j > 0:
            T[:j, j] = -tau * (T[:j, :j] @ (V[j:, :j].T @ v))
        T[j, j] = tau
    R = np.zeros((b, b))
    mm = min(m, b)
    R[:mm, :] = np.triu(A[:mm, :])
    return V, T, R


# ----------------------------------------------------------------------------------------------------------------------
# stage 1: panel factorisation = TSQR + Householder reconstruction
# ----------------------------------------------------------------------------------------------------------------------
def tsqr_hr(P, leaf=512):
    """Compact-WY Householder form of the QR factorisation of the panel P [m, b], obtained the way the device does it:
    leaf QRs of `leaf` rows, one QR of the stacked R factors, explicit thin Q1 only implicitly (through b x b blocks), modified
    LU of [I;0] - Q1 S.  Returns (Y [m, b] unit lower trapezoidal, T [b, b] upper, R [b, b] upper with P = (I - Y T Y^T)[R; 0])."""
    P = np.asarray(P, dtype=np.float64)
    m, b = P.shape
    assert m >= b, "the padded problem only produces panels with at least b rows"
    starts = list(range(0, m, leaf))
    if len(starts) > 1 and m - starts[-1] < b:   # a last leaf shorter than b rows joins its neighbour
        starts.pop()
    bounds = starts + [m]
    leaves = []
    Rst = []
    for i in range(len(starts)):
        Vi, Ti, Ri = householder_qr_vt(P[bounds[i]:bounds[i + 1]])
        leaves.append((Vi, Ti))
        Rst.append(Ri)
    L = len(leaves)
    if L == 1:                                               # a single leaf: Q_top = I
        R = Rst[0]
        Qtop = [np.eye(b)]
    else:
        Vt, Tt, R = householder_qr_vt(np.vstack(Rst))      # [L*b, b]
        Xt = Tt @ Vt[:b].T                                   # thin Q_top = [I;0] - Vt Xt
        Qtop = [-(Vt[i * b:(i + 1) * b] @ Xt) for i in range(L)]
        Qtop[0] = Qtop[0] + np.eye(b)
    X = [Ti @ Vi[:b].T for (Vi, Ti) in leaves]              # thin Q_i = [I;0] - V_i X_i
    Wtop = (np.eye(b) - leaves[0][0][:b] @ X[0]) @ Qtop[0]   # top b x b block of Q1
    # modified LU of M = [I;0] - Q1 S, the signs S chosen on the fly so that every pivot is >= 1 in magnitude
    Wt = Wtop.copy()
    S = np.zeros(b)
    Y1 = np.eye(b)
    U = np.zeros((b, b))
    for j in range(b):
        S[j] = -1.0 if Wt[j, j] >= 0 else 1.0
        piv = 1.0 - S[j] * Wt[j, j]
        lcol = -S[j] * Wt[j + 1:, j] / piv
        Y1[j + 1:, j] = lcol
        Wt[j + 1:, j + 1:] -= np.outer(lcol, Wt[j, j + 1:])
    for j in range(b):
        U[j, j:] = -S[j:] * Wt[j, j:]
        U[j, j] += 1.0
    T = U @ np.linalg.inv(Y1).T                              # T = U Y1^-T   (upper triangular)
    Uinv = np.linalg.inv(U)
    Y = np.zeros((m, b))
    for i in range(L):
        Gi = -(Qtop[i] * S[None, :]) @ Uinv                  # rows of leaf i:  Y = Q_i G_i = [G_i; 0] - V_i (X_i G_i)
        Vi = leaves[i][0]
        blk = -(Vi @ (X[i] @ Gi))
        blk[:b] += Gi
        Y[bounds[i]:bounds[i + 1]] = blk
    Y[:b] = Y1
    return Y, T, S[:, None] * R


def tsqr_hr3(P, leaf=512, fan=16):
    """tsqr_hr with a middle level for panels of more than fan leaves (device: n > 8224): the leaves' R factors are stacked `fan`
    at a time and factored again (the leaf kernel on the stack), the middle R factors go to the top QR.  Thin Q1 rows of leaf i in
    group g (position l):  Q_i . Mid_{g,l} . Top_g,  Mid_{g,l} = E_l - Vmid_g[block l] Xmid_g,  Top_g = E_g - Vtop[block g] Xtop."""
    P = np.asarray(P, dtype=np.float64)
    m, b = P.shape
    starts = list(range(0, m, leaf))
    if len(starts) > 1 and m - starts[-1] < b:
        starts.pop()
    bounds = starts + [m]
    L = len(starts)
    if L <= fan:
        return tsqr_hr(P, leaf)
    leaves, Rst = [], []
    for i in range(L):
        Vi, Ti, Ri = householder_qr_vt(P[bounds[i]:bounds[i + 1]])
        leaves.append((Vi, Ti))
        Rst.append(Ri)
    groups = [list(range(g0, min(g0 + fan, L))) for g0 in range(0, L, fan)]
    mids, Rmid = [], []
    for grp in groups:
        Vm, Tm, Rm = householder_qr_vt(np.vstack([Rst[i] for i in grp]))
        mids.append((Vm, Tm))
        Rmid.append(Rm)
    Vt, Tt, R = householder_qr_vt(np.vstack(Rmid))
    Xt = Tt @ Vt[:b].T
    eye = np.eye(b)
    top = [(eye if g == 0 else 0.0) - Vt[g * b:(g + 1) * b] @ Xt for g in range(len(groups))]
    X = [Ti @ Vi[:b].T for (Vi, Ti) in leaves]
    Qfac = []                                                # b x b factor behind every leaf's thin Q_i
    for g, grp in enumerate(groups):
        Vm, Tm = mids[g]
        Xm = Tm @ Vm[:b].T
        for l, i in enumerate(grp):
            mid = (eye if l == 0 else 0.0) - Vm[l * b:(l + 1) * b] @ Xm
            Qfac.append(mid @ top[g])
    Wt = (eye - leaves[0][0][:b] @ X[0]) @ Qfac[0]
    S = np.zeros(b)
    Y1 = np.eye(b)
    U = np.zeros((b, b))
    for j in range(b):
        S[j] = -1.0 if Wt[j, j] >= 0 else 1.0
        piv = 1.0 - S[j] * Wt[j, j]
        lcol = -S[j] * Wt[j + 1:, j] / piv
        Y1[j + 1:, j] = lcol
        Wt[j + 1:, j + 1:] -= np.outer(lcol, Wt[j, j + 1:])
    for j in range(b):
        U[j, j:] = -S[j:] * Wt[j, j:]
        U[j, j] += 1.0
    T = U @ np.linalg.inv(Y1).T
    Uinv = np.linalg.inv(U)
    Y = np.zeros((m, b))
    for i in range(L):
        Gi = -(Qfac[i] * S[None, :]) @ Uinv
        Vi = leaves[i][0]
        blk = -(Vi @ (X[i] @ Gi))
        blk[:b] += Gi
        Y[bounds[i]:bounds[i + 1]] = blk
    Y[:b] = Y1
    return Y, T, S[:, None] * R


def sy2sb(A, b, leaf=512, panel_qr=tsqr_hr):
    """dense symmetric A [n, n] (n a multiple of b) -> (B dense with bandwidth b, panels = [(row0, Y, T)]) with
    B = Q1^T A Q1,  Q1 = prod_p (I - Y_p T_p Y_p^T) (panel p acts on rows row0..n-1)."""
    A = np.array(A, dtype=np.float64)
    n = A.shape[0]
    panels = []
    for j0 in range(0, n - b, b):
        r0 = j0 + b
        m = n - r0
        if m < 2:
            break
        if panel_qr is tsqr_hr:
            Y, T, R = tsqr_hr(A[r0:, j0:j0 + b], leaf)
        else:
            Y, T, R = panel_qr(A[r0:, j0:j0 + b])
        A[r0:, j0:j0 + b] = 0.0
        rr = min(m, b)
        A[r0:r0 + rr, j0:j0 + b] = R[:rr]
        A[j0:j0 + b, r0:] = A[r0:, j0:j0 + b].T
        A22 = A[r0:, r0:]
        X = (A22 @ Y) @ T
        M = T.T @ (Y.T @ X)
        W = X - 0.5 * Y @ M
        A22 -= Y @ W.T + W @ Y.T
        panels.append((r0, Y, T))
    return A, panels


def apply_q1(panels, Z, agg=8):
    """Z <- Q1 Z with the panels aggregated `agg` at a time into one wide WY block:  T_wide^-1 = striu(V^T V) + diag(T_p^-1 blocks)"""
    Z = np.array(Z, dtype=np.float64)
    n = Z.shape[0]
    for g0 in reversed(range(0, len(panels), agg)):
        grp = panels[g0:g0 + agg]
        r0 = grp[0][0]
        b = grp[0][1].shape[1]
        V = np.zeros((n - r0, b * len(grp)))
        for i, (r, Y, T) in enumerate(grp):
            V[r - r0:, i * b:(i + 1) * b] = Y
        G = V.T @ V                                          # orthogonality of I - V T V^T  <=>  T^-1 + T^-T = V^T V
        Tw = np.linalg.inv(np.triu(G, 1) + 0.5 * np.diag(np.diag(G)))
        Z[r0:] -= V @ (Tw @ (V.T @ Z[r0:]))
    return Z


# ----------------------------------------------------------------------------------------------------------------------
# stage 2: bulge chasing
# ----------------------------------------------------------------------------------------------------------------------
def sb2st(B, b):
    """band matrix (dense storage, bandwidth b) -> (d, e, refl) with refl[(s, k)] = (row0, v, tau); T = Q2^T B Q2,
    Q2 = prod over sweeps s ascending, steps k ascending of H(s, k)."""
    A = np.array(B, dtype=np.float64)
    n = A.shape[0]
    refl = {}
    if b > 1:
        for s in range(n - 2):
            c0, c1 = s + 1, min(s + b, n - 1)
            v, tau, beta = house(A[c0:c1 + 1, s])
            A[c0, s] = A[s, c0] = beta
            A[c0 + 1:c1 + 1, s] = 0.0
            A[s, c0 + 1:c1 + 1] = 0.0
            k = 0
            while True:
                refl[(s, k)] = (c0, v, tau)
                # two-sided on the diagonal block
                D = A[c0:c1 + 1, c0:c1 + 1]
                p = tau * (D @ v)
                w = p - 0.5 * tau * (v @ p) * v
                D -= np.outer(v, w) + np.outer(w, v)
                r0, r1 = c1 + 1, min(c1 + b, n - 1)
                if r0 > n - 1:
                    break
                Bk = A[r0:r1 + 1, c0:c1 + 1]
                Bk -= tau * np.outer(Bk @ v, v)              # right-apply: fills the block (the bulge)
                v2, tau2, beta2 = house(Bk[:, 0])            # eliminate the bulge's first column
                Bk[0, 0] = beta2
                Bk[1:, 0] = 0.0
                if tau2 != 0.0:
                    u = v2 @ Bk[:, 1:]
                    Bk[:, 1:] -= tau2 * np.outer(v2, u)
                A[c0:c1 + 1, r0:r1 + 1] = Bk.T
                c0, c1, v, tau = r0, r1, v2, tau2
                k += 1
    return np.diag(A).copy(), np.diag(A, -1).copy(), refl


def apply_q2(refl, n, b, Z, g=None):
    """Z <- Q2 Z with the reflectors grouped g sweeps at a time (same step k) into WY blocks, applied in the order
    (group descending, k ascending) -- the order the device kernel uses.  g = None: reflector by reflector (reverse generation order)."""
    Z = np.array(Z, dtype=np.float64)
    if not refl:
        return Z
    if g is None:
        for key in sorted(refl.keys(), reverse=True):
            r0, v, tau = refl[key]
            if tau != 0.0:
                Z[r0:r0 + v.size] -= tau * np.outer(v, v @ Z[r0:r0 + v.size])
        return Z
    smax = max(s for s, _ in refl.keys())
    for S0 in reversed(range(0, smax + 1, g)):
        k = 0
        while True:
            members = [(s, refl[(s, k)]) for s in range(S0, min(S0 + g, smax + 1)) if (s, k) in refl]
            if not members:
                break
            rlo = min(r0 for _, (r0, v, tau) in members)
            rhi = max(r0 + v.size for _, (r0, v, tau) in members)
            V = np.zeros((rhi - rlo, len(members)))
            dinv = np.ones(len(members))
            for i, (s, (r0, v, tau)) in enumerate(members):
                if tau != 0.0:
                    V[r0 - rlo:r0 - rlo + v.size, i] = v
                    dinv[i] = 1.0 / tau
            Tinv = np.triu(V.T @ V, 1) + np.diag(dinv)
            Tm = np.linalg.inv(Tinv)
            Z[rlo:rhi] -= (V @ Tm) @ (V.T @ Z[rlo:rhi])
            k += 1
    return Z


def apply_q2_wavefront(refl, n, b, Z, g, lanes=4, lag=2):
    """Z <- Q2 Z in the order of the device's four-groups-per-pass kernel (sb2st.hip, q2_apply_wave4): `lanes` consecutive sweep
    groups travel down the rows together, group Gtop - w running `lag` steps behind group Gtop - w + 1 -- which keeps the blocks
    applied at the same time on disjoint rows (block (G, k) covers the row bands G + k and G + k + 1) and every pair of overlapping
    blocks in the order of `apply_q2`.  Blocks of one time step are applied in an arbitrary order here (reversed, on purpose)."""
    Z = np.array(Z, dtype=np.float64)
    if not refl:
        return Z
    smax = max(s for s, _ in refl.keys())
    ngroups = smax // g + 1

    def block(G, k):
        members = [(s, refl[(s, k)]) for s in range(G * g, min(G * g + g, smax + 1)) if (s, k) in refl]
        if not members:
            return None
        rlo = min(r0 for _, (r0, v, tau) in members)
        rhi = max(r0 + v.size for _, (r0, v, tau) in members)
        V = np.zeros((rhi - rlo, len(members)))
        dinv = np.ones(len(members))
        for i, (s, (r0, v, tau)) in enumerate(members):
            if tau != 0.0:
                V[r0 - rlo:r0 - rlo + v.size, i] = v
                dinv[i] = 1.0 / tau
        return rlo, rhi, V, np.linalg.inv(np.triu(V.T @ V, 1) + np.diag(dinv))

    for Gtop in range(ngroups - 1, -1, -lanes):
        nsteps = [max([k for (s, k) in refl.keys() if (Gtop - w) * g <= s < (Gtop - w) * g + g], default=-1) + 1 if Gtop - w >= 0 else 0
                  for w in range(lanes)]
        T = max([nk + lag * w for w, nk in enumerate(nsteps) if nk > 0], default=0)
        for t in range(T):
            rows = []
            for w in reversed(range(lanes)):
                k = t - lag * w
                if 0 <= k < nsteps[w]:
                    blk = block(Gtop - w, k)
                    if blk is None:
                        continue
                    rlo, rhi, V, Tm = blk
                    assert all(rhi <= a or rlo >= b_ for a, b_ in rows), "blocks of one time step overlap"
                    rows.append((rlo, rhi))
                    Z[rlo:rhi] -= (V @ Tm) @ (V.T @ Z[rlo:rhi])
    return Z


def apply_q2_t(refl, n, b, Z, g):
    """Z <- Q2^T Z, the forward order the device uses while the chase is still running: sweep groups ASCENDING (as they are
    produced), steps DESCENDING inside a group, every block transposed:  (I - V T V^T)^T = I - V T^T V^T."""
    Z = np.array(Z, dtype=np.float64)
    if not refl:
        return Z
    smax = max(s for s, _ in refl.keys())
    for S0 in range(0, smax + 1, g):
        kmax = max(k for (s, k) in refl.keys() if S0 <= s < S0 + g)
        for k in range(kmax, -1, -1):
            members = [(s, refl[(s, k)]) for s in range(S0, min(S0 + g, smax + 1)) if (s, k) in refl]
            if not members:
                continue
            rlo = min(r0 for _, (r0, v, tau) in members)
            rhi = max(r0 + v.size for _, (r0, v, tau) in members)
            V = np.zeros((rhi - rlo, len(members)))
            dinv = np.ones(len(members))
            for i, (s, (r0, v, tau)) in enumerate(members):
                if tau != 0.0:
                    V[r0 - rlo:r0 - rlo + v.size, i] = v
                    dinv[i] = 1.0 / tau
            Tm = np.linalg.inv(np.triu(V.T @ V, 1) + np.diag(dinv))
            Z[rlo:rhi] -= (V @ Tm.T) @ (V.T @ Z[rlo:rhi])
    return Z


# ----------------------------------------------------------------------------------------------------------------------
# stage 3: tridiagonal divide and conquer
# ----------------------------------------------------------------------------------------------------------------------
EPS = np.finfo(np.float64).eps


def secular_root(i, d, z2, rho, maxit=80):
    """i-th root (ascending) of 1 + rho sum_j z2_j / (d_j - lam) = 0, rho > 0, d strictly ascending, z2 > 0.
    Returned as (origin index o, offset mu): lam = d[o] + mu, with the offset accurate to working precision relative to itself.
    Safeguarded rational iteration in the shifted variable: bracket [lo, hi] in mu, a step of the two-pole "middle way"
    interpolation, bisection whenever the step leaves the bracket or does not shrink it enough."""
    k = d.size
    if k == 1:
        return 0, rho * z2[0]
    if i < k - 1:
        gap = d[i + 1] - d[i]
        # sign of the secular function at the midpoint decides the nearer pole
        mid = 0.5 * gap
        dl = d - d[i]
        fmid = 1.0 + rho * np.sum(z2 / (dl - mid))
        if fmid >= 0:            # root in the left half: origin d_i, mu in (0, gap/2]
            o, lo, hi = i, 0.0, mid
        else:                    # root in the right half: origin d_{i+1}, mu in [-gap/2, 0)
            o, lo, hi = i + 1, -mid, 0.0
    else:
        o = k - 1
        lo, hi = 0.0, rho * np.sum(z2)          # lam_max <= d_max + rho ||z||^2
    delta = d - d[o]                            # exact-ish differences to the origin
    # split the sum at the root's interval: psi (poles left of the root, <= 0 ... ) and phi (poles right)
    il = i                                     # poles 0..i lie left of the root, i+1.. right of it
    mu = 0.5 * (lo + hi)
    for it in range(maxit):
        den = delta - mu
        t = z2 / den
        psi = np.sum(t[:il + 1])
        phi = np.sum(t[il + 1:])
        dpsi = np.sum(t[:il + 1] / den[:il + 1])
        dphi = np.sum(t[il + 1:] / den[il + 1:])
        f = 1.0 + rho * (psi + phi)
        err = 8.0 * EPS * (1.0 + rho * (abs(psi) + abs(phi))) * k ** 0.5
        if f == 0.0 or abs(f) <= err:
            break
        if f > 0:
            hi = mu
        else:
            lo = mu
        # two-pole rational model through the nearest poles (LAPACK dlaed4's "middle way"):
        #   f(mu + eta) ~ c + a / (dk - mu - eta) + bq / (dk1 - mu - eta)
        if i < k - 1:
            dk, dk1 = delta[i] - mu, delta[i + 1] - mu        # dk < 0 < dk1
            a = rho * dpsi * dk * dk
            bq = rho * dphi * dk1 * dk1
            c = f - rho * dpsi * dk - rho * dphi * dk1
            # solve c + a/(dk - eta) + bq/(dk1 - eta) = 0  ->  c eta^2 - (c(dk+dk1) + a + bq) eta + (c dk dk1 + a dk1 + bq dk) = 0
            qa = c
            qb = -(c * (dk + dk1) + a + bq)
            qc = c * dk * dk1 + a * dk1 + bq * dk
            eta = _quad_small_root(qa, qb, qc)
        else:
            dk = delta[k - 1] - mu                            # < 0, last pole; model c + a / (dk - eta)
            a = rho * (dpsi) * dk * dk
            c = f - rho * dpsi * dk
            eta = dk + a / c if c != 0 else np.inf            # c + a/(dk-eta) = 0 -> eta = dk + a/c
        new = mu + eta
        if not np.isfinite(new) or new <= lo or new >= hi:
            new = 0.5 * (lo + hi)
        if new == mu or hi - lo <= 2 * EPS * max(abs(lo), abs(hi)):
            mu = new
            break
        mu = new
    return o, mu


def _quad_small_root(a, b, c):
    """the root of a x^2 + b x + c = 0 that is smaller in magnitude (the Newton-like correction), computed without cancellation"""
    if a == 0.0:
        return -c / b if b != 0 else np.inf
    disc = b * b - 4 * a * c
    if disc < 0:
        disc = 0.0
    sq = np.sqrt(disc)
    q = -0.5 * (b + (sq if b >= 0 else -sq))
    if q == 0.0:
        return 0.0
    r1 = c / q
    r2 = q / a
    return r1 if abs(r1) <= abs(r2) else r2


def merge_S(d1, d2, zraw, rho):
    """One D&C merge: eigen-decomposition of diag(d1, d2) + rho zraw zraw^T (zraw = [last row of Q1, first row of Q2], |zraw|^2 = 2
    before normalisation).  Returns (lam ascending [n], S [n, n]) such that the merged eigenvectors are blockdiag(Q1, Q2) @ S."""
    n1 = d1.size
    d = np.concatenate([d1, d2])
    n = d.size
    z = zraw / np.sqrt(2.0)
    rho = 2.0 * rho
    if rho < 0:                      # LAPACK dlaed2 negates the second half to make rho positive
        z = z.copy()
        z[n1:] = -z[n1:]
        d = d.copy()
        # (d + rho z z^T) with rho < 0: work with -(−d + |rho| z z^T)?  LAPACK instead flips z2's sign and uses |rho| ... only valid
        # because rho multiplies z z^T whose cross terms change sign; the diagonal terms need rho > 0.  The caller guarantees
        # rho > 0 by subtracting |beta| and flipping the sign of z's second half (see stedc) -- so this branch is unreachable.
        raise AssertionError("merge_S needs rho > 0")
    perm = np.argsort(d, kind="stable")
    ds, zs = d[perm], z[perm]
    tol = 8.0 * EPS * max(np.abs(ds).max(), np.abs(zs).max())
    # rows of Srow[j] will hold, for sorted position j, the combination of original columns it stands for: start with the permutation
    Rot = np.zeros((n, n))           # Rot[:, j] = column vector (in original coordinates) of sorted/rotated basis vector j
    Rot[perm, np.arange(n)] = 1.0
    defl = np.zeros(n, dtype=bool)
    if rho * np.abs(zs).max() <= tol:
        defl[:] = True
    else:
        defl = rho * np.abs(zs) <= tol
        prev = -1
        for j in range(n):
            if defl[j]:
                continue
            if prev >= 0:
                # try to deflate prev against j (close eigenvalues): rotate so that z[prev] = 0
                s_, c_ = zs[prev], zs[j]
                tau_ = np.hypot(c_, s_)
                t_ = ds[j] - ds[prev]
                c_, s_ = c_ / tau_, -s_ / tau_
                if abs(t_ * c_ * s_) <= tol:
                    zs[j] = tau_
                    zs[prev] = 0.0
                    a, bcol = Rot[:, prev].copy(), Rot[:, j].copy()
                    Rot[:, prev] = c_ * a + s_ * bcol
                    Rot[:, j] = -s_ * a + c_ * bcol
                    tdp = ds[prev] * c_ * c_ + ds[j] * s_ * s_
                    ds[j] = ds[prev] * s_ * s_ + ds[j] * c_ * c_
                    ds[prev] = tdp
                    defl[prev] = True
            prev = j
    nd = np.flatnonzero(~defl)
    k = nd.size
    lam = np.empty(n)
    S = np.zeros((n, n))
    if k:
        dk, zk = ds[nd], zs[nd]
        z2 = zk * zk
        org = np.empty(k, dtype=int)
        mu = np.empty(k)
        for i in range(k):
            org[i], mu[i] = secular_root(i, dk, z2, rho)
        # Gu-Eisenstat: zhat_j^2 = prod_i (lam_i - d_j) / prod_{i != j} (d_i - d_j)   (with the rho factor folded in: / rho)
        # lam_i - d_j = (d[org_i] - d_j) + mu_i
        diff = (dk[org][:, None] - dk[None, :]) + mu[:, None]       # [i, j] = lam_i - d_j
        zh = np.empty(k)
        for j in range(k):
            num = diff[:, j]
            den = np.delete(dk - dk[j], j)
            # interleave to avoid over/underflow: product of ratios
            val = num[j] if True else 0.0
            ratios = np.delete(num, j) / den
            zh[j] = np.sqrt(abs(val * np.prod(ratios) / rho))
        zh = np.copysign(zh, zk)
        Vk = zh[None, :] / (-diff)                                   # v_i[j] = zhat_j / (d_j - lam_i)
        Vk /= np.linalg.norm(Vk, axis=1, keepdims=True)
        lam_k = dk[org] + mu
        S[:, :k] = Rot[:, nd] @ Vk.T
        lam[:k] = lam_k
    dd = np.flatnonzero(defl)
    S[:, k:] = Rot[:, dd]
    lam[k:] = ds[dd]
    order = np.argsort(lam, kind="stable")
    return lam[order], S[:, order]


def stedc(d, e, leaf=64):
    """eigen-decomposition of the symmetric tridiagonal (d, e) by divide and conquer; leaves by numpy's eigh (device: LDS Jacobi)"""
    d = np.array(d, dtype=np.float64)
    e = np.array(e, dtype=np.float64)
    n = d.size
    if n <= leaf:
        T = np.diag(d) + np.diag(e, 1) + np.diag(e, -1)
        return np.linalg.eigh(T)
    n1 = (n // 2 + leaf - 1) // leaf * leaf if n > 2 * leaf else n // 2
    n1 = min(max(n1, 1), n - 1)
    beta = e[n1 - 1]
    ab = abs(beta)
    d1, d2 = d[:n1].copy(), d[n1:].copy()
    d1[-1] -= ab
    d2[0] -= ab
    l1, Q1 = stedc(d1, e[:n1 - 1], leaf)
    l2, Q2 = stedc(d2, e[n1:], leaf)
    sgn = 1.0 if beta >= 0 else -1.0
    zraw = np.concatenate([Q1[-1, :], sgn * Q2[0, :]])
    lam, S = merge_S(l1, l2, zraw, ab)
    Q = np.zeros((n, n))
    Q[:n1] = Q1 @ S[:n1]
    Q[n1:] = Q2 @ S[n1:]
    return lam, Q


# ----------------------------------------------------------------------------------------------------------------------
def eigh_twostage(A, b=32, leaf=512, g=32, agg=8):
    """the whole route on an n x n symmetric matrix; n is padded to a multiple of 64 with decoupled diagonal entries above the
    spectrum (they never mix: every reflector component on a padded row is exactly zero)"""
    A = np.asarray(A, dtype=np.float64)
    n = A.shape[0]
    npad = (n + 63) // 64 * 64
    Ap = np.zeros((npad, npad))
    Ap[:n, :n] = A
    big = 2.0 * np.abs(A).sum(1).max() + 1.0
    for i in range(n, npad):
        Ap[i, i] = big * (1.0 + (i - n) / 64.0)
    Bd, panels = sy2sb(Ap, b, leaf)
    d, e, refl = sb2st(Bd, b)
    lam, Z = stedc(d, e)
    Z = apply_q2(refl, npad, b, Z, g)
    Z = apply_q1(panels, Z, agg)
    return lam[:n], Z[:n, :n]
