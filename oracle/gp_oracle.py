"""CPU oracle for the GP hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A numpy/scipy (float64) restatement of the arithmetic the reference (IceLab-X/FidelityFusion) performs
on its GP hot path.  Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may import
this module; nothing under fidelityfusion_amd/ does.

Parity pin: the reference ships no tests or golden vectors for this path, so this
oracle is pinned against outputs of the reference itself, imported in the build container by
tests/golden/gen_goldens.py and committed as tests/golden/*.npz (see tests/test_oracle_golden.py:
agreement <= 1e-10 relative in fp64 on every fixture).

The dense factorisations live in a third-party dependency of the reference (PyTorch/ATen -> LAPACK
dpotrf/dtrsm/dpotrs; not vendored and not pinned by the reference: a comment says torch 1.11.0,
`GaussianProcess/cigp_v10.py:11`, committed logs say 2.1.1+cpu, `FidelityFusion_Models/log/ResGP/train.log:1`).
Their published algorithm (Cholesky A = L L^T, forward/back substitution) is restated twice here:
through LAPACK (`scipy.linalg`) for speed and as plain loops (`cholesky_unblocked`, `solve_lower_loops`)
for small cases; the two are cross-checked in the tests.

All `file:line` citations are relative to the reference root.
"""
import numpy as np
import scipy.linalg as sla

EPS = 1e-9          # GaussianProcess/kernel.py:21 ; gp_computation_pack.py:15
JITTER = 1e-6       # GaussianProcess/cigp_v10.py:13 ; gp_computation_pack.py:16 ; MFGP_ver2023May/base_gp/cigp.py:4
PI_TRUNC = 3.1415   # GaussianProcess/cigp_v10.py:15 ; gp_computation_pack.py:17 ; base_gp/cigp.py:6


# ----------------------------------------------------------------------------------------------------------
# kernels
# ----------------------------------------------------------------------------------------------------------
def cdist_sq(a, b):
    """`torch.cdist(a, b, p=2)**2` as called at GaussianProcess/kernel.py:104.

    ATen (third-party, unpinned) evaluates p=2 distances with more than 25 rows on either side through the
    expanded form ||a||^2 + ||b||^2 - 2ab^T, clamps at 1e-30 before the sqrt, and the reference squares
    the result again; with <=25 rows on both sides it sums squared differences directly.
    """
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    if a.shape[0] > 25 or b.shape[0] > 25:
        an = (a * a).sum(1, keepdims=True)
        bn = (b * b).sum(1, keepdims=True)
        r = np.concatenate([-2.0 * a, an, np.ones_like(an)], 1) @ np.concatenate([b, np.ones_like(bn), bn], 1).T
        return np.sqrt(np.maximum(r, 1e-30)) ** 2
    diff = a[:, None, :] - b[None, :, :]
    return np.sqrt((diff * diff).sum(-1)) ** 2


def ard_kernel(x1, x2, length_scales, signal_variance):
    """K1: ARDKernel.forward, GaussianProcess/kernel.py:88-105."""
    ell = np.abs(np.asarray(length_scales, dtype=np.float64)) + EPS
    sq = cdist_sq(np.asarray(x1) / ell, np.asarray(x2) / ell)
    return np.abs(float(np.ravel(signal_variance)[0])) * np.exp(-0.5 * sq)


def matern_profile(sq, nu, rho=1.0):
    """phi_nu(sq; rho) of MaternKernel.forward, GaussianProcess/kernel.py:161-166 (sq = cdist^2, clamped)."""
    if nu == 0.5:
        return np.exp(-np.sqrt(sq) / rho)
    if nu == 1.5:
        return (1 + np.sqrt(3 * sq) / rho) * np.exp(-np.sqrt(3 * sq) / rho)
    if nu == 2.5:
        return (1 + np.sqrt(5 * sq) / rho + 5 / 3 * sq / rho ** 2) * np.exp(-np.sqrt(5 * sq) / rho)
    raise ValueError("nu")


def matern_profile_m2d(sq, nu, rho=1.0):
    """-2 d(phi_nu)/d(sq): the factor of the length-scale gradient (equals phi for the squared exponential)."""
    if nu == 0.5:
        r = np.sqrt(sq)
        return np.exp(-r / rho) / (rho * r)
    if nu == 1.5:
        return 3.0 / rho ** 2 * np.exp(-np.sqrt(3 * sq) / rho)
    a = np.sqrt(5 * sq) / rho
    return 5.0 / (3.0 * rho ** 2) * (1 + a) * np.exp(-a)


def matern_kernel(x1, x2, length_scales, signal_variance, nu=2.5, rho=1.0):
    """MaternKernel.forward, GaussianProcess/kernel.py:147-166."""
    ell = np.abs(np.asarray(length_scales, dtype=np.float64)) + EPS
    sq = cdist_sq(np.asarray(x1) / ell, np.asarray(x2) / ell)
    return np.abs(float(np.ravel(signal_variance)[0])) * matern_profile(sq, nu, rho)


def sqdist_expanded(x1, x2):
    """||a||^2 + ||b||^2 - 2ab^T, unclamped (kernel.py:271 ; SE_kernel.py:37-41)."""
    x1 = np.asarray(x1, dtype=np.float64)
    x2 = np.asarray(x2, dtype=np.float64)
    return (x1 * x1).sum(1)[:, None] + (x2 * x2).sum(1)[None, :] - 2.0 * (x1 @ x2.T)


def se_kernel(x1, x2, length_scale, signal_variance):
    """K2: SquaredExponentialKernel.forward, GaussianProcess/kernel.py:258-272 (raw params are logs)."""
    ls = float(np.ravel(length_scale)[0])
    sv = float(np.ravel(signal_variance)[0])
    return np.exp(sv) ** 2 * np.exp(-0.5 * sqdist_expanded(x1, x2) / np.exp(ls) ** 2)


def matern_scalar_kernel(x1, x2, length_scale, signal_variance, nu):
    """MaternKernel_scalarLengthScale.forward, GaussianProcess/kernel.py:346-347: sv^2 (1 + sqrt(3 sqdist) / ls^2)^-nu on
    the unclamped norm-expansion distance (NaN where rounding leaves it negative, as in the reference)."""
    sq = sqdist_expanded(x1, x2)
    ls, sv, nu = (float(np.ravel(v)[0]) for v in (length_scale, signal_variance, nu))
    with np.errstate(invalid="ignore"):
        return sv ** 2 * np.power(1.0 + np.sqrt(3.0 * sq) / ls ** 2, -nu)


def se_kernel_2023(x1, x2, length_scale, scale, exp_format):
    """K3: SE_kernel.forward, MFGP_ver2023May/kernel/SE_kernel.py:20-44 (inputs >2-D are flattened :29-32)."""
    ls = float(np.ravel(length_scale)[0])
    sc = float(np.ravel(scale)[0])
    if exp_format:
        ls, sc = np.exp(ls), np.exp(sc)
    x1 = np.asarray(x1, dtype=np.float64).reshape(len(x1), -1) / ls
    x2 = np.asarray(x2, dtype=np.float64).reshape(len(x2), -1) / ls
    return sc * np.exp(-0.5 * sqdist_expanded(x1, x2))


# ----------------------------------------------------------------------------------------------------------
# dense factorisation / solves (rows C1, T1)
# ----------------------------------------------------------------------------------------------------------
class NotPositiveDefinite(np.linalg.LinAlgError):
    pass


def cholesky_lower(S):
    """torch.linalg.cholesky (cigp_v10.py:35,61 etc.): lower L with S = L L^T; raises if not PD."""
    try:
        return sla.cholesky(np.asarray(S, dtype=np.float64), lower=True, check_finite=False)
    except sla.LinAlgError as e:  # pragma: no cover
        raise NotPositiveDefinite(str(e))


def cholesky_unblocked(S):
    """Textbook column Cholesky (the algorithm LAPACK dpotf2 publishes); small cases only."""
    A = np.array(S, dtype=np.float64)
    n = A.shape[0]
    L = np.zeros_like(A)
    for j in range(n):
        d = A[j, j] - L[j, :j] @ L[j, :j]
        if not d > 0.0:
            raise NotPositiveDefinite("leading minor %d not positive" % (j + 1))
        L[j, j] = np.sqrt(d)
        L[j + 1:, j] = (A[j + 1:, j] - L[j + 1:, :j] @ L[j, :j]) / L[j, j]
    return L


def solve_lower(L, B):
    """torch.triangular_solve(B, L, upper=False) (cigp_v10.py:36,63)."""
    return sla.solve_triangular(L, B, lower=True, check_finite=False)


def solve_lower_loops(L, B):
    X = np.array(B, dtype=np.float64)
    for i in range(L.shape[0]):
        X[i] = (X[i] - L[i, :i] @ X[:i]) / L[i, i]
    return X


def cho_solve(L, B):
    """torch.cholesky_solve(B, L) = Sigma^{-1} B (cigp_v10.py:39 ; gp_computation_pack.py:76,106)."""
    return sla.cho_solve((L, True), B, check_finite=False)


# ----------------------------------------------------------------------------------------------------------
# Sigma assembly variants (rows S1-S4)
# ----------------------------------------------------------------------------------------------------------
def sigma_cigp(K, log_beta, y_var=None):
    """S1: cigp_v10.py:57-60.  y_var is an N x N matrix of which only the diagonal is used."""
    n = K.shape[0]
    S = K + (np.exp(-float(np.ravel(log_beta)[0])) + JITTER) * np.eye(n)
    if y_var is not None:
        S = S + np.diag(np.diag(np.asarray(y_var, dtype=np.float64)))
    return S


def sigma_pack(K, log_beta):
    """S2: gp_computation_pack.py:125-126 (jitter scaled by mean(K))."""
    n = K.shape[0]
    return K + (np.exp(-float(np.ravel(log_beta)[0])) + JITTER * K.mean()) * np.eye(n)


def sigma_basic(K, noise_variance, y_var=None):
    """S3: gp_basic.py:63-65,117-119 (noise_variance^2 on the diagonal, FULL y_var matrix added, no jitter)."""
    S = K + float(np.ravel(noise_variance)[0]) ** 2 * np.eye(K.shape[0])
    if y_var is not None:
        S = S + np.asarray(y_var, dtype=np.float64)
    return S


def noise_from_box(value):
    """GP_noise_box.get() in 'exp' format (MFGP_ver2023May/utils/gp_noise.py:17-22): the parameter is
    hard-wired float32, so exp() is evaluated in float32 before promotion to the fp64 Sigma."""
    return np.exp(np.float32(np.ravel(value)[0]))


def inv_noise_from_box(value):
    """`_noise.pow(-1)` (base_gp/cigp.py:81,126,93): still float32 arithmetic, promoted afterwards."""
    return float(np.float32(1.0) / noise_from_box(value))


def sigma_2023(K, inv_noise, y_var=0.0):
    """S4: base_gp/cigp.py:124-127; inv_noise = GP_noise_box.get().pow(-1); y_var broadcasts to ALL entries.
    The reference adds the jitter and the noise as two separate diagonal adds (:124,:126)."""
    n = K.shape[0]
    return (K + JITTER * np.eye(n)) + float(inv_noise) * np.eye(n) + y_var


# ----------------------------------------------------------------------------------------------------------
# likelihoods (rows L1, L2)
# ----------------------------------------------------------------------------------------------------------
def nll_v1_from_sigma(S, Y, pi_const=PI_TRUNC):
    """L1: 0.5||L^-1 Y||_F^2 + d sum(log L_ii) + 0.5 N d log(2*3.1415)   (cigp_v10.py:61-68)."""
    n, d = Y.shape
    L = cholesky_lower(S)
    G = solve_lower(L, Y)
    return 0.5 * (G * G).sum() + d * np.log(np.diag(L)).sum() + 0.5 * n * d * np.log(2.0 * pi_const), L, G


def ll_v2(Y, cov):
    """L2: Gaussian_log_likelihood(..., 'cholesky3'), gp_computation_pack.py:65-80.
    gamma = Sigma^{-1} y, so the quadratic term is y^T Sigma^{-2} y (reference quirk); true pi."""
    n, d = Y.shape
    L = cholesky_lower(cov)
    A = cho_solve(L, Y)
    return -0.5 * ((A * A).sum() + 2.0 * d * np.log(np.diag(L)).sum() + n * d * np.log(2.0 * np.pi)), L, A


def ll_alt(Y, cov, method):
    """L3: the other Kinv_method branches of Gaussian_log_likelihood (gp_computation_pack.py:55-63,82-88), quirks kept:
    'cholesky1' / 'direct' use the true y^T Sigma^-1 y but return a [d, d] matrix with the log-determinant counted
    twice; 'cholesky2' the Sigma^-2 form; the two torch_distribution_MN* branches evaluate N(y_i | y_i, cov) for the N
    rows of y (which must have length N): N copies of the normalising constant."""
    n = Y.shape[0]
    L = cholesky_lower(cov)
    logdet = 2.0 * np.log(np.diag(L)).sum()
    if method.startswith("torch_distribution_MN"):
        assert Y.shape[1] == n
        return np.full(n, -0.5 * n * np.log(2.0 * np.pi) - 0.5 * logdet)
    A = cho_solve(L, Y)
    quad = Y.T @ A if method in ("cholesky1", "direct") else A.T @ A
    return -0.5 * (quad + 2.0 * logdet + n * np.log(2.0 * np.pi))


def ll_v2_grads(Y, cov):
    """Closed-form d(LL_v2)/d(cov), d(LL_v2)/dY."""
    ll, L, A = ll_v2(Y, cov)
    d = Y.shape[1]
    B = cho_solve(L, A)
    Sinv = cho_solve(L, np.eye(cov.shape[0]))
    # torch's cholesky backward returns the symmetrised gradient w.r.t. cov
    g_cov = 0.5 * (A @ B.T + B @ A.T) - 0.5 * d * Sinv
    return ll, g_cov, -B


# ----------------------------------------------------------------------------------------------------------
# closed-form gradients of V1 (row G1)
# ----------------------------------------------------------------------------------------------------------
def _G_matrix(L, Y, d):
    n = L.shape[0]
    A = cho_solve(L, Y)
    Sinv = cho_solve(L, np.eye(n))
    return 0.5 * (d * Sinv - A @ A.T), A


def cigp_ll_and_grads(X, Y, length_scales, signal_variance, log_beta, y_var=None, kind="ard", nu=None, rho=1.0):
    """cigp.negative_log_likelihood (returns +LL = -nll; cigp_v10.py:50-69) and d(LL)/d{params, Y}.

    kind='ard': K1 with raw length_scales[D] (nu = 0.5 | 1.5 | 2.5 switches to MaternKernel's profile);
    kind='se': K2 with scalar log length_scale.
    """
    X = np.asarray(X, dtype=np.float64)
    Y = np.asarray(Y, dtype=np.float64)
    n, d = Y.shape
    beta_inv = np.exp(-float(np.ravel(log_beta)[0]))
    if kind == "ard":
        p = np.asarray(length_scales, dtype=np.float64)
        s = float(np.ravel(signal_variance)[0])
        ell = np.abs(p) + EPS
        sqc = cdist_sq(X / ell, X / ell)
        E = np.exp(-0.5 * sqc) if nu is None else matern_profile(sqc, nu, rho)
        K = np.abs(s) * E
    else:
        ls = float(np.ravel(length_scales)[0])
        sv = float(np.ravel(signal_variance)[0])
        sq = sqdist_expanded(X, X)
        K = np.exp(sv) ** 2 * np.exp(-0.5 * sq / np.exp(ls) ** 2)
    S = sigma_cigp(K, log_beta, y_var)
    nll, L, _ = nll_v1_from_sigma(S, Y)
    G, A = _G_matrix(L, Y, d)
    g = {"log_beta": -(-beta_inv * np.trace(G)), "Y": -A}
    if kind == "ard":
        if nu is None:
            W = G * K
        else:  # W = -2 G o dK/d(sq); entries sitting on cdist's clamp carry no gradient (the diagonal)
            W = np.where(sqc > 1e-30, G * np.abs(s) * matern_profile_m2d(np.maximum(sqc, 1e-30), nu, rho), 0.0)
        g["signal_variance"] = -(np.sign(s) * (G * E).sum())
        W = W.copy()
        np.fill_diagonal(W, 0.0)   # (x_i - x_i) = 0 exactly; keeps a huge Matern-1/2 W_ii out of the r_i x_i^2 - x_i W_ii x_i cancellation
        Xs = X
        r = W.sum(1)
        quad = 2.0 * ((r[:, None] * Xs * Xs).sum(0) - (Xs * (W @ Xs)).sum(0))
        g["length_scales"] = -(np.sign(p) / ell ** 3 * quad)
    else:
        g["signal_variance"] = -(2.0 * (G * K).sum())
        g["length_scale"] = -((G * K * sq).sum() / np.exp(ls) ** 2)
    return -nll, g


def pack_ll_and_grads(X, Y, length_scales, signal_variance, log_beta):
    """gp_computation_pack.negative_log_likelihood (:120-136), ARD kernel, with the mean(K) jitter term."""
    X = np.asarray(X, dtype=np.float64)
    Y = np.asarray(Y, dtype=np.float64)
    n, d = Y.shape
    p = np.asarray(length_scales, dtype=np.float64)
    s = float(np.ravel(signal_variance)[0])
    ell = np.abs(p) + EPS
    E = np.exp(-0.5 * cdist_sq(X / ell, X / ell))
    K = np.abs(s) * E
    beta_inv = np.exp(-float(np.ravel(log_beta)[0]))
    S = sigma_pack(K, log_beta)
    nll, L, _ = nll_v1_from_sigma(S, Y)
    G, A = _G_matrix(L, Y, d)
    trG = np.trace(G)
    # dSigma = dK + JITTER * mean(dK) * I  ->  effective weight on dK is G + JITTER*trG/n^2 * 11^T
    Geff = G + JITTER * trG / (n * n)
    W = Geff * K
    r = W.sum(1)
    quad = 2.0 * ((r[:, None] * X * X).sum(0) - (X * (W @ X)).sum(0))
    g = {
        "log_beta": -(-beta_inv * trG),
        "Y": -A,
        "signal_variance": -(np.sign(s) * (Geff * E).sum()),
        "length_scales": -(np.sign(p) / ell ** 3 * quad),
    }
    return -nll, g


def cigp2023_nll_and_grads(X, Y, length_scale, scale, exp_format, noise_value, y_var=0.0):
    """2023 CIGP.compute_loss (+nll; base_gp/cigp.py:99-136) with SE_kernel (K3) and an 'exp' noise box
    (utils/gp_noise.py:9-24): noise = exp(value); Sigma = K + (1e-6 + 1/noise) I + y_var."""
    X = np.asarray(X, dtype=np.float64).reshape(len(X), -1)
    Y = np.asarray(Y, dtype=np.float64)
    n, d = Y.shape
    ls_raw = float(np.ravel(length_scale)[0])
    sc_raw = float(np.ravel(scale)[0])
    ls = np.exp(ls_raw) if exp_format else ls_raw
    sc = np.exp(sc_raw) if exp_format else sc_raw
    sq = sqdist_expanded(X / ls, X / ls)
    E = np.exp(-0.5 * sq)
    K = sc * E
    inv_noise = inv_noise_from_box(noise_value)
    S = sigma_2023(K, inv_noise, y_var)
    nll, L, _ = nll_v1_from_sigma(S, Y)
    G, A = _G_matrix(L, Y, d)
    g = {"Y": A, "noise_value": -np.trace(G) * inv_noise}
    dls = (G * K * sq).sum() / ls          # dK/dls = K * sq / ls   (sq already holds the 1/ls^2 factor)
    dsc = (G * E).sum()
    if exp_format:
        dls *= ls
        dsc *= sc
    g["length_scale"] = dls
    g["scale"] = dsc
    return nll, g


# ----------------------------------------------------------------------------------------------------------
# posteriors (rows P1-P3)
# ----------------------------------------------------------------------------------------------------------
# ---------------------------------------------------------------------------------------------------------------
# Remaining kernels of GaussianProcess/kernel.py and their parameter gradients for a given upstream
# weight  Gw = d(value)/dK  (for value = +LL of the V1 likelihood: Gw = -G; for a product kernel: Gw o K_other).
# ---------------------------------------------------------------------------------------------------------------
def linear_kernel(x1, x2, length_scales, signal_variance, center):
    """LinearKernel.forward, kernel.py:45-63: |s| * ((x1 - c)/l) ((x2 - c)/l)^T  (raw l: no abs, no eps)."""
    ls = np.asarray(length_scales, dtype=np.float64)
    c = np.asarray(center, dtype=np.float64)
    z1 = (np.asarray(x1, dtype=np.float64) - c) / ls
    z2 = (np.asarray(x2, dtype=np.float64) - c) / ls
    return z1 @ z2.T * abs(float(np.ravel(signal_variance)[0]))


def linear_kernel_grads(X, length_scales, signal_variance, center, Gw):
    ls = np.asarray(length_scales, dtype=np.float64)
    c = np.asarray(center, dtype=np.float64)
    s = float(np.ravel(signal_variance)[0])
    Z = (np.asarray(X, dtype=np.float64) - c) / ls
    dZ = abs(s) * (Gw + Gw.T) @ Z                     # value depends on Z through both factors
    return {"length_scales": -(dZ * Z).sum(0) / ls,  # dZ/dl = -Z/l
            "center": -dZ.sum(0) / ls,
            "signal_variance": np.sign(s) * (Gw * (Z @ Z.T)).sum()}


def rq_kernel(x1, x2, length_scale, signal_variance, alpha):
    """RationalQuadraticKernel.forward, kernel.py:297-310 (norm-expansion distance, scalar raw parameters)."""
    ls, sv, al = (float(np.ravel(v)[0]) for v in (length_scale, signal_variance, alpha))
    return sv ** 2 * np.power(1.0 + 0.5 * sqdist_expanded(x1, x2) / al / ls ** 2, -al)


def rq_kernel_grads(X, length_scale, signal_variance, alpha, Gw):
    ls, sv, al = (float(np.ravel(v)[0]) for v in (length_scale, signal_variance, alpha))
    u = 0.5 * sqdist_expanded(X, X) / al / ls ** 2
    base = 1.0 + u
    phi = np.power(base, -al)
    return {"signal_variance": 2.0 * sv * (Gw * phi).sum(),
            "length_scale": sv ** 2 * (Gw * (-al) * np.power(base, -al - 1.0) * (-2.0 * u / ls)).sum(),
            "alpha": sv ** 2 * (Gw * phi * (u / base - np.log(base))).sum()}


def ard_kernel_grads(X, length_scales, signal_variance, Gw, nu=None, rho=1.0):
    """ARDKernel (nu=None) / MaternKernel gradients for an upstream weight on K(X, X); same algebra as
    cigp_ll_and_grads, sign-free."""
    X = np.asarray(X, dtype=np.float64)
    p = np.asarray(length_scales, dtype=np.float64)
    s = float(np.ravel(signal_variance)[0])
    ell = np.abs(p) + EPS
    sqc = cdist_sq(X / ell, X / ell)
    E = np.exp(-0.5 * sqc) if nu is None else matern_profile(sqc, nu, rho)
    Gs = 0.5 * (Gw + Gw.T)
    if nu is None:
        W = Gs * np.abs(s) * E
    else:
        W = np.where(sqc > 1e-30, Gs * np.abs(s) * matern_profile_m2d(np.maximum(sqc, 1e-30), nu, rho), 0.0)
    W = W.copy()
    np.fill_diagonal(W, 0.0)
    r = W.sum(1)
    quad = 2.0 * ((r[:, None] * X * X).sum(0) - (X * (W @ X)).sum(0))   # sum_ij W_ij (x_ik - x_jk)^2
    # dK/dl_k = K' d(sq)/dl_k with d(sq)/dl_k = -2 sign(p) (x_ik - x_jk)^2 / l^3 and W = -2 Gw o K'
    return {"signal_variance": np.sign(s) * (Gw * E).sum(),
            "length_scales": np.sign(p) / ell ** 3 * quad}


def composed_ll_and_grads(X, Y, parts, combine, sigma_fn, dsigma_scalar, variant="v1"):
    """LL (+LL as the reference's `negative_log_likelihood` returns it, or the V2 `log_likelihood`) of a GP whose
    kernel is the sum / product of `parts` = [(K_fn(X)->K, grads_fn(X, Gw)->dict), ...]; sigma_fn(K) -> Sigma.
    Returns LL, [grads per part], d LL / d Sigma (symmetric), d LL / dY."""
    Ks = [kf(X) for kf, _ in parts]
    if isinstance(combine, str):
        combine = (combine, 0, 1)
    K = tree_value(combine, Ks)
    S = sigma_fn(K)
    Y = np.asarray(Y, dtype=np.float64)
    d = Y.shape[1]
    if variant == "v1":
        nll, L, _ = nll_v1_from_sigma(S, Y)
        G, A = _G_matrix(L, Y, d)
        ll, dS, dY = -nll, -G, -A
    else:
        ll, dS, dY = ll_v2_grads(Y, S)
    dK = dS + dsigma_scalar(dS, K)      # chain through any K-dependent diagonal term (pack's mean(K) jitter)
    ups = [None] * len(parts)
    tree_upstreams(combine, Ks, dK, ups)
    return ll, [gf(X, ups[i]) for i, (_, gf) in enumerate(parts)], dS, dY


def tree_value(expr, Ks):
    """A nested Sum / Product composition (SumKernel / ProductKernel objects holding each other, kernel.py:172-236) as an expression:
    an int names a leaf's matrix in Ks, ("sum" | "prod", left, right) a node."""
    if isinstance(expr, int):
        return Ks[expr]
    op, a, b = expr
    va, vb = tree_value(a, Ks), tree_value(b, Ks)
    return va + vb if op == "sum" else va * vb


def tree_upstreams(expr, Ks, G, out):
    """reverse sweep of tree_value: out[leaf] = d sum(G o root) / d leaf matrix (what autograd hands every leaf kernel)"""
    if isinstance(expr, int):
        out[expr] = G if out[expr] is None else out[expr] + G
        return
    op, a, b = expr
    if op == "sum":
        tree_upstreams(a, Ks, G, out)
        tree_upstreams(b, Ks, G, out)
    else:
        tree_upstreams(a, Ks, G * tree_value(b, Ks), out)
        tree_upstreams(b, Ks, G * tree_value(a, Ks), out)


def se_kernel_grads(X, length_scale, signal_variance, Gw):
    """SquaredExponentialKernel (raw parameters are logs, kernel.py:258-272): d sum(Gw o K) / d{length_scale, signal_variance}"""
    ls, sv = float(np.ravel(length_scale)[0]), float(np.ravel(signal_variance)[0])
    sq = sqdist_expanded(X, X)
    K = np.exp(sv) ** 2 * np.exp(-0.5 * sq / np.exp(ls) ** 2)
    return {"signal_variance": 2.0 * (Gw * K).sum(), "length_scale": (Gw * K * sq).sum() / np.exp(ls) ** 2}


def linear_input_grads(x1, x2, length_scales, signal_variance, center, Gw):
    """d sum(Gw o K_lin(x1, x2)) / dx1, / dx2 (autograd through kernel.py:45-63)"""
    ls = np.asarray(length_scales, dtype=np.float64)
    c = np.asarray(center, dtype=np.float64)
    a = abs(float(np.ravel(signal_variance)[0]))
    return a * (Gw @ (np.asarray(x2) - c)) / ls ** 2, a * (Gw.T @ (np.asarray(x1) - c)) / ls ** 2


def ard_input_grads(x1, x2, length_scales, signal_variance, Gw, nu=None, rho=1.0):
    """d sum(Gw o K(x1, x2)) / dx1, / dx2 for ARDKernel (nu=None) / MaternKernel -- what autograd returns through
    kernel.py:100-105 / :138-169 (entries on cdist's clamp carry no gradient)."""
    x1, x2 = np.asarray(x1, dtype=np.float64), np.asarray(x2, dtype=np.float64)
    ell = np.abs(np.asarray(length_scales, dtype=np.float64)) + EPS
    s = float(np.ravel(signal_variance)[0])
    sq = cdist_sq(x1 / ell, x2 / ell)
    m2d = np.exp(-0.5 * sq) if nu is None else matern_profile_m2d(np.maximum(sq, 1e-30), nu, rho)
    Wt = np.where(sq > 1e-30, Gw * np.abs(s) * m2d, 0.0)          # = -2 Gw o dK/d(sq)
    w2 = 1.0 / ell ** 2
    g1 = -w2 * (Wt.sum(1)[:, None] * x1 - Wt @ x2)
    g2 = w2 * (Wt.T @ x1 - Wt.sum(0)[:, None] * x2)
    return g1, g2


def conditional_gaussian_grads(Y, Sigma, K_s, Gmu, Gc):
    """Gradients of sum(Gmu o mu) + sum(Gc o cov) for mu, cov = conditional_gaussian(...): (dY, dSigma symmetric,
    dK_s, dK_ss)."""
    L = cholesky_lower(Sigma)
    alpha = cho_solve(L, np.asarray(Y, dtype=np.float64))
    B = cho_solve(L, np.asarray(K_s, dtype=np.float64))
    Gs = Gc + Gc.T
    dy = B @ Gmu
    dS = -0.5 * (dy @ alpha.T + alpha @ dy.T) + 0.5 * B @ Gs @ B.T
    return dy, dS, alpha @ Gmu.T - B @ Gs, Gc


def _mode_dot(t, M, mode):
    return np.moveaxis(np.tensordot(t, M, axes=([mode], [1])), -1, mode)


def hogp_ll(Ks, Y, noise_variance):
    """HOGP_simple.log_likelihood, two_fidelity_models/hogp_simple.py:79-126: Ks = [K_x, K_1, ...] (already evaluated),
    Y [N, d1, ...].  Returns (loss = +NLL / (N prod d), A, g, eigen pairs)."""
    eig = [np.linalg.eigh(np.asarray(K, dtype=np.float64)) for K in Ks]
    A = eig[0][0]
    for lam, _ in eig[1:]:
        A = np.multiply.outer(A, lam)
    A = A + 1.0 / float(np.ravel(noise_variance)[0])
    T1 = np.asarray(Y, dtype=np.float64)
    for i, (_, U) in enumerate(eig):
        T1 = _mode_dot(T1, U.T, i)
    T3, g = T1 * A ** -0.5, T1 / A
    for i, (_, U) in enumerate(eig):
        T3 = _mode_dot(T3, U, i)
        g = _mode_dot(g, U, i)
    nd = A.size
    ll = -0.5 * nd * np.log(2.0 * np.pi) - 0.5 * np.log(A).sum() - 0.5 * (T3 ** 2).sum()
    return -ll / nd, A, g, eig


def hogp_forward(K_star, K_ss_diag, Ks, A, g, eig):
    """HOGP_simple.forward (:46-77); K_star = k(x_test, x_train)."""
    mean = g
    for i, M in enumerate([K_star] + list(Ks[1:])):
        mean = _mode_dot(mean, M, i)
    dk = np.asarray(K_ss_diag, dtype=np.float64)
    for K in Ks[1:]:
        dk = np.multiply.outer(dk, np.diag(K))
    S2 = A                                              # (A * A^-1/2)^2
    evs = [(K_star @ np.linalg.inv(Ks[0]) @ eig[0][1]) ** 2] + [U ** 2 for _, U in eig[1:]]
    sp = S2
    for i, M in enumerate(evs):
        sp = _mode_dot(sp, M, i)
    return mean, dk + sp


def tensor_linear(x, vectors):
    """Tensor_linear.forward, gp_computation_pack.py:155-159: `y = mode_dot(x, vectors[i], i + 1)` is evaluated on the
    INPUT for every i, so the value returned is the last mode's product alone."""
    i = len(vectors) - 1
    V = np.asarray(vectors[i], dtype=np.float64)            # [h_i, l_i]
    return np.moveaxis(np.tensordot(np.asarray(x, dtype=np.float64), V, axes=([i + 1], [1])), -1, i + 1)


def cigp_forward(X, Y, Xs, kernel_fn, log_beta):
    """P1: cigp.forward, cigp_v10.py:24-48.  y_var is ignored; the noise scalar lands on EVERY entry."""
    K = kernel_fn(X, X)
    L = cholesky_lower(sigma_cigp(K, log_beta, None))
    kx = kernel_fn(X, Xs)
    V = solve_lower(L, kx)
    mean = kx.T @ cho_solve(L, np.asarray(Y, dtype=np.float64))
    var = kernel_fn(Xs, Xs) - V.T @ V + np.exp(-float(np.ravel(log_beta)[0]))
    return mean, var


def conditional_gaussian(Y, Sigma, K_s, K_ss):
    """P2: conditional_Gaussian(..., 'cholesky3'), gp_computation_pack.py:103-110."""
    L = cholesky_lower(Sigma)
    mu = K_s.T @ cho_solve(L, np.asarray(Y, dtype=np.float64))
    V = solve_lower(L, K_s)           # the reference forms L.inverse() @ K_s
    return mu, K_ss - V.T @ V


def gp_basic_forward(X, Y, Xs, kernel_fn, noise_variance, y_var=None):
    """GP_basic.forward('cholesky3'), gp_basic.py:40-92: returns (mu.squeeze(), cov)."""
    S = sigma_basic(kernel_fn(X, X), noise_variance, y_var)
    mu, cov = conditional_gaussian(Y, S, kernel_fn(X, Xs), kernel_fn(Xs, Xs))
    return np.squeeze(mu), cov


def cigp2023_forward(X, Y, Xs, length_scale, scale, exp_format, noise_value, x_var=0.0):
    """P3: 2023 CIGP.forward, base_gp/cigp.py:61-97 (diag variance expanded to [Nt, d])."""
    kf = lambda a, b: se_kernel_2023(a, b, length_scale, scale, exp_format)
    inv_noise = inv_noise_from_box(noise_value)
    L = cholesky_lower(sigma_2023(kf(X, X), inv_noise, 0.0))
    kx = kf(X, Xs)
    V = solve_lower(L, kx)
    u = kx.T @ cho_solve(L, np.asarray(Y, dtype=np.float64))
    vd = np.diag(kf(Xs, Xs)).reshape(-1, 1) - (V * V).sum(0).reshape(-1, 1) + inv_noise
    return u, np.broadcast_to(vd, u.shape) + x_var


# ----------------------------------------------------------------------------------------------------------
# synthetic workload + CPU baseline leg
# ----------------------------------------------------------------------------------------------------------
def synthetic_xy(n, D, d, seed=0):
    """Deterministic synthetic workload: X ~ U[0,1)^D, Y = sin(2 pi X w) + 0.1 randn, globally normalised
    the way FidelityFusion_Models/MF_data.py:26-28 normalises y (mean / unbiased std over all entries)."""
    rng = np.random.default_rng(seed)
    X = rng.random((n, D))
    W = rng.random((D, d))
    Y = np.sin(2.0 * np.pi * (X @ W)) + 0.1 * rng.standard_normal((n, d))
    Y = (Y - Y.mean()) / (Y.std(ddof=1) + 1e-10)
    return X, Y


def nlml_forward_ard(X, Y, length_scales, signal_variance, log_beta):
    """One forward of cigp.negative_log_likelihood (+LL), ARD kernel: what bench.py's cpu_baseline leg times."""
    K = ard_kernel(X, X, length_scales, signal_variance)
    nll, _, _ = nll_v1_from_sigma(sigma_cigp(K, log_beta), np.asarray(Y, dtype=np.float64))
    return -nll
