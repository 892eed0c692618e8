"""CPU: pin the oracle (oracle/gp_oracle.py) against vectors captured from the imported reference
(tests/golden/gen_goldens.py).  fp64 tolerance 1e-10 relative unless a fixture documents otherwise."""
import numpy as np
import pytest

from oracle import gp_oracle as O

RTOL = 1e-10


def close(a, b, rtol=RTOL, atol=None):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    scale = max(np.abs(b).max(), 1e-300) if b.size else 1.0
    atol = rtol * scale if atol is None else atol
    np.testing.assert_allclose(a, b.reshape(a.shape), rtol=rtol, atol=atol)


@pytest.mark.parametrize("D", [1, 5, 16])
def test_k_ard(golden, D):
    g = golden(f"k_ard_D{D}")
    for a, b, key in (("x1", "x2", "K12"), ("x1", "x1", "K11"), ("xs", "xs", "Kss"), ("x1", "xs", "K1s")):
        close(O.ard_kernel(g[a], g[b], g["length_scales"], g["signal_variance"]), g[key], 1e-12)


@pytest.mark.parametrize("D", [1, 5, 16])
def test_k_se(golden, D):
    g = golden(f"k_se_D{D}")
    close(O.se_kernel(g["x1"], g["x2"], g["length_scale"], g["signal_variance"]), g["K12"], 1e-12)
    close(O.se_kernel(g["x1"], g["x1"], g["length_scale"], g["signal_variance"]), g["K11"], 1e-12)


@pytest.mark.parametrize("D", [1, 5, 16])
@pytest.mark.parametrize("fmt", ["lin", "exp"])
def test_k_se2023(golden, D, fmt):
    g = golden(f"k_se2023_D{D}_{fmt}")
    close(O.se_kernel_2023(g["x1"], g["x2"], g["length_scale"], g["scale"], bool(g["exp_format"])), g["K12"], 1e-12)
    close(O.se_kernel_2023(g["x1"], g["x1"], g["length_scale"], g["scale"], bool(g["exp_format"])), g["K11"], 1e-12)


CIGP_CASES = ["ard_d1", "ard_d7", "ard_d7_yvar", "se_d3", "se_d3_yvar", "ard_n64", "ard_n1"]


@pytest.mark.parametrize("tag", CIGP_CASES)
def test_nlml_v1_cigp(golden, tag):
    g = golden("nlml_v1_cigp_" + tag)
    kind = "ard" if "length_scales" in g else "se"
    ls = g["length_scales"] if kind == "ard" else g["length_scale"]
    ll, gr = O.cigp_ll_and_grads(g["X"], g["Y"], ls, g["signal_variance"], g["log_beta"], g.get("y_var"), kind)
    close(ll, g["ll"])
    close(gr["log_beta"], g["g_log_beta"], 1e-9)
    close(gr["Y"], g["g_Y"], 1e-9)
    close(gr["signal_variance"], g["g_signal_variance"], 1e-9)
    close(gr["length_scales" if kind == "ard" else "length_scale"],
          g["g_length_scales" if kind == "ard" else "g_length_scale"], 1e-9)
    # posterior P1
    if kind == "ard":
        kf = lambda a, b: O.ard_kernel(a, b, ls, g["signal_variance"])
    else:
        kf = lambda a, b: O.se_kernel(a, b, ls, g["signal_variance"])
    mean, var = O.cigp_forward(g["X"], g["Y"], g["Xs"], kf, g["log_beta"])
    close(mean, g["mean"], 1e-9)
    close(var, g["var"], 1e-9)


@pytest.mark.parametrize("tag", ["d1", "d5"])
def test_nlml_v1_pack(golden, tag):
    g = golden("nlml_v1_pack_" + tag)
    ll, gr = O.pack_ll_and_grads(g["X"], g["Y"], g["length_scales"], g["signal_variance"], g["log_beta"])
    close(ll, g["ll"])
    for k in ("log_beta", "Y", "signal_variance", "length_scales"):
        close(gr[k], g["g_" + k], 1e-9)


@pytest.mark.parametrize("tag", ["d1", "d4_yvar"])
def test_cigp2023(golden, tag):
    g = golden("nlml_v1_cigp2023_" + tag)
    fmt = bool(g["exp_format"])
    nll, gr = O.cigp2023_nll_and_grads(g["X"], g["Y"], g["length_scale"], g["scale"], fmt, g["noise_value"],
                                       float(g["y_var"]))
    # the noise box is hard-wired float32 in the reference (utils/gp_noise.py:17,19): exp() and pow(-1) run
    # in fp32 (torch's fp32 exp and numpy's may differ by 1 ulp = 6e-8 in the noise), hence the looser bars
    close(nll, g["nll"], 1e-6)
    # d nll/d noise = -tr(G)/noise is a difference of two O(N) terms; fp32 rounding of the noise moves it by ~1e-5
    close(gr["noise_value"], g["g_noise_value"], 1e-6, atol=5e-5)
    close(gr["length_scale"], g["g_length_scale"], 1e-6)
    close(gr["scale"], g["g_scale"], 1e-6)
    close(gr["Y"], g["g_Y"], 1e-6)
    u, vd = O.cigp2023_forward(g["X"], g["Y"], g["Xs"], g["length_scale"], g["scale"], fmt, g["noise_value"])
    close(u, g["u"], 1e-6)
    close(vd, g["var_diag"], 1e-6)


@pytest.mark.parametrize("tag", ["d1", "d6"])
def test_v2_and_conditional(golden, tag):
    g = golden("nlml_v2_" + tag)
    ll, g_cov, g_Y = O.ll_v2_grads(g["Y"], g["cov"])
    close(ll, g["ll"])
    assert tuple(int(v) for v in g["ll_shape"]) == ((1, 1) if g["Y"].shape[1] == 1 else ())
    close(g_cov, g["g_cov"], 1e-8)
    close(g_Y, g["g_Y"], 1e-9)
    mu, cov = O.conditional_gaussian(g["Y"], g["cov"], g["Ks"], g["Kss"])
    close(mu, g["mu"], 1e-9)
    close(cov, g["cond_cov"], 1e-9)


def test_kinv_alternatives(golden):
    """row L3: the alternative Kinv_methods, incl. the two torch_distribution_MN* branches (d == N only)"""
    g = golden("kinv_methods")
    for tag in ("d1", "d3"):
        for meth in ("cholesky1", "cholesky2", "direct"):
            close(O.ll_alt(g[f"{tag}_Y"], g[f"{tag}_cov"], meth), g[f"{tag}_{meth}_ll"], 1e-10)
    g = golden("kinv_mn")
    for meth in ("torch_distribution_MN1", "torch_distribution_MN2"):
        ll = O.ll_alt(g["Y"], g["cov"], meth)
        assert ll.shape == g[f"{meth}_ll"].shape
        close(ll, g[f"{meth}_ll"], 1e-12)
        # d(sum_i R_i ll_i)/dcov = -1/2 sum(R) cov^-1
        close(-0.5 * g[f"{meth}_R"].sum() * np.linalg.inv(g["cov"]), 0.5 * (g[f"{meth}_gcov"] + g[f"{meth}_gcov"].T), 1e-9)


def test_withmean_and_multitask(golden):
    """CIGP_withMean (cigp_withMean.py:44-62: conditional Gaussian / Sigma^-2 likelihood of the residual y - m(x), m a small
    MLP) and MultiTaskGP_cigp.CIGP (:20-50: diagonal covariance expanded; log-determinant counted once)"""
    g = golden("gp_withmean_multitask")
    X, Y, Xq = g["X"], g["Y"], g["Xq"]

    def mlp(x):
        h = x @ g["mean_func__0__weight"].T + g["mean_func__0__bias"]
        h = np.where(h > 0, h, 0.01 * h)                      # nn.LeakyReLU default slope
        return h @ g["mean_func__2__weight"].T + g["mean_func__2__bias"]
    kf = lambda a, b: O.ard_kernel(a, b, g["kernel__length_scales"], g["kernel__signal_variance"])
    S = O.sigma_basic(kf(X, X), g["noise_variance"])
    mu, cov = O.conditional_gaussian(Y - mlp(X), S, kf(X, Xq), kf(Xq, Xq))
    close(mu + mlp(Xq), g["mu"], 1e-9)
    close(cov, g["cov"], 1e-9)
    close(O.ll_v2(Y - mlp(X), S)[0], g["ll"], 1e-10)
    kf0 = lambda a, b: O.ard_kernel(a, b, np.ones(2), np.ones(1))   # the second model keeps the kernel's initial parameters
    S0 = O.sigma_basic(kf0(X, X), np.array([0.4]))
    for tag, Ym in (("d3", Y), ("d1", Y[:, :1])):
        mu, cov = O.conditional_gaussian(Ym, S0, kf0(X, Xq), kf0(Xq, Xq))
        close(np.squeeze(mu), g[f"mt_{tag}_mu"], 1e-9)
        close(np.broadcast_to(np.diag(cov)[:, None], mu.shape), g[f"mt_{tag}_cov"], 1e-9)
        ll, L, A = O.ll_v2(Ym, S0)
        d = Ym.shape[1]
        ll_once = ll + 0.5 * (d - 1) * (2.0 * np.log(np.diag(L)).sum() + len(X) * np.log(2.0 * np.pi))
        close(ll_once, g[f"mt_{tag}_ll"], 1e-10)


@pytest.mark.parametrize("tag", ["d1", "d6"])
def test_gp_basic(golden, tag):
    g = golden("gp_basic_" + tag)
    kf = lambda a, b: O.ard_kernel(a, b, g["length_scales"], g["signal_variance"])
    S = O.sigma_basic(kf(g["X"], g["X"]), g["noise_variance"])
    ll, _, _ = O.ll_v2(g["Y"], S)
    close(ll, g["ll"])
    mu, var = O.gp_basic_forward(g["X"], g["Y"], g["Xs"], kf, g["noise_variance"])
    close(mu, g["mu"], 1e-9)
    close(var, g["var"], 1e-9)
    S = O.sigma_basic(kf(g["X"], g["X"]), g["noise_variance"], g["y_var"])
    ll, _, _ = O.ll_v2(g["Y"], S)
    close(ll, g["ll_yvar"])
    mu, var = O.gp_basic_forward(g["X"], g["Y"], g["Xs"], kf, g["noise_variance"], g["y_var"])
    close(mu, g["mu_yvar"], 1e-9)
    close(var, g["var_yvar"], 1e-9)


def test_cigar_blocks_sum(golden):
    g = golden("cigar_blocks")
    tot = 0.0
    for f in range(int(g["F"])):
        ll, _ = O.cigp_ll_and_grads(g[f"X{f}"], g[f"Y{f}"], g[f"length_scales{f}"], g[f"signal_variance{f}"],
                                    g[f"log_beta{f}"])
        close(ll, g[f"ll{f}"])
        tot += ll
    close(tot, g["ll_sum"])


def test_unblocked_restatements_agree_with_lapack():
    rng = np.random.default_rng(3)
    B = rng.standard_normal((40, 40))
    S = B @ B.T + 40 * np.eye(40)
    close(O.cholesky_unblocked(S), O.cholesky_lower(S), 1e-12)
    Y = rng.standard_normal((40, 3))
    L = O.cholesky_lower(S)
    close(O.solve_lower_loops(L, Y), O.solve_lower(L, Y), 1e-12)
    with pytest.raises(np.linalg.LinAlgError):
        O.cholesky_unblocked(-S)
    with pytest.raises(np.linalg.LinAlgError):
        O.cholesky_lower(-S)


def test_pi_quirk_constant():
    # L1 uses 3.1415, not pi: the constant term differs by 0.5*N*d*log(pi/3.1415)
    X, Y = O.synthetic_xy(50, 3, 2, seed=1)
    K = O.ard_kernel(X, X, np.ones(3), 1.0)
    a, _, _ = O.nll_v1_from_sigma(O.sigma_cigp(K, 1.0), Y)
    b, _, _ = O.nll_v1_from_sigma(O.sigma_cigp(K, 1.0), Y, pi_const=np.pi)
    close(b - a, 0.5 * 50 * 2 * np.log(np.pi / 3.1415), 1e-9)


@pytest.mark.parametrize("nu", ["05", "15", "25"])
def test_matern(golden, nu):
    """SURVEY 8f 'next' row 2: MaternKernel (GaussianProcess/kernel.py:109-169) behind the same cigp likelihood"""
    g = golden("matern_nu" + nu)
    v, rho = float(g["nu"]), float(g["rho"])
    close(O.matern_kernel(g["X"], g["x2"], g["length_scales"], g["signal_variance"], v, rho), g["K12"], 1e-12)
    ll, gr = O.cigp_ll_and_grads(g["X"], g["Y"], g["length_scales"], g["signal_variance"], g["log_beta"], nu=v, rho=rho)
    close(ll, g["ll"])
    # nu = 0.5: phi'(s) ~ 1/sqrt(s) blows up on the diagonal, where torch.cdist's expanded form leaves rounding noise
    # (~4e-16 instead of 0) ABOVE its 1e-30 clamp; the reference's autograd then pushes ~1e-7 of pure rounding noise
    # into the length-scale gradient.  That noise is not reproducible arithmetic, hence the looser bar there.
    for k in ("log_beta", "Y", "signal_variance", "length_scales"):
        close(gr[k], g["g_" + k], 1e-6 if (nu == "05" and k == "length_scales") else 1e-8)
    kf = lambda a, b: O.matern_kernel(a, b, g["length_scales"], g["signal_variance"], v, rho)
    mean, var = O.cigp_forward(g["X"], g["Y"], g["Xs"], kf, g["log_beta"])
    close(mean, g["mean"], 1e-9)
    close(var, g["var"], 1e-9)


# ------------------------------------------------------------------ Linear / RQ / Sum / Product kernels
def _pg(g, prefix):
    """parameters / gradients stored as p__<module path> / g__<module path> by the generator"""
    return ({k[len("p__" + prefix):]: g[k] for k in g if k.startswith("p__" + prefix)},
            {k[len("g__" + prefix):]: g[k] for k in g if k.startswith("g__" + prefix)})


def _lin_part(p):
    return (lambda X: O.linear_kernel(X, X, p["length_scales"], p["signal_variance"], p["center"]),
            lambda X, Gw: O.linear_kernel_grads(X, p["length_scales"], p["signal_variance"], p["center"], Gw))


def _ard_part(p, nu=None):
    kf = (lambda X: O.ard_kernel(X, X, p["length_scales"], p["signal_variance"])) if nu is None else \
         (lambda X: O.matern_kernel(X, X, p["length_scales"], p["signal_variance"], nu, 1.0))
    return kf, lambda X, Gw: O.ard_kernel_grads(X, p["length_scales"], p["signal_variance"], Gw, nu=nu)


def _rq_part(p):
    return (lambda X: O.rq_kernel(X, X, p["length_scale"], p["signal_variance"], p["alpha"]),
            lambda X, Gw: O.rq_kernel_grads(X, p["length_scale"], p["signal_variance"], p["alpha"], Gw))


def test_k_linear_rq(golden):
    g = golden("k_linear_rq")
    close(O.linear_kernel(g["x1"], g["x2"], g["p__length_scales"], g["p__signal_variance"], g["p__center"]), g["K_lin"], 1e-12)
    close(O.rq_kernel(g["x1"], g["x2"], g["rq_length_scale"], g["rq_signal_variance"], g["rq_alpha"]), g["K_rq"], 1e-12)


def test_cigp_sum_linear_matern(golden):
    """cigp over SumKernel(LinearKernel, MaternKernel) -- the reference demos' kernel (cigp_v10.py:81,111,147)"""
    g = golden("cigp_sum_linear_matern")
    p1, g1 = _pg(g, "kernel__kernel1__")
    p2, g2 = _pg(g, "kernel__kernel2__")
    lb = g["p__log_beta"]
    ll, (o1, o2), dS, dY = O.composed_ll_and_grads(g["X"], g["Y"], [_lin_part(p1), _ard_part(p2, nu=2.5)], "sum",
                                                   lambda K: O.sigma_cigp(K, lb), lambda dS, K: 0.0)
    close(ll, g["ll"])
    close(dY, g["g_Y"], 1e-8)
    close(-np.exp(-lb[0]) * np.trace(dS), g["g__log_beta"], 1e-8)
    for k in g1:
        close(o1[k], g1[k], 1e-8)
    for k in g2:
        close(o2[k], g2[k], 1e-8)
    kf = lambda a, b: (O.linear_kernel(a, b, p1["length_scales"], p1["signal_variance"], p1["center"]) +
                       O.matern_kernel(a, b, p2["length_scales"], p2["signal_variance"], 2.5, 1.0))
    mean, var = O.cigp_forward(g["X"], g["Y"], g["Xs"], kf, lb)
    close(mean, g["mean"], 1e-8)
    close(var, g["var"], 1e-8)


def _se_part(p):
    return (lambda X: O.se_kernel(X, X, p["length_scale"], p["signal_variance"]),
            lambda X, Gw: O.se_kernel_grads(X, p["length_scale"], p["signal_variance"], Gw))


def _check_parts(g, prefixes, outs, tol=1e-8):
    for pre, o in zip(prefixes, outs):
        _, gr = _pg(g, pre)
        assert gr, pre
        for k in gr:
            close(o[k], gr[k], tol)


def test_nested_compositions(golden):
    """nested Sum / Product kernels (kernel.py:172-236 compose arbitrary modules): three leaves under cigp, four leaves balanced
    under gp_computation_pack (mean(K) jitter chain), four leaves as a right-deep chain under GP_basic (V2)"""
    g = golden("cigp_nested3")      # Sum(Product(ARD, RQ), Linear)
    pre = ["kernel__kernel1__kernel1__", "kernel__kernel1__kernel2__", "kernel__kernel2__"]
    ps = [_pg(g, q)[0] for q in pre]
    lb = g["p__log_beta"]
    expr = ("sum", ("prod", 0, 1), 2)
    ll, outs, dS, dY = O.composed_ll_and_grads(g["X"], g["Y"], [_ard_part(ps[0]), _rq_part(ps[1]), _lin_part(ps[2])], expr,
                                               lambda K: O.sigma_cigp(K, lb), lambda dS, K: 0.0)
    close(ll, g["ll"])
    close(dY, g["g_Y"], 1e-8)
    close(-np.exp(-lb[0]) * np.trace(dS), g["g__log_beta"], 1e-8)
    _check_parts(g, pre, outs)
    kf = lambda a, b: (O.ard_kernel(a, b, ps[0]["length_scales"], ps[0]["signal_variance"]) *
                       O.rq_kernel(a, b, ps[1]["length_scale"], ps[1]["signal_variance"], ps[1]["alpha"]) +
                       O.linear_kernel(a, b, ps[2]["length_scales"], ps[2]["signal_variance"], ps[2]["center"]))
    mean, var = O.cigp_forward(g["X"], g["Y"], g["Xs"], kf, lb)
    close(mean, g["mean"], 1e-8)
    close(var, g["var"], 1e-8)

    g = golden("pack_nested4_balanced")      # Sum(Product(ARD, Matern 1.5), Product(Linear, SE))
    pre = ["kernel1__kernel1__", "kernel1__kernel2__", "kernel2__kernel1__", "kernel2__kernel2__"]
    ps = [_pg(g, q)[0] for q in pre]
    lb, n = g["log_beta"], g["X"].shape[0]
    expr = ("sum", ("prod", 0, 1), ("prod", 2, 3))
    ll, outs, dS, dY = O.composed_ll_and_grads(
        g["X"], g["Y"], [_ard_part(ps[0]), _ard_part(ps[1], nu=1.5), _lin_part(ps[2]), _se_part(ps[3])], expr,
        lambda K: O.sigma_pack(K, lb), lambda dS, K: O.JITTER * np.trace(dS) / (n * n))
    close(ll, g["ll"])
    close(dY, g["g_Y"], 1e-8)
    close(-np.exp(-lb[0]) * np.trace(dS), g["g_log_beta"], 1e-8)
    _check_parts(g, pre, outs)

    g = golden("gpbasic_nested4_chain")      # Product(RQ, Sum(ARD, Sum(Linear, Matern 2.5)))
    pre = ["kernel__kernel1__", "kernel__kernel2__kernel1__", "kernel__kernel2__kernel2__kernel1__", "kernel__kernel2__kernel2__kernel2__"]
    ps = [_pg(g, q)[0] for q in pre]
    nv = g["p__noise_variance"]
    expr = ("prod", 0, ("sum", 1, ("sum", 2, 3)))
    ll, outs, dS, dY = O.composed_ll_and_grads(
        g["X"], g["Y"], [_rq_part(ps[0]), _ard_part(ps[1]), _lin_part(ps[2]), _ard_part(ps[3], nu=2.5)], expr,
        lambda K: O.sigma_basic(K, nv), lambda dS, K: 0.0, variant="v2")
    close(ll, g["ll"])
    close(dY, g["g_Y"], 1e-8)
    close(2.0 * nv[0] * np.trace(dS), g["g__noise_variance"], 1e-8)
    _check_parts(g, pre, outs)


def test_pair_input_gradients(golden):
    """gradients of SumKernel(LinearKernel, MaternKernel)(x1, x2) w.r.t. both inputs for an upstream dK"""
    g = golden("pair_sum_linear_matern_xgrad")
    p1, _ = _pg(g, "kernel1__")
    p2, _ = _pg(g, "kernel2__")
    K = (O.linear_kernel(g["x1"], g["x2"], p1["length_scales"], p1["signal_variance"], p1["center"]) +
         O.matern_kernel(g["x1"], g["x2"], p2["length_scales"], p2["signal_variance"], 2.5, 1.0))
    close(K, g["K"], 1e-12)
    a1, a2 = O.linear_input_grads(g["x1"], g["x2"], p1["length_scales"], p1["signal_variance"], p1["center"], g["R"])
    b1, b2 = O.ard_input_grads(g["x1"], g["x2"], p2["length_scales"], p2["signal_variance"], g["R"], nu=2.5)
    close(a1 + b1, g["g_x1"], 1e-9)
    close(a2 + b2, g["g_x2"], 1e-9)


def test_cigp_rq_yvar(golden):
    g = golden("cigp_rq_yvar")
    p, gr = _pg(g, "kernel__")
    lb = g["p__log_beta"]
    K = O.rq_kernel(g["X"], g["X"], p["length_scale"], p["signal_variance"], p["alpha"])
    nll, L, _ = O.nll_v1_from_sigma(O.sigma_cigp(K, lb, g["y_var"]), g["Y"])
    close(-nll, g["ll"])
    G, A = O._G_matrix(L, g["Y"], g["Y"].shape[1])
    close(-A, g["g_Y"], 1e-8)
    close(np.exp(-lb[0]) * np.trace(G), g["g__log_beta"], 1e-8)
    o = O.rq_kernel_grads(g["X"], p["length_scale"], p["signal_variance"], p["alpha"], -G)
    for k in gr:
        close(o[k], gr[k], 1e-8)
    kf = lambda a, b: O.rq_kernel(a, b, p["length_scale"], p["signal_variance"], p["alpha"])
    mean, var = O.cigp_forward(g["X"], g["Y"], g["Xs"], kf, lb)
    close(mean, g["mean"], 1e-8)
    close(var, g["var"], 1e-8)


def test_pack_prod_ard_rq(golden):
    """gp_computation_pack.negative_log_likelihood over ProductKernel(ARDKernel, RationalQuadraticKernel)"""
    g = golden("pack_prod_ard_rq")
    p1, g1 = _pg(g, "kernel1__")
    p2, g2 = _pg(g, "kernel2__")
    lb = g["log_beta"]
    n = g["X"].shape[0]
    ll, (o1, o2), dS, dY = O.composed_ll_and_grads(
        g["X"], g["Y"], [_ard_part(p1), _rq_part(p2)], "prod", lambda K: O.sigma_pack(K, lb),
        lambda dS, K: O.JITTER * np.trace(dS) / (n * n))       # Sigma's mean(K) jitter: every dK entry gets tr(dS)/n^2
    close(ll, g["ll"])
    close(dY, g["g_Y"], 1e-8)
    close(-np.exp(-lb[0]) * np.trace(dS), g["g_log_beta"], 1e-8)
    for k in g1:
        close(o1[k], g1[k], 1e-8)
    for k in g2:
        close(o2[k], g2[k], 1e-8)


def test_gpbasic_sum_linear_ard(golden):
    g = golden("gpbasic_sum_linear_ard")
    p1, g1 = _pg(g, "kernel__kernel1__")
    p2, g2 = _pg(g, "kernel__kernel2__")
    nv = g["p__noise_variance"]
    ll, (o1, o2), dS, dY = O.composed_ll_and_grads(g["X"], g["Y"], [_lin_part(p1), _ard_part(p2)], "sum",
                                                   lambda K: O.sigma_basic(K, nv), lambda dS, K: 0.0, variant="v2")
    close(ll, g["ll"])
    close(dY, g["g_Y"], 1e-8)
    close(2.0 * nv[0] * np.trace(dS), g["g__noise_variance"], 1e-8)
    for k in g1:
        close(o1[k], g1[k], 1e-8)
    for k in g2:
        close(o2[k], g2[k], 1e-8)
    kf = lambda a, b: (O.linear_kernel(a, b, p1["length_scales"], p1["signal_variance"], p1["center"]) +
                       O.ard_kernel(a, b, p2["length_scales"], p2["signal_variance"]))
    mu, var = O.gp_basic_forward(g["X"], g["Y"], g["Xs"], kf, nv)
    close(mu, g["mu"], 1e-8)
    close(var, g["var"], 1e-8)


@pytest.mark.parametrize("tag,nv", [("eq", 1), ("up", 1), ("two_mode", 2)])
def test_tensor_linear(golden, tag, nv):
    g = golden("tensor_linear")
    vs = [g[f"{tag}_v{i}"] for i in range(nv)]
    close(O.tensor_linear(g[f"{tag}_x"], vs), g[f"{tag}_y"], 1e-12)


# ------------------------------------------------------------------ posterior in the loop
def test_kernel_input_grads(golden):
    g = golden("kernel_input_grads")
    for tag, nu in (("ard", None), ("matern15", 1.5)):
        ls, sv = g[f"{tag}_p__length_scales"], g[f"{tag}_p__signal_variance"]
        g1, g2 = O.ard_input_grads(g[f"{tag}_x1"], g[f"{tag}_x2"], ls, sv, g[f"{tag}_R"], nu=nu)
        _, g2a = O.ard_input_grads(g[f"{tag}_x2"], g[f"{tag}_x2"], ls, sv, g[f"{tag}_Rs"], nu=nu)
        g1b, _ = O.ard_input_grads(g[f"{tag}_x2"], g[f"{tag}_x2"], ls, sv, g[f"{tag}_Rs"], nu=nu)
        close(g1, g[f"{tag}_gx1"], 1e-9)
        close(g2 + g2a + g1b, g[f"{tag}_gx2"], 1e-9)


def test_cigp_forward_grads(golden):
    """cigp.forward (cigp_v10.py:24-48) differentiated w.r.t. x_test and y: kernel input gradients chained through the
    closed-form conditional-Gaussian backward"""
    g = golden("cigp_forward_grads")
    ls, sv, lb = g["p__kernel__length_scales"], g["p__kernel__signal_variance"], g["p__log_beta"]
    kf = lambda a, b: O.ard_kernel(a, b, ls, sv)
    X, Y, xs = g["X"], g["Y"], g["xs"]
    mean, var = O.cigp_forward(X, Y, xs, kf, lb)
    close(mean, g["mean"], 1e-9)
    close(var, g["var"], 1e-9)
    S = O.sigma_cigp(kf(X, X), lb)
    dy, dS, dKs, dKss = O.conditional_gaussian_grads(Y, S, kf(X, xs), g["R1"], g["R2"])
    close(dy, g["g_Y"], 1e-8)
    _, gxs_a = O.ard_input_grads(X, xs, ls, sv, dKs)
    g1, g2 = O.ard_input_grads(xs, xs, ls, sv, dKss)
    close(gxs_a + g1 + g2, g["g_xs"], 1e-8)
    # log_beta: Sigma's diagonal and the noise added to every entry of var both carry exp(-log_beta)
    close(-np.exp(-lb[0]) * (np.trace(dS) + g["R2"].sum()), g["g__log_beta"], 1e-8)


def test_hogp_block(golden):
    """H1-H2: HOGP_simple (GAR's block): loss, cached A and g, posterior mean and the reference's variance expression"""
    g = golden("hogp_block")
    ls, sv = g["length_scales"], g["signal_variance"]
    kf = lambda a, b: O.ard_kernel(a, b, ls, sv)
    grids = [np.arange(d, dtype=np.float64).reshape(-1, 1) for d in g["Y"].shape[1:]]
    # the shared D-dimensional ARD kernel meets the 1-column grids through numpy/torch broadcasting of x / length_scales
    Ks = [kf(g["X"], g["X"])] + [kf(np.repeat(gr, len(ls), 1), np.repeat(gr, len(ls), 1)) for gr in grids]
    loss, A, gg, eig = O.hogp_ll(Ks, g["Y"], g["noise_variance"])
    close(loss, g["loss"], 1e-10)
    close(A, g["A"], 1e-10)
    close(gg, g["g"], 1e-7)
    for i in range(3):
        close(eig[i][0], g[f"eig{i}"], 1e-9, atol=1e-12)
    mean, var = O.hogp_forward(kf(g["Xt"], g["X"]), np.diag(kf(g["Xt"], g["Xt"])), Ks, A, gg, eig)
    close(mean, g["mean"], 1e-7)
    close(var, g["var"], 1e-6)


@pytest.mark.parametrize("name", ["nlml_v1_cigp_ard_d1", "nlml_v1_cigp_ard_d7", "nlml_v1_cigp_ard_n64", "nlml_v1_cigp_ard_n1"])
def test_torch_cpu_ref_matches_reference_fixture(golden, name):
    """oracle/torch_cpu_ref.py (the CPU baseline bench.py times on the GPU box's host) against the imported reference's
    LL and autograd gradients: the same torch operator sequence, so agreement is at rounding level."""
    import torch
    from oracle import torch_cpu_ref as T
    g = golden(name)
    t = lambda k: torch.tensor(g[k])
    ll, gr = T.cigp_ll_and_grads(t("X"), t("Y"), t("length_scales"), t("signal_variance"), t("log_beta"))
    assert abs(float(ll) - float(g["ll"])) <= 1e-12 * abs(float(g["ll"]))
    for k, v in gr.items():
        ref = g["g_" + k]
        assert np.abs(v.numpy() - ref).max() <= 1e-10 * max(np.abs(ref).max(), 1e-300), k
    r = T.time_cigp(t("X"), t("Y"), t("length_scales"), t("signal_variance"), t("log_beta"), repeats=1, budget_s=5.0)
    assert r["fwd_s"] > 0 and abs(r["ll"] - float(g["ll"])) <= 1e-12 * abs(float(g["ll"]))


def test_matern_scalar_kernel_oracle(golden):
    g = golden("k_matern_scalar")
    K = O.matern_scalar_kernel(g["x1"], g["x2"], g["length_scale"], g["signal_variance"], g["nu"])
    assert np.abs(K - g["K"]).max() <= 1e-13 * np.abs(g["K"]).max()
