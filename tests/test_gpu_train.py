"""K Adam steps per library call (cigp_v10.train_many -> ffgp_train_raw) against the reference's own loop -- one
`optimizer.zero_grad(); loss = -gpr.negative_log_likelihood(x, y); loss.backward(); optimizer.step()` per iteration through the
drop-in modules and torch.optim.Adam (FidelityFusion_Models/ResGP.py:78-112) -- and against the reference-generated fixtures of that
loop (tests/golden/resgp_chain.npz, train_log_resgp.npz).  The trajectory must be the per-step path's to 1e-12 (relative; the Adam
arithmetic is torch's operation for operation, the likelihood and gradients are the same launches)."""
import copy

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    torch.set_default_dtype(torch.float64)
    yield
    torch.set_default_dtype(torch.float32)


DEV = "cuda:0"


def T(a):
    return torch.tensor(np.asarray(a), dtype=torch.float64, device=DEV)


def rel(a, b):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a, dtype=np.float64)
    b = b.detach().cpu().numpy() if isinstance(b, torch.Tensor) else np.asarray(b, dtype=np.float64)
    return float(np.abs(a.reshape(b.shape) - b).max() / max(np.abs(b).max(), 1e-300))


def reference_loop(models, xs, ys, steps, lr):
    """the reference's loop: a fresh Adam per model, the loss recorded before the update"""
    trace = np.zeros((len(models), steps))
    for f, (m, x, y) in enumerate(zip(models, xs, ys)):
        opt = torch.optim.Adam(m.parameters(), lr=lr)
        for k in range(steps):
            opt.zero_grad()
            loss = -m.negative_log_likelihood(x, y)
            loss.backward()
            opt.step()
            trace[f, k] = float(loss.detach())
    return trace


def make_models(shapes, seed, kinds=None):
    from fidelityfusion_amd import kernel
    from fidelityfusion_amd.cigp_v10 import cigp
    from oracle import gp_oracle as O
    rng = np.random.default_rng(seed)
    models, xs, ys = [], [], []
    for f, (n, D, d) in enumerate(shapes):
        kind = (kinds or ["ard"] * len(shapes))[f]
        if kind == "ard":
            k = kernel.ARDKernel(D)
            with torch.no_grad():
                k.length_scales.copy_(torch.tensor(rng.uniform(0.6, 1.6, D) * rng.choice([-1.0, 1.0], D)))
        elif kind == "se":
            k = kernel.SquaredExponentialKernel(0.3, 0.2)
        else:
            k = kernel.MaternKernel(D, nu=2.5)
        models.append(cigp(k, 0.7 + 0.1 * f).double().to(DEV))
        X, Y = O.synthetic_xy(n, D, d, seed=seed + f)
        xs.append(T(X))
        ys.append(T(Y))
    return models, xs, ys


def params_of(m):
    return [p.detach().cpu().numpy().copy() for p in m.parameters()]


@pytest.mark.parametrize("shapes,kinds", [
    ([(24, 3, 2)], None),                                   # the one-kernel path (n <= 40)
    ([(128, 4, 1)], None),                                  # one diagonal block
    ([(300, 5, 2)], ["matern"]),                            # the blocked path, in order
    ([(700, 2, 3)], ["se"]),                                # scalar log length scale (broadcast), blocked path
    ([(60, 3, 1), (128, 2, 2), (17, 4, 1)], None),          # three small models: ONE launch per step for all
    ([(60, 3, 1), (400, 2, 2), (90, 16, 1)], ["ard", "se", "ard"]),   # a mix: the two small ones in one call, the larger one in its own
    ([(300, 3, 1), (513, 2, 2), (250, 4, 1)], ["ard", "matern", "se"]),   # three larger models: side by side from host threads
])
def test_train_many_follows_the_reference_loop(shapes, kinds):
    from fidelityfusion_amd.cigp_v10 import train_many
    steps, lr = 25, 1e-2
    models, xs, ys = make_models(shapes, 11, kinds)
    twins = [copy.deepcopy(m) for m in models]
    trace, state = train_many(models, xs, ys, steps, lr=lr)
    ref = reference_loop(twins, xs, ys, steps, lr)
    assert trace.shape == (len(models), steps) and trace.is_cuda
    assert rel(trace, ref) < 1e-12, rel(trace, ref)
    for m, t in zip(models, twins):
        for a, b in zip(params_of(m), params_of(t)):
            assert rel(a, b) < 1e-12
    # the optimisers continue where they stopped: 25 + 15 steps in two calls = 40 steps of the loop
    trace2, state = train_many(models, xs, ys, 15, lr=lr, state=state)
    twins2 = [copy.deepcopy(m) for m in make_models(shapes, 11, kinds)[0]]
    ref40 = reference_loop(twins2, xs, ys, 40, lr)
    assert rel(torch.cat([trace, trace2], dim=1), ref40) < 1e-11
    for m, t in zip(models, twins2):
        for a, b in zip(params_of(m), params_of(t)):
            assert rel(a, b) < 1e-11
    # parameters were updated in place behind autograd's back: their version counters moved, so a cached posterior is rebuilt
    with torch.no_grad():
        mean, _ = models[0](xs[0], ys[0], xs[0][:5])
        mean_t, _ = twins2[0](xs[0], ys[0], xs[0][:5])
    assert rel(mean, mean_t) < 1e-9


@pytest.mark.parametrize("shapes,kinds,yvar", [
    ([(16, 2, 1)], None, False), ([(32, 5, 1)], None, False), ([(47, 3, 2)], ["matern"], False), ([(64, 16, 1)], None, True),
    ([(100, 5, 16)], ["se"], False), ([(128, 5, 1)], None, True), ([(113, 7, 3)], ["matern"], False),
    ([(64, 5, 1)] * 16, None, False), ([(128, 4, 2), (20, 3, 1), (77, 2, 5)], ["ard", "se", "matern"], True),
])
def test_one_launch_trainer_is_the_launch_per_stage_trainer(shapes, kinds, yvar):
    """round 6: models of up to 128 points train inside ONE persistent kernel launch (csrc/train.hip: one workgroup per model, every Adam
    step inside the kernel -- Sigma in LDS, blocked Cholesky + inverse + Sigma^-1 on the matrix cores, gradient sums, links, Adam).  The
    launch-per-stage path it replaces (option train_persist = 0; itself held to torch.optim.Adam's loop above) must give the same
    trajectory: losses and final parameters to 1e-11, Adam moments carried over a second call"""
    from fidelityfusion_amd import _lib
    from fidelityfusion_amd.cigp_v10 import train_many
    steps, lr = 30, 2e-2
    runs = {}
    for persist in (1, 0):
        models, xs, ys = make_models(shapes, 23, kinds)
        if yvar:
            rng = np.random.default_rng(5)
            ys = [[y, torch.diag(T(rng.uniform(0.01, 0.3, y.shape[0])))] for y in ys]
        _lib.set_option("train_persist", persist, 0)
        try:
            tr1, state = train_many(models, xs, ys, steps, lr=lr)
            tr2, _ = train_many(models, xs, ys, 7, lr=lr, state=state)
        finally:
            _lib.set_option("train_persist", 1, 0)
        runs[persist] = (torch.cat([tr1, tr2], dim=1).clone(), [params_of(m) for m in models])
    assert torch.isfinite(runs[1][0]).all()
    assert rel(runs[1][0], runs[0][0]) < 1e-11, rel(runs[1][0], runs[0][0])
    for pa, pb in zip(runs[1][1], runs[0][1]):
        for a, b in zip(pa, pb):
            assert rel(a, b) < 1e-10, (a, b)


def test_one_launch_trainer_stops_only_the_model_that_failed():
    """a Sigma that is not positive definite stops THAT model (LinAlgError from the call, its parameters as they were); the other
    models of the launch have trained on: their parameters are those of training them alone"""
    from fidelityfusion_amd.cigp_v10 import train_many
    models, xs, ys = make_models([(50, 2, 1), (90, 3, 1), (128, 2, 2)], 3)
    solo, sx, sy = make_models([(50, 2, 1), (90, 3, 1), (128, 2, 2)], 3)
    before = params_of(models[1])
    bad = [ys[1], -3.0 * torch.eye(90, device=DEV, dtype=torch.float64)]
    with pytest.raises(torch.linalg.LinAlgError):
        train_many(models, xs, [ys[0], bad, ys[2]], 6)
    for a, b in zip(params_of(models[1]), before):
        assert np.array_equal(a, b)
    for f in (0, 2):
        train_many([solo[f]], [sx[f]], [sy[f]], 6)
        for a, b in zip(params_of(models[f]), params_of(solo[f])):
            assert rel(a, b) < 1e-12


@pytest.mark.parametrize("n,D,d", [(1, 1, 1), (2, 3, 1), (15, 2, 2), (16, 16, 16), (17, 1, 3), (31, 8, 1), (33, 9, 5), (64, 8, 16), (127, 16, 16),
                                    (128, 16, 16), (128, 1, 1)])
def test_one_workgroup_kernel_edge_shapes_against_the_oracle(n, D, d):
    """csrc/train.hip at the corners of what it accepts -- one point, one stage and one row more, the D <= 8 / <= 16 instantiations at
    their limits, 16 target columns, the full 128 x 128 image -- (i) in evaluate mode through the drop-in module: value and every gradient
    `loss.backward()` leaves against the numpy oracle of GaussianProcess/cigp_v10.py:50-69; (ii) as the trainer: three Adam steps
    against the reference's loop"""
    from fidelityfusion_amd import kernel
    from fidelityfusion_amd.cigp_v10 import cigp, train_many
    from oracle import gp_oracle as O
    rng = np.random.default_rng(1000 * n + 10 * D + d)
    X = rng.uniform(0, 1, (n, D))
    Y = rng.standard_normal((n, d))
    ls = rng.uniform(0.5, 1.5, D) * rng.choice([-1.0, 1.0], D)
    sv, lb = [-0.9], [0.4]
    k = kernel.ARDKernel(D)
    with torch.no_grad():
        k.length_scales.copy_(torch.tensor(ls))
        k.signal_variance.copy_(torch.tensor(sv))
    m = cigp(k, lb[0]).double().to(DEV)
    Xt, Yt = T(X), T(Y).requires_grad_(True)
    ll = m.negative_log_likelihood(Xt, Yt)
    ll.backward()
    ll_ref, g_ref = O.cigp_ll_and_grads(X, Y, ls, sv, lb)
    assert rel(ll.detach(), ll_ref) < 1e-10
    assert rel(k.length_scales.grad, g_ref["length_scales"]) < 1e-8
    assert rel(k.signal_variance.grad, g_ref["signal_variance"]) < 1e-8
    assert rel(m.log_beta.grad, g_ref["log_beta"]) < 1e-8
    assert rel(Yt.grad, g_ref["Y"]) < 1e-8
    twin = copy.deepcopy(m)
    for q in list(m.parameters()) + list(twin.parameters()):
        q.grad = None
    trace, _ = train_many([m], [Xt], [Yt.detach()], 3, lr=1e-2)
    ref = reference_loop([twin], [Xt], [Yt.detach()], 3, 1e-2)
    assert rel(trace, ref) < 1e-11
    for a, b in zip(params_of(m), params_of(twin)):
        assert rel(a, b) < 1e-11


def test_train_many_on_the_reference_fixture(golden):
    """tests/golden/resgp_chain.npz: the reference's train_ResGP on a seeded two-fidelity problem, 5 Adam steps per fidelity
    (FidelityFusion_Models/ResGP.py:67-112; the second fidelity's targets come with a y_var matrix) -- both fidelities as ONE
    train_many call"""
    from fidelityfusion_amd import kernel
    from fidelityfusion_amd.cigp_v10 import cigp, train_many
    g = golden("resgp_chain")
    gprs = [cigp(kernel.SquaredExponentialKernel(), 1.0).double().to(DEV) for _ in range(2)]
    xs = [T(g["x0n"]), T(g["x_res"])]
    ys = [T(g["y0n"]), [T(g["y_res_mean"]), T(g["y_res_var"])]]
    trace, _ = train_many(gprs, xs, ys, 5, lr=1e-2)
    assert rel(-trace.reshape(-1), g["ll_trace"]) < 1e-8
    for f in range(2):
        assert rel(gprs[f].log_beta, g[f"gpr_list__{f}__log_beta"]) < 1e-8
        assert rel(gprs[f].kernel.length_scale, g[f"gpr_list__{f}__kernel__length_scale"]) < 1e-8
        assert rel(gprs[f].kernel.signal_variance, g[f"gpr_list__{f}__kernel__signal_variance"]) < 1e-8


def test_train_many_reaches_the_reference_log(golden):
    """the reference's own committed log (FidelityFusion_Models/log/ResGP/train.log:201,401): 199 and 200 Adam steps on fidelity 0
    of the demo land on the logged parameters to the log's reproducibility (+-2e-3, tests/test_gpu_parity.py holds the per-step
    path to the same bar) -- here as two train_many calls"""
    from fidelityfusion_amd import kernel
    from fidelityfusion_amd.cigp_v10 import cigp, train_many
    g = golden("train_log_resgp")
    m = cigp(kernel.SquaredExponentialKernel(), 1.0).double().to(DEV)
    x, y = T(g["x0n"]), T(g["y0n"])
    _, state = train_many([m], [x], [y], 199, lr=1e-2)
    seen199 = [float(m.log_beta), float(m.kernel.length_scale), float(m.kernel.signal_variance)]
    train_many([m], [x], [y], 1, lr=1e-2, state=state)
    seen200 = [float(m.log_beta), float(m.kernel.length_scale), float(m.kernel.signal_variance)]
    assert np.abs(np.array(seen199) - g["line201"]).max() <= 2e-3
    assert np.abs(np.array(seen200) - g["line401"][:3]).max() <= 2e-3


def test_train_many_reports_a_matrix_that_is_not_positive_definite():
    from fidelityfusion_amd.cigp_v10 import train_many
    models, xs, ys = make_models([(50, 2, 1), (90, 3, 1)], 3)
    before = params_of(models[1])
    bad = [ys[1], -3.0 * torch.eye(90, device=DEV, dtype=torch.float64)]
    with pytest.raises(torch.linalg.LinAlgError):
        train_many(models, xs, [ys[0], bad], 4)
    # no parameter of the failing call moved past the step that failed (here: the first), and the handle is usable again
    for a, b in zip(params_of(models[1]), before):
        assert np.array_equal(a, b)
    trace, _ = train_many(models, xs, ys, 3)
    assert torch.isfinite(trace).all()


def test_train_many_runs_the_reference_loop_for_models_it_cannot_fuse():
    """CPU-resident fp32 models (the reference's default) are not eligible for the fused call: train_many then IS the reference's
    loop through the drop-in modules, same return values"""
    from fidelityfusion_amd import kernel
    from fidelityfusion_amd.cigp_v10 import cigp, train_many
    torch.set_default_dtype(torch.float32)
    try:
        rng = np.random.default_rng(0)
        x = torch.tensor(rng.uniform(0, 1, (40, 2)), dtype=torch.float32)
        y = torch.tensor(np.sin(4 * rng.uniform(0, 1, (40, 1))), dtype=torch.float32)
        m = cigp(kernel.ARDKernel(2), 1.0)
        t = copy.deepcopy(m)
        trace, state = train_many([m], [x], [y], 6, lr=1e-2)
        ref = reference_loop([t], [x], [y], 6, 1e-2)
    finally:
        torch.set_default_dtype(torch.float64)
    assert state["fused"] is False and rel(trace, ref) < 1e-5
