#!/usr/bin/env python3
"""Generate golden input/output vectors by importing the *reference* (IceLab-X/FidelityFusion,
read-only at /root/reference) in THIS container.  The reference has no tests of its own
(SURVEY.md section 4), so parity is pinned by outputs of the reference itself run here.

Only data (inputs, hyper-parameters, expected outputs) is written to tests/golden/*.npz;
no reference source travels.  Run:

    PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg python tests/golden/gen_goldens.py

The GPU box has no /root/reference: tests there read only the committed .npz files.
"""
import os
import sys
import types

sys.dont_write_bytecode = True
os.environ.setdefault("MPLBACKEND", "Agg")

import numpy as np
import torch

REF = os.environ.get("FFGP_REFERENCE", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))


def _install_tensorly_stub():
    """tensorly is not installed here; the Cholesky hot path never calls it, but several reference
    modules import it at module top.  Provide mode_dot & friends with real mode-n-product semantics."""
    tl = types.ModuleType("tensorly")
    tenalg = types.ModuleType("tensorly.tenalg")

    def mode_dot(tensor, matrix, mode):
        # mode-n product: contracts tensor's axis `mode` with matrix's axis 1
        t = torch.movedim(tensor, mode, -1)
        r = t @ matrix.transpose(-1, -2) if matrix.dim() == 2 else t @ matrix
        return torch.movedim(r, -1, mode)

    def multi_mode_dot(tensor, matrices, modes=None, transpose=False):
        modes = list(range(len(matrices))) if modes is None else modes
        for m, mode in zip(matrices, modes):
            tensor = mode_dot(tensor, m.T if transpose else m, mode)
        return tensor

    def tucker_to_tensor(tucker, **kw):
        core, factors = tucker
        return multi_mode_dot(core, factors)

    # The stub is asserted, not trusted: the published definition of the mode-n product, written out as an
    # einsum per mode -- (T x_n M)[..., j, ...] = sum_i T[..., i, ...] M[j, i] -- must reproduce it on random tensors of orders 2-4,
    # for every mode, with rectangular factors; multi_mode_dot (all modes, a subset, transposed factors) and the Tucker
    # reconstruction (core x_0 U_0 x_1 U_1 ...) against their own einsums.
    def _self_check():
        g = torch.Generator().manual_seed(20240607)
        letters = "abcd"
        for order in (2, 3, 4):
            shape = [3, 4, 5, 2][:order]
            T_ = torch.randn(shape, generator=g, dtype=torch.float64)
            Ms = [torch.randn((shape[m] + 1 + m, shape[m]), generator=g, dtype=torch.float64) for m in range(order)]
            for m in range(order):
                sub = letters[:order]
                out = sub.replace(sub[m], "z")
                want = torch.einsum("%s,z%s->%s" % (sub, sub[m], out), T_, Ms[m])
                assert torch.allclose(mode_dot(T_, Ms[m], m), want, rtol=0, atol=1e-13), ("mode_dot", order, m)
            sub = letters[:order]
            expr = sub + "," + ",".join(letters[m].upper() + letters[m] for m in range(order)) + "->" + sub.upper()
            want = torch.einsum(expr, T_, *Ms)
            assert torch.allclose(multi_mode_dot(T_, Ms), want, rtol=0, atol=1e-12), ("multi_mode_dot", order)
            assert torch.allclose(tucker_to_tensor((T_, Ms)), want, rtol=0, atol=1e-12), ("tucker_to_tensor", order)
            if order >= 3:      # a subset of the modes, and transposed factors
                want = torch.einsum("%s,%s->%s" % (sub, "z" + sub[1], sub.replace(sub[1], "z")), T_, Ms[1])
                assert torch.allclose(multi_mode_dot(T_, [Ms[1]], modes=[1]), want, rtol=0, atol=1e-13)
                assert torch.allclose(multi_mode_dot(T_, [M.T.contiguous() for M in Ms], transpose=True), torch.einsum(expr, T_, *Ms),
                                      rtol=0, atol=1e-12)
    _self_check()

    tl.set_backend = lambda name: None
    tl.tenalg = tenalg
    tl.ones = lambda shape, **kw: torch.ones(shape)
    tl.tensor_to_vec = lambda t: t.reshape(-1)
    tl.tucker_to_tensor = tucker_to_tensor
    tenalg.mode_dot = mode_dot
    tenalg.multi_mode_dot = multi_mode_dot
    tl.tucker_tensor = types.ModuleType("tensorly.tucker_tensor")
    tl.tucker_tensor.tucker_to_tensor = tucker_to_tensor
    sys.modules["tensorly"] = tl
    sys.modules["tensorly.tenalg"] = tenalg
    sys.modules["tensorly.tucker_tensor"] = tl.tucker_tensor


def npy(t):
    if isinstance(t, torch.Tensor):
        return t.detach().cpu().numpy().astype(np.float64)
    return np.asarray(t, dtype=np.float64)


ONLY = [t for t in os.environ.get("FFGP_GOLDEN_ONLY", "").split(",") if t]   # regenerate just these fixtures


def save(name, **arrays):
    if ONLY and name not in ONLY:
        return
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **{k: npy(v) for k, v in arrays.items()})
    print("wrote", path, {k: np.shape(v) for k, v in arrays.items()})


def make_xy(g, n, D, d, scale=1.0):
    X = torch.rand(n, D, generator=g, dtype=torch.float64) * scale
    W = torch.rand(D, d, generator=g, dtype=torch.float64)
    Y = torch.sin(2 * np.pi * X @ W) + 0.1 * torch.randn(n, d, generator=g, dtype=torch.float64)
    return X, Y


def main():
    _install_tensorly_stub()
    sys.path.insert(0, REF)
    cwd = os.getcwd()
    os.chdir("/tmp")  # keep any stray relative output away from the reference and the repo
    torch.set_default_dtype(torch.float64)

    import GaussianProcess.kernel as rk
    import GaussianProcess.gp_computation_pack as rpack
    from GaussianProcess.cigp_v10 import cigp as RCIGP
    from GaussianProcess.gp_basic import GP_basic as RGPB
    from MFGP_ver2023May.base_gp.cigp import CIGP as RCIGP23
    from MFGP_ver2023May.kernel.SE_kernel import SE_kernel as RSE23

    g = torch.Generator().manual_seed(20260103)

    # ------------------------------------------------------------------ kernels K1-K3
    for D in (1, 5, 16):
        x1 = torch.rand(65, D, generator=g) * 3 - 1
        x2 = torch.rand(33, D, generator=g) * 3 - 1
        xs = torch.rand(17, D, generator=g)  # <=25 rows: cdist takes the direct (non-matmul) path
        ls = (torch.rand(D, generator=g) * 1.5 + 0.5) * torch.where(torch.rand(D, generator=g) > 0.5, 1.0, -1.0)
        sv = torch.tensor([-1.7])
        k = rk.ARDKernel(D)
        with torch.no_grad():
            k.length_scales.copy_(ls)
            k.signal_variance.copy_(sv)
            save(f"k_ard_D{D}", x1=x1, x2=x2, xs=xs, length_scales=ls, signal_variance=sv,
                 K12=k(x1, x2), K11=k(x1, x1), Kss=k(xs, xs), K1s=k(x1, xs))
        k = rk.SquaredExponentialKernel(length_scale=0.3, signal_variance=-0.2)
        with torch.no_grad():
            save(f"k_se_D{D}", x1=x1, x2=x2, length_scale=k.length_scale, signal_variance=k.signal_variance,
                 K12=k(x1, x2), K11=k(x1, x1))
        for fmt, lsv, scv in ((False, 1.3, 0.7), (True, 1.3, 0.7)):
            k = RSE23(fmt, lsv, scv)
            with torch.no_grad():
                save(f"k_se2023_D{D}_{'exp' if fmt else 'lin'}", x1=x1, x2=x2, length_scale=k.length_scale,
                     scale=k.scale, exp_format=float(fmt), K12=k(x1, x2), K11=k(x1, x1))

    # ------------------------------------------------------------------ NLML V1 (cigp_v10), grads
    for tag, n, D, d, use_yvar, kern in (
        ("ard_d1", 257, 5, 1, False, "ard"),
        ("ard_d7", 257, 5, 7, False, "ard"),
        ("ard_d7_yvar", 257, 5, 7, True, "ard"),
        ("se_d3", 130, 3, 3, False, "se"),
        ("se_d3_yvar", 130, 3, 3, True, "se"),
        ("ard_n64", 64, 2, 2, False, "ard"),
        ("ard_n1", 1, 2, 1, False, "ard"),
    ):
        X, Y = make_xy(g, n, D, d)
        Y = Y.clone().requires_grad_(True)
        if kern == "ard":
            k = rk.ARDKernel(D)
            with torch.no_grad():
                k.length_scales.copy_((torch.rand(D, generator=g) + 0.5) *
                                      torch.where(torch.rand(D, generator=g) > 0.3, 1.0, -1.0))
                k.signal_variance.copy_(torch.tensor([-0.9 if n > 1 else 1.2]))
        else:
            k = rk.SquaredExponentialKernel(length_scale=-0.4, signal_variance=0.25)
        m = RCIGP(k, log_beta=0.7)
        yv = None
        if use_yvar:
            B = torch.rand(n, n, generator=g)
            yv = (B @ B.T) / n * 0.05 + torch.diag(torch.rand(n, generator=g) * 0.3)
        ll = m.negative_log_likelihood(X, [Y, yv] if use_yvar else Y)
        ll.backward()
        arrs = dict(X=X, Y=Y, log_beta=m.log_beta, ll=ll, g_log_beta=m.log_beta.grad, g_Y=Y.grad,
                    g_signal_variance=k.signal_variance.grad, signal_variance=k.signal_variance)
        if kern == "ard":
            arrs.update(length_scales=k.length_scales, g_length_scales=k.length_scales.grad)
        else:
            arrs.update(length_scale=k.length_scale, g_length_scale=k.length_scale.grad)
        if use_yvar:
            arrs["y_var"] = yv
        # prediction P1 (y_var ignored, noise on all entries)
        Xs = torch.rand(19, D, generator=g)
        with torch.no_grad():
            mean, var = m(X, [Y.detach(), yv] if use_yvar else Y.detach(), Xs)
        arrs.update(Xs=Xs, mean=mean, var=var)
        save(f"nlml_v1_cigp_{tag}", **arrs)

    # ------------------------------------------------------------------ NLML V1 pack variant (mean-K jitter)
    for tag, n, D, d in (("d1", 200, 4, 1), ("d5", 129, 6, 5)):
        X, Y = make_xy(g, n, D, d)
        Y = Y.clone().requires_grad_(True)
        k = rk.ARDKernel(D)
        with torch.no_grad():
            k.length_scales.copy_(torch.rand(D, generator=g) + 0.4)
            k.signal_variance.copy_(torch.tensor([1.9]))
        log_beta = torch.tensor([0.3], requires_grad=True)
        ll = rpack.negative_log_likelihood(k, log_beta, X, Y)
        ll.backward()
        save(f"nlml_v1_pack_{tag}", X=X, Y=Y, log_beta=log_beta, ll=ll, g_log_beta=log_beta.grad, g_Y=Y.grad,
             length_scales=k.length_scales, g_length_scales=k.length_scales.grad,
             signal_variance=k.signal_variance, g_signal_variance=k.signal_variance.grad)

    # ------------------------------------------------------------------ 2023 CIGP (S4/L1/P3)
    for tag, n, D, d, y_var in (("d1", 150, 3, 1, 0.0), ("d4_yvar", 128, 5, 4, 0.01)):
        X, Y = make_xy(g, n, D, d)
        Y = Y.clone().requires_grad_(True)
        m = RCIGP23({"noise": {"init_value": 20.0, "format": "exp"},
                     "kernel": {"SE": {"noise_exp_format": True, "length_scale": 1.0, "scale": 1.0}}})
        with torch.no_grad():
            m.kernel.length_scale.copy_(torch.tensor(0.8))
            m.kernel.scale.copy_(torch.tensor(1.4))
        nll = m.compute_loss(X, Y, y_var=y_var)
        nll.backward()
        Xs = torch.rand(21, D, generator=g)
        u, vd = m.forward(Xs)
        save(f"nlml_v1_cigp2023_{tag}", X=X, Y=Y, y_var=y_var, noise_value=m.noise_box.value,
             exp_format=float(m.kernel.noise_exp_format is True),
             length_scale=m.kernel.length_scale, scale=m.kernel.scale, nll=nll,
             g_noise_value=m.noise_box.value.grad, g_length_scale=m.kernel.length_scale.grad,
             g_scale=m.kernel.scale.grad, g_Y=Y.grad, Xs=Xs, u=u, var_diag=vd)

    # ------------------------------------------------------------------ V2 (Sigma^-2 quirk), cond. Gaussian, GP_basic
    for tag, n, D, d in (("d1", 120, 3, 1), ("d6", 140, 4, 6)):
        X, Y = make_xy(g, n, D, d)
        k = rk.ARDKernel(D)
        with torch.no_grad():
            k.length_scales.copy_(torch.rand(D, generator=g) + 0.5)
            cov = k(X, X) + 0.2 * torch.eye(n)
        cov = cov.clone().requires_grad_(True)
        Yg = Y.clone().requires_grad_(True)
        ll = rpack.Gaussian_log_likelihood(Yg, cov, Kinv_method="cholesky3")
        ll.sum().backward()
        Xs = torch.rand(23, D, generator=g)
        with torch.no_grad():
            Ks, Kss = k(X, Xs), k(Xs, Xs)
            mu, c = rpack.conditional_Gaussian(Y, cov.detach(), Ks, Kss, Kinv_method="cholesky3")
        save(f"nlml_v2_{tag}", Y=Y, cov=cov, ll=ll, ll_shape=np.array(ll.shape, dtype=np.float64),
             g_cov=cov.grad, g_Y=Yg.grad, Ks=Ks, Kss=Kss, mu=mu, cond_cov=c)
        # GP_basic end-to-end ('cholesky3')
        gp = RGPB(k, noise_variance=0.45)
        Yb = Y.clone().requires_grad_(True)
        llb = gp.log_likelihood(X, Yb)
        llb.sum().backward()
        with torch.no_grad():
            mub, varb = gp.forward(X, Y, Xs)
        B = torch.rand(n, n, generator=g)
        yv = (B @ B.T) / n * 0.02
        with torch.no_grad():
            llv = gp.log_likelihood(X, [Y, yv])
            muv, varv = gp.forward(X, [Y, yv], Xs)
        save(f"gp_basic_{tag}", X=X, Y=Y, Xs=Xs, length_scales=k.length_scales, signal_variance=k.signal_variance,
             noise_variance=gp.noise_variance, ll=llb, g_noise_variance=gp.noise_variance.grad,
             g_length_scales=k.length_scales.grad, g_signal_variance=k.signal_variance.grad, g_Y=Yb.grad,
             mu=mub, var=varb, y_var=yv, ll_yvar=llv, mu_yvar=muv, var_yvar=varv)

    # ------------------------------------------------------------------ ResGP chain, 2024 API, 5 Adam steps/fidelity
    from FidelityFusion_Models.ResGP import ResGP as RResGP, train_ResGP
    from FidelityFusion_Models.MF_data import MultiFidelityDataManager
    torch.manual_seed(7)
    x_all = torch.rand(120, 2) * 4
    il = torch.sort(torch.randperm(120)[:80]).values
    ih = torch.sort(torch.randperm(120)[:50]).values
    xl, xh = x_all[il], x_all[ih]
    f = lambda x: torch.sin(x.sum(1, keepdim=True))
    yl = f(xl) - 0.4 * torch.sin(2 * xl[:, :1]) + 0.05 * torch.rand(80, 1)
    yh = f(xh) + 0.05 * torch.rand(50, 1)
    xt = torch.rand(15, 2) * 4
    data = [{"raw_fidelity_name": "0", "fidelity_indicator": 0, "X": xl, "Y": yl},
            {"raw_fidelity_name": "1", "fidelity_indicator": 1, "X": xh, "Y": yh}]
    mgr = MultiFidelityDataManager(data)
    model = RResGP(2, [rk.SquaredExponentialKernel() for _ in range(2)], if_nonsubset=True)
    losses = []
    _orig = RCIGP.negative_log_likelihood

    def _spy(self, x, y):
        r = _orig(self, x, y)
        losses.append(float(r))
        return r

    RCIGP.negative_log_likelihood = _spy
    import io, contextlib
    with contextlib.redirect_stdout(io.StringIO()):
        train_ResGP(model, mgr, max_iter=5, lr_init=1e-2, debugger=None)
    RCIGP.negative_log_likelihood = _orig
    with torch.no_grad():
        xtn = mgr.normalizelayer[1].normalize_x(xt)
        yp, vp = model(mgr, xtn)
    sd = {k.replace(".", "__"): v for k, v in model.state_dict().items()}
    x0n, y0n = mgr.get_data(0, normal=True)
    xr, yr = mgr.get_data_by_name("res-1")
    save("resgp_chain", xl=xl, yl=yl, xh=xh, yh=yh, xt=xt, xtn=xtn, x0n=x0n, y0n=y0n,
         x_res=xr, y_res_mean=yr[0], y_res_var=yr[1], ll_trace=np.array(losses), ypred=yp, var_pred=vp, **sd)

    # ------------------------------------------------------------------ CIGAR-style per-fidelity blocks (sharded sum)
    F, n, D, d = 4, 256, 4, 32
    blocks = {}
    total = 0.0
    for fi in range(F):
        X, Y = make_xy(g, n, D, d)
        k = rk.ARDKernel(D)
        with torch.no_grad():
            k.length_scales.copy_(torch.rand(D, generator=g) + 0.5)
        m = RCIGP(k, log_beta=1.0)
        ll = m.negative_log_likelihood(X, Y)
        total += float(ll)
        blocks.update({f"X{fi}": X, f"Y{fi}": Y, f"length_scales{fi}": k.length_scales,
                       f"signal_variance{fi}": k.signal_variance, f"log_beta{fi}": m.log_beta, f"ll{fi}": ll})
    save("cigar_blocks", F=float(F), ll_sum=total, **blocks)

    # ------------------------------------------------------------------ 2023 ResGP joint loss (config-1 plumbing)
    try:
        from MFGP_ver2023May.ResGP import ResGP as RResGP23
        xin = np.load(os.path.join(REF, "assets/MF_data/Poisson_data/input.npy"))
        xtr = torch.tensor(xin[:128], dtype=torch.float64)
        xev = torch.tensor(xin[128:160], dtype=torch.float64)
        gg = torch.Generator().manual_seed(5)
        W = torch.rand(xtr.shape[1], 8, generator=gg)
        f0 = lambda x: torch.sin(x @ W)
        ys = [f0(xtr) * 0.8 + 0.1, f0(xtr)]
        m23 = RResGP23({"fidelity_shapes": [(8,), (8,)]}) if False else None
        cfg = {"fidelity_shapes": [ys[0].shape[1:], ys[1].shape[1:]]}
        m23 = RResGP23(cfg)
        m23 = m23.double()
        opt = torch.optim.Adam(m23.parameters(), lr=0.01)
        tr = []
        for _ in range(10):
            opt.zero_grad()
            nll = m23.compute_loss(xtr, ys)
            tr.append(float(nll))
            nll.backward()
            opt.step()
        with torch.no_grad():
            pm, pv = m23(xev)
        sd = {k.replace(".", "__"): v for k, v in m23.state_dict().items()}
        save("resgp2023_demo", x_train=xtr, y0=ys[0], y1=ys[1], x_eval=xev, nll_trace=np.array(tr),
             pred_mean=pm, pred_var=pv, **sd)
    except Exception as e:  # noqa
        print("resgp2023_demo skipped:", repr(e))

    # ------------------------------------------------------------------ loose known answers from the reference's own log
    rows = {}
    try:
        with open(os.path.join(REF, "FidelityFusion_Models/log/ResGP/train.log")) as fh:
            lines = fh.readlines()
        import re
        for ln in (201, 401, 601):
            nums = [float(v) for v in re.findall(r"tensor\(\[?(-?\d+\.\d+)", lines[ln - 1])]
            rows[f"line{ln}"] = np.array(nums)
        # the inputs that log was produced from: the demo's own data recipe (FidelityFusion_Models/ResGP.py:117-133, seed 1,
        # default dtype fp32 as in the demo), fidelity 0 as the reference's data manager normalises it -- so that a test
        # can re-run the 200 Adam steps of fidelity 0 on the drop-in and land on the logged parameters
        _dt = torch.get_default_dtype()
        torch.set_default_dtype(torch.float32)
        try:
            from FidelityFusion_Models.MF_data import MultiFidelityDataManager as _Mgr
            torch.manual_seed(1)
            x_all = torch.rand(500, 1) * 20
            x_low = x_all[torch.sort(torch.randperm(500)[:300]).values]
            x_h1 = x_all[torch.sort(torch.randperm(500)[:300]).values]
            x_h2 = x_all[torch.sort(torch.randperm(500)[:250]).values]
            y_low = torch.sin(x_low) - 0.5 * torch.sin(2 * x_low) + torch.rand(300, 1) * 0.1 - 0.05
            y_h1 = torch.sin(x_h1) - 0.3 * torch.sin(2 * x_h1) + torch.rand(300, 1) * 0.1 - 0.05
            y_h2 = torch.sin(x_h2) + torch.rand(250, 1) * 0.1 - 0.05
            mgr_log = _Mgr([{"raw_fidelity_name": "0", "fidelity_indicator": 0, "X": x_low, "Y": y_low},
                            {"raw_fidelity_name": "1", "fidelity_indicator": 1, "X": x_h1, "Y": y_h1},
                            {"raw_fidelity_name": "2", "fidelity_indicator": 2, "X": x_h2, "Y": y_h2}])
            x0n, y0n = mgr_log.get_data(0, normal=True)
            rows["x0n"], rows["y0n"] = x0n, y0n
        finally:
            torch.set_default_dtype(_dt)
        save("train_log_resgp", **rows)
    except Exception as e:  # noqa
        print("train_log_resgp skipped:", repr(e))


    # ------------------------------------------------------------------ Matern (SURVEY 8f "next" row 2): K, LL, grads, posterior
    g2 = torch.Generator().manual_seed(777)   # own stream of random numbers: added after the fixtures above were frozen
    for nu in (0.5, 1.5, 2.5):
        n, D, d = 150, 4, 3
        X, Y = make_xy(g2, n, D, d)
        Y = Y.clone().requires_grad_(True)
        k = rk.MaternKernel(D, nu=nu, rho=1.3)
        with torch.no_grad():
            k.length_scales.copy_((torch.rand(D, generator=g2) + 0.5) *
                                  torch.where(torch.rand(D, generator=g2) > 0.3, 1.0, -1.0))
            k.signal_variance.copy_(torch.tensor([-1.1]))
        m = RCIGP(k, log_beta=0.9)
        ll = m.negative_log_likelihood(X, Y)
        ll.backward()
        Xs = torch.rand(19, D, generator=g2)
        x2 = torch.rand(40, D, generator=g2)
        with torch.no_grad():
            mean, var = m(X, Y.detach(), Xs)
            K12 = k(X, x2)
        save(f"matern_nu{str(nu).replace('.', '')}", X=X, Y=Y, x2=x2, K12=K12, nu=nu, rho=1.3, log_beta=m.log_beta, ll=ll,
             g_log_beta=m.log_beta.grad, g_Y=Y.grad, length_scales=k.length_scales, g_length_scales=k.length_scales.grad,
             signal_variance=k.signal_variance, g_signal_variance=k.signal_variance.grad, Xs=Xs, mean=mean, var=var)

    # ------------------------------------------------------------------ Linear / RQ / Sum / Product kernels
    g3 = torch.Generator().manual_seed(888)   # own stream again: older fixtures stay bit-reproducible

    def rnd(*shape, lo=0.5, hi=1.5):
        return torch.rand(*shape, generator=g3) * (hi - lo) + lo

    def grads_of(mod):
        return {"g__" + n.replace(".", "__"): p.grad for n, p in mod.named_parameters()}

    def params_of(mod):
        return {"p__" + n.replace(".", "__"): p for n, p in mod.named_parameters()}

    D = 3
    x1, x2 = torch.rand(33, D, generator=g3) * 2 - 0.5, torch.rand(21, D, generator=g3) * 2 - 0.5
    lin = rk.LinearKernel(D)
    rq = rk.RationalQuadraticKernel(length_scale=0.8, signal_variance=-1.2, alpha=1.7)
    with torch.no_grad():
        lin.length_scales.copy_(rnd(D) * torch.tensor([1.0, -1.0, 1.0]))
        lin.center.copy_(rnd(D, lo=-0.3, hi=0.3))
        lin.signal_variance.copy_(torch.tensor([-0.7]))
    with torch.no_grad():
        save("k_linear_rq", x1=x1, x2=x2, K_lin=lin(x1, x2), K_rq=rq(x1, x2), **params_of(lin),
             rq_length_scale=rq.length_scale, rq_signal_variance=rq.signal_variance, rq_alpha=rq.alpha)

    # cigp with the reference demos' own kernel: SumKernel(LinearKernel, MaternKernel)  (cigp_v10.py:81,111,147)
    n, d = 120, 2
    X, Y = make_xy(g3, n, D, d)
    Y = Y.clone().requires_grad_(True)
    k = rk.SumKernel(rk.LinearKernel(D), rk.MaternKernel(D))
    with torch.no_grad():
        k.kernel1.length_scales.copy_(rnd(D, lo=1.0, hi=2.0))
        k.kernel1.center.copy_(rnd(D, lo=-0.2, hi=0.2))
        k.kernel1.signal_variance.copy_(torch.tensor([0.3]))
        k.kernel2.length_scales.copy_(rnd(D) * torch.tensor([-1.0, 1.0, 1.0]))
        k.kernel2.signal_variance.copy_(torch.tensor([1.4]))
    m = RCIGP(k, log_beta=1.2)
    ll = m.negative_log_likelihood(X, Y)
    ll.backward()
    Xs = torch.rand(17, D, generator=g3)
    with torch.no_grad():
        mean, var = m(X, Y.detach(), Xs)
    save("cigp_sum_linear_matern", X=X, Y=Y, Xs=Xs, ll=ll, g_Y=Y.grad, mean=mean, var=var, **params_of(m), **grads_of(m))

    # cigp with the rational-quadratic kernel (learnable alpha), y_var given
    X, Y = make_xy(g3, n, D, d)
    Y = Y.clone().requires_grad_(True)
    y_var = torch.diag(torch.rand(n, generator=g3) * 0.05) + 0.01 * torch.rand(n, n, generator=g3)  # only the diagonal counts
    m = RCIGP(rk.RationalQuadraticKernel(length_scale=0.6, signal_variance=1.3, alpha=0.9), log_beta=0.7)
    ll = m.negative_log_likelihood(X, [Y, y_var])
    ll.backward()
    with torch.no_grad():
        mean, var = m(X, [Y.detach(), y_var], Xs)
    save("cigp_rq_yvar", X=X, Y=Y, y_var=y_var, Xs=Xs, ll=ll, g_Y=Y.grad, mean=mean, var=var, **params_of(m), **grads_of(m))

    # gp_computation_pack.negative_log_likelihood with ProductKernel(ARDKernel, RationalQuadraticKernel)
    X, Y = make_xy(g3, n, D, 1)
    Y = Y.clone().requires_grad_(True)
    k = rk.ProductKernel(rk.ARDKernel(D), rk.RationalQuadraticKernel(length_scale=1.1, signal_variance=0.9, alpha=2.2))
    with torch.no_grad():
        k.kernel1.length_scales.copy_(rnd(D) * torch.tensor([1.0, 1.0, -1.0]))
        k.kernel1.signal_variance.copy_(torch.tensor([-1.6]))
    log_beta = torch.nn.Parameter(torch.tensor([1.5]))
    ll = rpack.negative_log_likelihood(k, log_beta, X, Y)
    ll.backward()
    save("pack_prod_ard_rq", X=X, Y=Y, ll=ll, g_Y=Y.grad, log_beta=log_beta, g_log_beta=log_beta.grad, **params_of(k),
         **grads_of(k))

    # GP_basic with SumKernel(LinearKernel, ARDKernel): V2 likelihood + conditional-Gaussian forward
    X, Y = make_xy(g3, n, D, d)
    Y = Y.clone().requires_grad_(True)
    k = rk.SumKernel(rk.LinearKernel(D), rk.ARDKernel(D))
    with torch.no_grad():
        k.kernel1.length_scales.copy_(rnd(D, lo=1.0, hi=2.0))
        k.kernel1.signal_variance.copy_(torch.tensor([0.5]))
        k.kernel2.length_scales.copy_(rnd(D))
    m = RGPB(k, noise_variance=0.4)
    ll = m.log_likelihood(X, Y)
    ll.sum().backward()
    with torch.no_grad():
        mu, var = m(X, Y.detach(), Xs)
    save("gpbasic_sum_linear_ard", X=X, Y=Y, Xs=Xs, ll=ll, g_Y=Y.grad, mu=mu, var=var, **params_of(m), **grads_of(m))

    # ------------------------------------------------------------------ Tensor_linear (gp_computation_pack.py:138-159) + CIGAR chain (X1, config 4 plumbing)
    g4 = torch.Generator().manual_seed(999)
    tl_cases = {"eq": ((6,), (6,)), "up": ((6,), (9,)), "two_mode": ((3, 4), (3, 6))}
    tl = {}
    for tag, (ls_, hs_) in tl_cases.items():
        mod = rpack.Tensor_linear(ls_, hs_)
        with torch.no_grad():
            for v in mod.vectors:
                v.add_(0.1 * torch.randn(v.shape, generator=g4))
        x = torch.randn(5, *ls_, generator=g4, requires_grad=True)
        y = mod(x)
        R = torch.randn(y.shape, generator=g4)
        (y * R).sum().backward()
        tl.update({f"{tag}_x": x, f"{tag}_y": y, f"{tag}_R": R, f"{tag}_gx": x.grad})
        for i, v in enumerate(mod.vectors):
            tl.update({f"{tag}_v{i}": v, f"{tag}_gv{i}": v.grad})
    save("tensor_linear", **tl)

    from FidelityFusion_Models.CIGAR import CIGAR as RCIGAR, train_CIGAR
    torch.manual_seed(11)
    dO, Dx = 12, 2
    pool = torch.rand(90, Dx) * 3
    Wm = torch.rand(Dx, dO)
    fgen = lambda x, a: torch.sin(x @ Wm * a) + 0.3 * a * torch.cos(x.sum(1, keepdim=True))
    # nested subsets: the data manager's partial-overlap fill reshapes to (-1, 1) and only works for d = 1
    # (MF_data.py:293); with x_2 in x_1 in x_0 it takes the "full subset" branch (:286-289)
    perm = torch.randperm(90)
    idx = [torch.sort(perm[:n_]).values for n_ in (60, 40, 30)]
    xs = [pool[i] for i in idx]
    ys = [fgen(xs[0], 0.8) + 0.02 * torch.rand(60, dO), fgen(xs[1], 0.9) + 0.02 * torch.rand(40, dO),
          fgen(xs[2], 1.0) + 0.02 * torch.rand(30, dO)]
    xt = torch.rand(8, Dx) * 3
    data = [{"raw_fidelity_name": str(i), "fidelity_indicator": i, "X": xs[i], "Y": ys[i]} for i in range(3)]
    mgr = MultiFidelityDataManager(data)
    model = RCIGAR(3, [rk.SquaredExponentialKernel() for _ in range(3)], [(dO,)] * 3, if_nonsubset=True)
    losses, fills = [], []
    _orig = RCIGP.negative_log_likelihood
    _orig_fill = MultiFidelityDataManager.get_nonsubset_fill_data

    def _spy(self, x, y):
        r = _orig(self, x, y)
        losses.append(float(r))
        return r

    def _spy_fill(self, mdl, f1, f2):
        r = _orig_fill(self, mdl, f1, f2)
        fills.append(r)
        return r

    RCIGP.negative_log_likelihood = _spy
    MultiFidelityDataManager.get_nonsubset_fill_data = _spy_fill
    with contextlib.redirect_stdout(io.StringIO()):
        train_CIGAR(model, mgr, max_iter=4, lr_init=1e-2, debugger=None)
    RCIGP.negative_log_likelihood = _orig
    MultiFidelityDataManager.get_nonsubset_fill_data = _orig_fill
    with torch.no_grad():
        xtn = mgr.normalizelayer[2].normalize_x(xt)
        yp, vp = model(mgr, xtn)
    sd = {k.replace(".", "__"): v for k, v in model.state_dict().items()}
    x0n, y0n = mgr.get_data(0, normal=True)
    extra = {}
    for fi, (sx, ylo, yhi) in enumerate(fills, start=1):
        extra.update({f"fill{fi}_x": sx, f"fill{fi}_ylow_mean": ylo[0], f"fill{fi}_ylow_var": ylo[1],
                      f"fill{fi}_yhigh_mean": yhi[0], f"fill{fi}_yhigh_var": yhi[1]})
        xr, yr = mgr.get_data_by_name(f"res-{fi}")
        extra.update({f"res{fi}_x": xr, f"res{fi}_mean": yr[0], f"res{fi}_var": yr[1]})
    save("cigar_chain", x0n=x0n, y0n=y0n, xtn=xtn, ll_trace=np.array(losses), ypred=yp, var_pred=vp, **extra, **sd)

    # ------------------------------------------------------------------ posterior in the loop: input gradients
    g5 = torch.Generator().manual_seed(4242)
    kin = {}
    D = 3
    kernels = {"ard": rk.ARDKernel(D), "se": rk.SquaredExponentialKernel(0.3, 0.2), "matern15": rk.MaternKernel(D, nu=1.5),
               "rq": rk.RationalQuadraticKernel(0.9, 1.1, 1.4)}
    with torch.no_grad():
        kernels["ard"].length_scales.copy_(torch.tensor([0.7, -1.3, 0.9]))
        kernels["matern15"].length_scales.copy_(torch.tensor([1.2, 0.8, -0.6]))
    for tag, k in kernels.items():
        a = (torch.rand(23, D, generator=g5) * 2).requires_grad_(True)
        b = (torch.rand(17, D, generator=g5) * 2).requires_grad_(True)
        R = torch.randn(23, 17, generator=g5)
        Rs = torch.randn(17, 17, generator=g5)
        ((k(a, b) * R).sum() + (k(b, b) * Rs).sum()).backward()
        kin.update({f"{tag}_x1": a, f"{tag}_x2": b, f"{tag}_R": R, f"{tag}_Rs": Rs, f"{tag}_gx1": a.grad, f"{tag}_gx2": b.grad})
        kin.update({f"{tag}_p__{n_}": p_ for n_, p_ in k.named_parameters()})
    save("kernel_input_grads", **kin)

    # cigp.forward differentiated w.r.t. x_test / y / parameters (what an acquisition optimiser does, acq.py:48-58)
    n, d, nt = 70, 2, 9
    X, Y = make_xy(g5, n, D, d)
    Y = Y.clone().requires_grad_(True)
    k = rk.ARDKernel(D)
    with torch.no_grad():
        k.length_scales.copy_(torch.tensor([0.8, 1.1, -0.9]))
        k.signal_variance.copy_(torch.tensor([1.3]))
    m = RCIGP(k, log_beta=1.4)
    xs = torch.rand(nt, D, generator=g5).requires_grad_(True)
    R1, R2 = torch.randn(nt, d, generator=g5), torch.randn(nt, nt, generator=g5)
    mean, var = m(X, Y, xs)
    ((mean * R1).sum() + (var * R2).sum()).backward()
    save("cigp_forward_grads", X=X, Y=Y, xs=xs, R1=R1, R2=R2, mean=mean, var=var, g_xs=xs.grad, g_Y=Y.grad,
         **params_of(m), **grads_of(m))

    # Bayesian_optimization/cigp.py CIGP_withMean.forward (normalisers + conditional_Gaussian) and its x_test gradient
    try:
        from Bayesian_optimization.cigp import CIGP_withMean as RBO
        xtr = torch.rand(40, 2, generator=g5) * 5
        ytr = torch.hstack([torch.sin(xtr.sum(1, keepdim=True)), torch.cos(xtr[:, :1])]) + 0.1 * torch.randn(40, 2, generator=g5)
        kb = rk.ARDKernel(2)
        bo = RBO(2, 2, kb, 0.3)
        xq = (torch.rand(11, 2, generator=g5) * 5).requires_grad_(True)
        Rm, Rc = torch.randn(11, 2, generator=g5), torch.randn(11, 2, generator=g5)
        mu, cov = bo(xtr, ytr, xq)
        ((mu * Rm).sum() + (cov * Rc).sum()).backward()
        ll = bo.log_likelihood(xtr, ytr)
        save("bo_cigp_withmean", xtr=xtr, ytr=ytr, xq=xq, Rm=Rm, Rc=Rc, mu=mu, cov=cov, g_xq=xq.grad, ll=ll,
             noise_variance=bo.noise_variance, length_scales=kb.length_scales, signal_variance=kb.signal_variance)
    except Exception as e:  # noqa
        print("bo_cigp_withmean skipped:", repr(e))

    # ------------------------------------------------------------------ CAR: GP_basic blocks behind the MC fidelity kernel (last item)
    from FidelityFusion_Models.CAR_ContinuousAutoRegression import ContinuousAutoRegression as RCAR, train_CAR
    torch.manual_seed(21)
    pool = torch.rand(100, 1) * 10
    perm = torch.randperm(100)
    idx = [torch.sort(perm[:n_]).values for n_ in (60, 45, 30)]
    xs = [pool[i] for i in idx]
    ys = [torch.sin(xs[0]) - 0.5 * torch.sin(2 * xs[0]) + 0.05 * torch.rand(60, 1),
          torch.sin(xs[1]) - 0.3 * torch.sin(2 * xs[1]) + 0.05 * torch.rand(45, 1),
          torch.sin(xs[2]) + 0.05 * torch.rand(30, 1)]
    xt = torch.linspace(0, 10, 13).reshape(-1, 1)
    mgr = MultiFidelityDataManager([{"raw_fidelity_name": str(i), "fidelity_indicator": i, "X": xs[i], "Y": ys[i]} for i in range(3)])
    car = RCAR(3, [rk.ARDKernel(1) for _ in range(3)], b_init=1.0)
    lls, overlaps = [], []
    _orig_ll = RGPB.log_likelihood
    _orig_ov = MultiFidelityDataManager.get_overlap_input_data

    def _spy_ll(self, x, y, *a, **kw):
        r = _orig_ll(self, x, y, *a, **kw)
        lls.append(float(r))
        return r

    def _spy_ov(self, *a, **kw):
        r = _orig_ov(self, *a, **kw)
        overlaps.append(r)
        return r

    RGPB.log_likelihood = _spy_ll
    MultiFidelityDataManager.get_overlap_input_data = _spy_ov
    with contextlib.redirect_stdout(io.StringIO()):
        train_CAR(car, mgr, max_iter=4, lr_init=1e-2)
    RGPB.log_likelihood = _orig_ll
    MultiFidelityDataManager.get_overlap_input_data = _orig_ov
    with torch.no_grad():
        yp, vp = car(mgr, xt)
    extra = {}
    for fi, ov in enumerate(overlaps, start=1):
        extra.update({f"ov{fi}_ylow": ov[1], f"ov{fi}_x": ov[2], f"ov{fi}_yhigh": ov[3]})
        # what CAR.forward reads back: get_data(-i) NORMALISES the stored residual set with a Normalizer of its own
        # (MF_data.py:134-135,141-143), although the block was trained on the un-normalised one -- kept as is
        rx, ry = mgr.get_data(-fi)
        extra.update({f"res{fi}_x_fwd": rx, f"res{fi}_y_fwd": ry})
    x0, y0 = mgr.get_data(0)
    sd = {k.replace(".", "__"): v for k, v in car.state_dict().items()}
    save("car_chain", x0=x0, y0=y0, xt=xt, ll_trace=np.array(lls), ypred=yp, var_pred=vp, **extra, **sd)

    # ------------------------------------------------------------------ HOGP block (H1-H2; GAR's per-fidelity model, config 5)
    from FidelityFusion_Models.two_fidelity_models.hogp_simple import HOGP_simple as RHOGP
    g6 = torch.Generator().manual_seed(31337)
    n, Dx, d1, d2, nt = 48, 2, 5, 4, 7
    Xh = torch.rand(n, Dx, generator=g6) * 6
    Wh = torch.rand(Dx, d1 * d2, generator=g6)
    Yh = (torch.sin(Xh @ Wh) + 0.05 * torch.randn(n, d1 * d2, generator=g6)).reshape(n, d1, d2).requires_grad_(True)
    Xth = torch.rand(nt, Dx, generator=g6) * 6
    kh = rk.ARDKernel(Dx)
    with torch.no_grad():
        kh.length_scales.copy_(torch.tensor([0.9, 1.4]))
        kh.signal_variance.copy_(torch.tensor([1.2]))
    hm = RHOGP(kh, 0.8, [d1, d2]).double()   # the grid is built with .float() (hogp_simple.py:35)
    loss = hm.log_likelihood(Xh, Yh)
    loss.backward()
    with torch.no_grad():
        mu_h, var_h = hm.forward(Xh, Xth)
    save("hogp_block", X=Xh, Y=Yh, Xt=Xth, loss=loss, g_Y=Yh.grad, g_noise_variance=hm.noise_variance.grad,
         g_length_scales=kh.length_scales.grad, g_signal_variance=kh.signal_variance.grad,
         length_scales=kh.length_scales, signal_variance=kh.signal_variance, noise_variance=hm.noise_variance,
         A=hm.A, g=hm.g, mean=mu_h, var=var_h, eig0=hm.K_eigen[0].value, eig1=hm.K_eigen[1].value, eig2=hm.K_eigen[2].value)

    # ------------------------------------------------------------------ GAR chain (config 5 plumbing): HOGP blocks + Tensor_linear, 2 fidelities
    from FidelityFusion_Models.GAR import GAR as RGAR, train_GAR
    torch.manual_seed(33)
    oshape = (4, 3)
    pool = torch.rand(60, 2) * 4
    perm = torch.randperm(60)
    idx = [torch.sort(perm[:n_]).values for n_ in (40, 28)]
    xs = [pool[i] for i in idx]
    Wg = torch.rand(2, 12)
    fg = lambda x, a: (torch.sin(x @ Wg * a) + 0.2 * a * torch.cos(x.sum(1, keepdim=True))).reshape(-1, *oshape)
    ys = [fg(xs[0], 0.8) + 0.02 * torch.rand(40, *oshape), fg(xs[1], 1.0) + 0.02 * torch.rand(28, *oshape)]
    xt = torch.rand(6, 2) * 4
    mgr = MultiFidelityDataManager([{"raw_fidelity_name": str(i), "fidelity_indicator": i, "X": xs[i], "Y": ys[i]} for i in range(2)])
    gar = RGAR(2, [rk.SquaredExponentialKernel() for _ in range(2)], [oshape, oshape], if_nonsubset=True).double()
    hl, fills = [], []
    _orig_h = RHOGP.log_likelihood
    _orig_fill = MultiFidelityDataManager.get_nonsubset_fill_data

    def _spy_h(self, x, y):
        r = _orig_h(self, x, y)
        hl.append(float(r))
        return r

    def _spy_fill2(self, mdl, f1, f2):
        r = _orig_fill(self, mdl, f1, f2)
        fills.append(r)
        return r

    RHOGP.log_likelihood = _spy_h
    MultiFidelityDataManager.get_nonsubset_fill_data = _spy_fill2
    with contextlib.redirect_stdout(io.StringIO()):
        train_GAR(gar, mgr, max_iter=3, lr_init=1e-2, debugger=None)
    RHOGP.log_likelihood = _orig_h
    MultiFidelityDataManager.get_nonsubset_fill_data = _orig_fill
    with torch.no_grad():
        xtn = mgr.normalizelayer[1].normalize_x(xt)
        yp, vp = gar(mgr, xtn)
    x0n, y0n = mgr.get_data(0, normal=True)
    sx, ylo, yhi = fills[0]
    xr, _ = mgr.get_data_by_name("res-1")
    sd = {k.replace(".", "__"): v for k, v in gar.state_dict().items()}
    save("gar_chain", x0n=x0n, y0n=y0n, xtn=xtn, loss_trace=np.array(hl), ypred=yp, var_pred=vp, fill_x=sx,
         fill_ylow_mean=ylo[0], fill_ylow_var=ylo[1], fill_yhigh_mean=yhi[0], fill_yhigh_var=yhi[1], res_x=xr, **sd)

    # ------------------------------------------------------------------ L3: the other Kinv_methods of Gaussian_log_likelihood / conditional_Gaussian
    g7 = torch.Generator().manual_seed(2718)
    alt = {}
    for tag, d in (("d1", 1), ("d3", 3)):
        n = 40
        Xa, Ya = make_xy(g7, n, 2, d)
        ka = rk.ARDKernel(2)
        cov0 = (ka(Xa, Xa) + 0.3 * torch.eye(n)).detach()
        Ks = ka(Xa, torch.rand(6, 2, generator=g7)).detach()
        Kss = torch.eye(6)
        alt.update({f"{tag}_Y": Ya, f"{tag}_cov": cov0, f"{tag}_Ks": Ks, f"{tag}_Kss": Kss})
        for meth in ("cholesky1", "cholesky2", "direct"):
            yv = Ya.clone().requires_grad_(True)
            cv = cov0.clone().requires_grad_(True)
            ll = rpack.Gaussian_log_likelihood(yv, cv, Kinv_method=meth)
            R = torch.randn(ll.shape, generator=g7)
            (ll * R).sum().backward()
            alt.update({f"{tag}_{meth}_ll": ll, f"{tag}_{meth}_R": R, f"{tag}_{meth}_gY": yv.grad, f"{tag}_{meth}_gcov": cv.grad})
        for meth in ("cholesky1", "direct"):
            mu, cc = rpack.conditional_Gaussian(Ya, cov0, Ks, Kss, Kinv_method=meth)
            alt.update({f"{tag}_{meth}_mu": mu, f"{tag}_{meth}_ccov": cc})
    save("kinv_methods", **alt)

    # the two torch_distribution_MN* branches (:85-88; gp_basic.py:147-151): MultivariateNormal(loc=y, ...) reads y's LAST
    # axis as the event axis, so they only run when d == N, and log_prob(y) at loc = y is the normalising constant N times
    g8 = torch.Generator().manual_seed(3141)
    n = 12
    Xm, Ym = make_xy(g8, n, 2, n)
    km = rk.ARDKernel(2)
    covm = (km(Xm, Xm) + 0.2 * torch.eye(n)).detach()
    mn = {"X": Xm, "Y": Ym, "cov": covm}
    for meth in ("torch_distribution_MN1", "torch_distribution_MN2"):
        cv = covm.clone().requires_grad_(True)
        ll = rpack.Gaussian_log_likelihood(Ym, cv, Kinv_method=meth)
        R = torch.randn(ll.shape, generator=g8)
        (ll * R).sum().backward()
        mn.update({f"{meth}_ll": ll, f"{meth}_R": R, f"{meth}_gcov": cv.grad})
        gb = RGPB(rk.ARDKernel(2), noise_variance=0.5)
        llb = gb.log_likelihood(Xm, Ym, Kinv_method=meth)
        llb.sum().backward()
        mn.update({f"{meth}_basic_ll": llb, f"{meth}_basic_g_noise": gb.noise_variance.grad,
                   f"{meth}_basic_g_ls": gb.kernel.length_scales.grad, f"{meth}_basic_g_sv": gb.kernel.signal_variance.grad})
    save("kinv_mn", **mn)

    # ------------------------------------------------------------------ the two hand-written GP modules next to cigp_v10:
    # CIGP_withMean (cigp_withMean.py:29-64) and MultiTaskGP_cigp.CIGP (:14-50); both import their siblings by bare name
    sys.path.insert(0, os.path.join(REF, "GaussianProcess"))
    with contextlib.redirect_stdout(io.StringIO()):
        import cigp_withMean as rwm
        import MultiTaskGP_cigp as rmt
        import kernel as rk_bare
    g9 = torch.Generator().manual_seed(1618)
    Xw, Yw = make_xy(g9, 48, 2, 3)
    Xq = torch.rand(7, 2, generator=g9)
    torch.manual_seed(77)                                   # the mean MLP's initialisation
    mw = rwm.CIGP_withMean(2, 3, kernel=rk_bare.ARDKernel(2), noise_variance=0.6)
    sdw = {k.replace(".", "__"): v.clone() for k, v in mw.state_dict().items()}
    xq = Xq.clone().requires_grad_(True)
    Yg = Yw.clone().requires_grad_(True)
    mu, cov = mw(Xw, Yg, xq)
    R1, R2 = torch.randn(mu.shape, generator=g9), torch.randn(cov.shape, generator=g9)
    ((mu * R1).sum() + (cov * R2).sum()).backward()
    fw = {f"fwd_g_{k.replace('.', '__')}": p.grad.clone() for k, p in mw.named_parameters()}
    fw.update(fwd_g_xq=xq.grad.clone(), fwd_g_Y=Yg.grad.clone())
    for p_ in mw.parameters():
        p_.grad = None
    llw = mw.log_likelihood(Xw, Yw)
    llw.backward()
    lw = {f"ll_g_{k.replace('.', '__')}": p.grad.clone() for k, p in mw.named_parameters()}
    mm = rmt.CIGP(rk_bare.ARDKernel(2), noise_variance=0.4)
    out_mt = {}
    for tag, Ym in (("d3", Yw), ("d1", Yw[:, :1].contiguous())):
        for p_ in mm.parameters():
            p_.grad = None
        with torch.no_grad():
            mu_m, cov_m = mm(Xw, Ym, Xq)
        ll_m = mm.log_likelihood(Xw, Ym)
        ll_m.backward()
        out_mt.update({f"mt_{tag}_mu": mu_m, f"mt_{tag}_cov": cov_m, f"mt_{tag}_ll": ll_m,
                       f"mt_{tag}_g_noise": mm.noise_variance.grad.clone(), f"mt_{tag}_g_ls": mm.kernel.length_scales.grad.clone(),
                       f"mt_{tag}_g_sv": mm.kernel.signal_variance.grad.clone()})
    save("gp_withmean_multitask", X=Xw, Y=Yw, Xq=Xq, R1=R1, R2=R2, mu=mu, cov=cov, ll=llw, **sdw, **fw, **lw, **out_mt)

    # ------------------------------------------------------------------ AR and NAR chains (the remaining 2024 trainers on cigp)
    from FidelityFusion_Models.AR_autoRegression import AR as RAR, train_AR
    from FidelityFusion_Models.NAR import NAR as RNAR, train_NAR
    for tag, Model, trainer in (("ar", RAR, train_AR), ("nar", RNAR, train_NAR)):
        torch.manual_seed(41)
        x_all = torch.rand(110, 1) * 8
        il = torch.sort(torch.randperm(110)[:70]).values
        ih = torch.sort(torch.randperm(110)[:45]).values
        xl, xh = x_all[il], x_all[ih]
        yl = torch.sin(xl) - 0.4 * torch.sin(2 * xl) + 0.05 * torch.rand(70, 1)
        yh = torch.sin(xh) + 0.05 * torch.rand(45, 1)
        xt = torch.linspace(0, 8, 11).reshape(-1, 1)
        mgr = MultiFidelityDataManager([{"raw_fidelity_name": "0", "fidelity_indicator": 0, "X": xl, "Y": yl},
                                        {"raw_fidelity_name": "1", "fidelity_indicator": 1, "X": xh, "Y": yh}])
        kl = [rk.SquaredExponentialKernel(), rk.SquaredExponentialKernel()]
        model = Model(2, kl, if_nonsubset=True) if tag == "nar" else Model(2, kl, rho_init=1.0, if_nonsubset=True)
        losses, fills = [], []
        _orig = RCIGP.negative_log_likelihood
        _orig_fill = MultiFidelityDataManager.get_nonsubset_fill_data

        def _spy(self, x, y):
            r = _orig(self, x, y)
            losses.append(float(r))
            return r

        def _spy_fill3(self, mdl, f1, f2):
            r = _orig_fill(self, mdl, f1, f2)
            fills.append(r)
            return r

        RCIGP.negative_log_likelihood = _spy
        MultiFidelityDataManager.get_nonsubset_fill_data = _spy_fill3
        with contextlib.redirect_stdout(io.StringIO()):
            trainer(model, mgr, max_iter=5, lr_init=1e-2, debugger=None)
        RCIGP.negative_log_likelihood = _orig
        MultiFidelityDataManager.get_nonsubset_fill_data = _orig_fill
        with torch.no_grad():
            xtn = mgr.normalizelayer[1].normalize_x(xt)
            yp, vp = model(mgr, xtn)
        x0n, y0n = mgr.get_data(0, normal=True)
        sx, ylo, yhi = fills[0]
        sd = {k.replace(".", "__"): v for k, v in model.state_dict().items()}
        save(f"{tag}_chain", x0n=x0n, y0n=y0n, xtn=xtn, ll_trace=np.array(losses), ypred=yp, var_pred=vp, fill_x=sx,
             fill_ylow_mean=ylo[0], fill_ylow_var=ylo[1], fill_yhigh_mean=yhi[0], fill_yhigh_var=yhi[1], **sd)

    # ------------------------------------------------------------------ 2023-API HOGP (base_gp/hogp.py): loss, gradients, forward
    try:
        from MFGP_ver2023May.base_gp.hogp import HOGP as RHOGP23
        g8 = torch.Generator().manual_seed(777123)
        n, Dx, d1, d2, nt = 40, 2, 4, 3, 6
        X23 = torch.rand(n, Dx, generator=g8) * 5
        Y23 = torch.sin(X23 @ torch.rand(Dx, d1 * d2, generator=g8)).reshape(n, d1, d2) + 0.05 * torch.randn(n, d1, d2, generator=g8)
        Y23 = Y23.requires_grad_(True)
        Xt23 = torch.rand(nt, Dx, generator=g8) * 5
        h23 = RHOGP23({"fidelity_shapes": [d1, d2], "noise": {"init_value": 0.7, "format": "linear"}}).double()
        with torch.no_grad():
            for i, kk in enumerate(h23.kernel_list):
                kk.length_scale.fill_(0.8 + 0.3 * i)
                kk.scale.fill_(1.1 + 0.1 * i)
        loss23 = h23.compute_loss(X23, Y23, y_var=0.05)
        loss23.backward()
        mu23, var23 = h23.forward(Xt23)
        extra = {}
        for i, kk in enumerate(h23.kernel_list):
            extra.update({f"k{i}_length_scale": kk.length_scale, f"k{i}_scale": kk.scale,
                          f"g_k{i}_length_scale": kk.length_scale.grad, f"g_k{i}_scale": kk.scale.grad})
        save("hogp2023_block", X=X23, Y=Y23, Xt=Xt23, loss=loss23, g_Y=Y23.grad, noise=h23.noise_box.value,
             g_noise=h23.noise_box.value.grad, mean=mu23, var=var23, A=h23.A, **extra)
    except Exception as e:  # noqa
        print("hogp2023_block skipped:", repr(e))

    # ------------------------------------------------------------------ MaternKernel_scalarLengthScale (kernel.py:312-347): cross-covariance
    # between DISTINCT point sets (on coincident points the unclamped sqrt is NaN at rounding-dependent places), values and
    # autograd gradients w.r.t. length_scale, signal_variance, nu and both inputs
    g9 = torch.Generator().manual_seed(424242)
    xa = (torch.rand(41, 3, generator=g9) * 2).requires_grad_(True)
    xb = (torch.rand(29, 3, generator=g9) * 2 + 0.05).requires_grad_(True)
    km = rk.MaternKernel_scalarLengthScale(length_scale=1.3, signal_variance=-0.8, nu=1.7)
    Kms = km(xa, xb)
    Rm = torch.rand(41, 29, generator=g9)
    (Kms * Rm).sum().backward()
    save("k_matern_scalar", x1=xa, x2=xb, K=Kms, R=Rm, length_scale=km.length_scale, signal_variance=km.signal_variance, nu=km.nu,
         g_length_scale=km.length_scale.grad, g_signal_variance=km.signal_variance.grad, g_nu=km.nu.grad, g_x1=xa.grad, g_x2=xb.grad)

    # ------------------------------------------------------------------ nested Sum / Product compositions (kernel.py:172-236 compose
    # arbitrary modules) and input gradients through a composed kernel (the acquisition loops differentiate the posterior w.r.t.
    # the query points): three leaves under cigp, four leaves (balanced / chain) under gp_computation_pack and GP_basic
    g10 = torch.Generator().manual_seed(777001)

    def rnd10(*shape, lo=0.5, hi=1.5):
        return torch.rand(*shape, generator=g10) * (hi - lo) + lo

    def set_(p, v):
        with torch.no_grad():
            p.copy_(v if isinstance(v, torch.Tensor) else torch.tensor([v]))

    D, n, d = 3, 96, 2
    # (a) the demo pair SumKernel(LinearKernel, MaternKernel): kernel value and gradients w.r.t. BOTH inputs for an upstream dK
    kp = rk.SumKernel(rk.LinearKernel(D), rk.MaternKernel(D))
    set_(kp.kernel1.length_scales, rnd10(D, lo=1.0, hi=2.0))
    set_(kp.kernel1.center, rnd10(D, lo=-0.2, hi=0.2))
    set_(kp.kernel1.signal_variance, 0.4)
    set_(kp.kernel2.length_scales, rnd10(D) * torch.tensor([1.0, -1.0, 1.0]))
    set_(kp.kernel2.signal_variance, -1.3)
    xa = (torch.rand(37, D, generator=g10) * 2 - 0.5).requires_grad_(True)
    xb = (torch.rand(26, D, generator=g10) * 2 - 0.5).requires_grad_(True)
    Kp = kp(xa, xb)
    Rp = torch.rand(37, 26, generator=g10) - 0.3
    (Kp * Rp).sum().backward()
    save("pair_sum_linear_matern_xgrad", x1=xa, x2=xb, K=Kp, R=Rp, g_x1=xa.grad, g_x2=xb.grad, **params_of(kp), **grads_of(kp))

    # (b) three leaves: SumKernel(ProductKernel(ARDKernel, RationalQuadraticKernel), LinearKernel) under cigp -- likelihood, every
    # gradient, the posterior and the gradient of (sum(mean) + trace(var)) w.r.t. the query points
    k3 = rk.SumKernel(rk.ProductKernel(rk.ARDKernel(D), rk.RationalQuadraticKernel(length_scale=0.9, signal_variance=1.1, alpha=1.4)),
                      rk.LinearKernel(D))
    set_(k3.kernel1.kernel1.length_scales, rnd10(D) * torch.tensor([1.0, 1.0, -1.0]))
    set_(k3.kernel1.kernel1.signal_variance, -1.2)
    set_(k3.kernel2.length_scales, rnd10(D, lo=1.0, hi=2.0))
    set_(k3.kernel2.center, rnd10(D, lo=-0.2, hi=0.2))
    set_(k3.kernel2.signal_variance, 0.35)
    X, Y = make_xy(g10, n, D, d)
    Y = Y.clone().requires_grad_(True)
    m3 = RCIGP(k3, log_beta=1.1)
    ll = m3.negative_log_likelihood(X, Y)
    ll.backward()
    Xs = torch.rand(19, D, generator=g10).requires_grad_(True)
    mean, var = m3(X, Y.detach(), Xs)
    gXs, = torch.autograd.grad(mean.sum() + var.diagonal().sum(), Xs)
    save("cigp_nested3", X=X, Y=Y, Xs=Xs, ll=ll, g_Y=Y.grad, mean=mean, var=var, g_Xs=gXs, **params_of(m3), **grads_of(m3))

    # (c) four leaves, balanced: SumKernel(ProductKernel(ARD, Matern nu=1.5), ProductKernel(Linear, SquaredExponential)) under
    # gp_computation_pack.negative_log_likelihood (mean(K) jitter)
    k4 = rk.SumKernel(rk.ProductKernel(rk.ARDKernel(D), rk.MaternKernel(D, nu=1.5)),
                      rk.ProductKernel(rk.LinearKernel(D), rk.SquaredExponentialKernel(length_scale=0.2, signal_variance=-0.1)))
    set_(k4.kernel1.kernel1.length_scales, rnd10(D))
    set_(k4.kernel1.kernel1.signal_variance, 1.3)
    set_(k4.kernel1.kernel2.length_scales, rnd10(D, lo=1.0, hi=2.5) * torch.tensor([-1.0, 1.0, 1.0]))
    set_(k4.kernel1.kernel2.signal_variance, 0.8)
    set_(k4.kernel2.kernel1.length_scales, rnd10(D, lo=1.0, hi=2.0))
    set_(k4.kernel2.kernel1.center, rnd10(D, lo=-0.3, hi=0.3))
    set_(k4.kernel2.kernel1.signal_variance, -0.6)
    X, Y = make_xy(g10, n, D, 1)
    Y = Y.clone().requires_grad_(True)
    log_beta = torch.nn.Parameter(torch.tensor([1.3]))
    ll = rpack.negative_log_likelihood(k4, log_beta, X, Y)
    ll.backward()
    save("pack_nested4_balanced", X=X, Y=Y, ll=ll, g_Y=Y.grad, log_beta=log_beta, g_log_beta=log_beta.grad, **params_of(k4),
         **grads_of(k4))

    # (d) four leaves, chain with the deep operand on the RIGHT: ProductKernel(RQ, SumKernel(ARD, SumKernel(Linear, Matern nu=2.5)))
    # under GP_basic (V2 likelihood), plus the conditional-Gaussian forward
    kc = rk.ProductKernel(rk.RationalQuadraticKernel(length_scale=1.2, signal_variance=0.9, alpha=2.1),
                          rk.SumKernel(rk.ARDKernel(D), rk.SumKernel(rk.LinearKernel(D), rk.MaternKernel(D))))
    set_(kc.kernel2.kernel1.length_scales, rnd10(D))
    set_(kc.kernel2.kernel1.signal_variance, 0.7)
    set_(kc.kernel2.kernel2.kernel1.length_scales, rnd10(D, lo=1.0, hi=2.0))
    set_(kc.kernel2.kernel2.kernel1.signal_variance, 0.25)
    set_(kc.kernel2.kernel2.kernel2.length_scales, rnd10(D) * torch.tensor([1.0, -1.0, -1.0]))
    set_(kc.kernel2.kernel2.kernel2.signal_variance, 1.1)
    X, Y = make_xy(g10, n, D, d)
    Y = Y.clone().requires_grad_(True)
    mb = RGPB(kc, noise_variance=0.45)
    ll = mb.log_likelihood(X, Y)
    ll.sum().backward()
    Xs2 = torch.rand(15, D, generator=g10)
    with torch.no_grad():
        mu, var = mb(X, Y.detach(), Xs2)
    save("gpbasic_nested4_chain", X=X, Y=Y, Xs=Xs2, ll=ll, g_Y=Y.grad, mu=mu, var=var, **params_of(mb), **grads_of(mb))

    os.chdir(cwd)


if __name__ == "__main__":
    main()
