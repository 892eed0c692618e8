"""Multi-GPU tiled Cholesky of ONE block (fidelityfusion_amd/tiled.py): the distributed algorithm proven on
CPU -- world size 2 and 3 over gloo with the local operations injected (TorchOps) -- and, on the GPU box, the same code
with the shipped HIP operations (one rank, and two gloo ranks sharing the box's GPU)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _problem(n, D, d, seed=0):
    from oracle import gp_oracle as O
    X, Y = O.synthetic_xy(n, D, d, seed=seed)
    w = np.linspace(0.7, 1.4, D)
    return X, Y, w, 1.3, np.exp(-1.0) + 1e-6


def _reference(X, Y, w, amp, dadd):
    d2 = (((X * w)[:, None, :] - (X * w)[None, :, :]) ** 2).sum(-1)
    S = amp * np.exp(-0.5 * np.maximum(d2, 1e-30)) + dadd * np.eye(len(X))
    L = np.linalg.cholesky(S)
    G = np.linalg.solve(L, Y)
    nll = 0.5 * (G * G).sum() + Y.shape[1] * np.log(np.diag(L)).sum() + 0.5 * len(X) * Y.shape[1] * np.log(2 * 3.1415)
    return L, nll


def _worker(rank, world, port, n, nb, use_hip, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from fidelityfusion_amd import tiled
        from tiled_torch_ops import TorchOps
        X, Y, w, amp, dadd = _problem(n, 4, 3)
        ops = tiled.HipOps(torch.device("cuda", 0)) if use_hip else TorchOps()
        t = lambda a: torch.tensor(np.asarray(a, dtype=np.float64))
        tc = tiled.TiledCholesky(n, nb=nb, ops=ops)
        tc.assemble(t(X), t(w), t([amp]), dadd, clamp=1e-30)
        assert sorted(tc.panels) == tc.layout.owned(rank)            # a rank holds its own panels only
        rc = tc.factor()
        nll = tc.nll_v1(t(Y))
        L = tc.gather_dense_factor().numpy()
        # non-PD input: every rank reports the same first failing pivot
        bad = tiled.TiledCholesky(n, nb=nb, ops=ops)
        S = torch.eye(n, dtype=torch.float64)
        S[n // 2 + 5, n // 2 + 5] = -1.0
        bad.load_dense(S)
        rc_bad = bad.factor()
        if rank == 0:
            out.put((rc, nll, L, rc_bad, [tc.layout.bytes_per_rank(r) for r in range(world)]))
    finally:
        dist.destroy_process_group()


def _run(world, n, nb, use_hip=False):
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, nb, use_hip, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = q.get()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    return res


def _check(res, n, world):
    rc, nll, L, rc_bad, mem = res
    X, Y, w, amp, dadd = _problem(n, 4, 3)
    Lref, nll_ref = _reference(X, Y, w, amp, dadd)
    assert rc == 0
    assert np.abs(L - Lref).max() <= 1e-12 * np.abs(Lref).max()
    assert abs(nll - nll_ref) <= 1e-11 * abs(nll_ref)
    assert rc_bad == n // 2 + 6                                        # 1-based index of the first non-positive pivot
    assert sum(mem) == sum(8 * (n - k0) * min(nb_, n - k0) for nb_ in [_NB] for k0 in range(0, n, nb_))
    if n >= 4 * world * _NB // 2:
        assert max(mem) <= (1.0 / world + 0.25) * sum(mem)             # block-cyclic: no rank holds much more than its share


_NB = 128


@pytest.mark.parametrize("world,n", [(2, 1000), (3, 906), (2, 128), (1, 520)])
def test_tiled_cholesky_torch_ops_over_gloo(world, n):
    """the distributed algorithm (panel broadcast with look-ahead, local updates, reduce-to-owner substitution) on CPU:
    factor, likelihood and failing-pivot index against dense LAPACK; ragged last panel, n < nb * world, world = 1"""
    _check(_run(world, n, _NB), n, world)


def test_panel_layout_and_wire_budget():
    from fidelityfusion_amd.tiled import PanelLayout
    lay = PanelLayout(262144, 512, 8)
    assert lay.npanels == 512 and lay.owner(9) == 1 and lay.rows(3) == 262144 - 1536
    per = [lay.bytes_per_rank(r) for r in range(8)]
    assert max(per) < 36e9 and sum(per) == sum(8 * (262144 - k * 512) * 512 for k in range(512))   # 275 GB over 8 ranks
    with pytest.raises(ValueError):
        PanelLayout(100, 33, 2)


@pytest.mark.gpu
def test_tiled_cholesky_hip_ops_single_rank_matches_fused_path():
    """the tiled code on the shipped HIP operations (ffgp_assemble / ffgp_potrf_rows / ffgp_gemm / ffgp_trsm_lower), one
    rank: same value as the single-GPU fused NLML"""
    from fidelityfusion_amd import functional as F
    from fidelityfusion_amd import tiled
    n, D, d = 3000, 6, 5
    from oracle import gp_oracle as O
    X, Y = O.synthetic_xy(n, D, d, seed=2)
    dev = torch.device("cuda", 0)
    Xd, Yd = torch.tensor(X, device=dev), torch.tensor(Y, device=dev)
    w, amp, dadd = torch.full((D,), 0.9, dtype=torch.float64, device=dev), torch.tensor([1.2], dtype=torch.float64, device=dev), 0.37
    ref = float(F.nlml(Xd, Yd, w, amp, diag_add=torch.tensor([dadd], dtype=torch.float64, device=dev), clamp=1e-30))
    got = tiled.tiled_nll(Xd, Yd, w, amp, dadd, clamp=1e-30, nb=512, ops=tiled.HipOps(dev))
    assert abs(got - ref) <= 1e-11 * abs(ref)
    with pytest.raises(torch.linalg.LinAlgError):
        tiled.tiled_nll(Xd, Yd, w, amp, -2.0, clamp=1e-30, nb=512, ops=tiled.HipOps(dev))


@pytest.mark.gpu
def test_tiled_cholesky_hip_ops_two_gloo_ranks_share_the_gpu():
    """two ranks (gloo, host-staged broadcasts) running the HIP operations on the box's one GPU reproduce dense LAPACK"""
    _check(_run(2, 1000, _NB, use_hip=True), 1000, 2)


def _worker_nccl_one_rank(port, out):
    """a ONE-rank `nccl` process group on the box's GPU: every collective of the tiled factorisation (asynchronous panel broadcast,
    reduce-to-owner of the substitution's partial sums, status MIN / value SUM all-reduces) goes through RCCL"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        from fidelityfusion_amd import functional as F
        from fidelityfusion_amd import tiled
        from oracle import gp_oracle as O
        n, D, d = 2500, 5, 3
        X, Y = O.synthetic_xy(n, D, d, seed=8)
        Xd, Yd = torch.tensor(X, device=dev), torch.tensor(Y, device=dev)
        w, amp, dadd = torch.full((D,), 1.1, dtype=torch.float64, device=dev), torch.tensor([0.9], dtype=torch.float64, device=dev), 0.4
        ref = float(F.nlml(Xd, Yd, w, amp, diag_add=torch.tensor([dadd], dtype=torch.float64, device=dev), clamp=1e-30))
        tc = tiled.TiledCholesky(n, nb=512, ops=tiled.HipOps(dev))
        assert tc.comm and tc.comm_dev.type == "cuda" and dist.get_backend() == "nccl"
        tc.assemble(Xd, w, amp, dadd, clamp=1e-30)
        rc = tc.factor()
        got = tc.nll_v1(Yd)
        L = tc.gather_dense_factor()
        bad = tiled.TiledCholesky(n, nb=512, ops=tiled.HipOps(dev))
        S = torch.eye(n, dtype=torch.float64)
        S[1500, 1500] = -1.0
        rc_bad = bad.load_dense(S).factor()
        out.put((rc, got, ref, float(L.abs().max()), rc_bad))
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.timeout(300)
def test_tiled_cholesky_hip_ops_one_rank_nccl_group_runs_rccl():
    """the RCCL path of tiled.py executes on the 1-GPU box (1-rank communicator): same value as the fused path"""
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    p = ctx.Process(target=_worker_nccl_one_rank, args=(_free_port(), q))
    p.start()
    p.join(240)
    assert p.exitcode == 0
    rc, got, ref, lmax, rc_bad = q.get()
    assert rc == 0 and abs(got - ref) <= 1e-11 * abs(ref) and lmax > 0
    assert rc_bad == 1501
