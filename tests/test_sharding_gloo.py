"""CPU, world_size 2 over gloo: the per-fidelity sharding logic (partition, one all-reduce of the F-vector,
posterior gather).  The per-block evaluator is injected -- here the CPU oracle, on the GPU box the fused HIP path
(tests/test_gpu_parity.py::test_cigar_blocks_sum_golden covers that evaluator)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from fidelityfusion_amd import sharding

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cigar_blocks.npz")


def test_partition_lpt():
    assert sharding.partition_lpt([1, 1, 1, 1], 4) == [0, 1, 2, 3]
    own = sharding.partition_lpt([8, 1, 1, 1, 1, 1, 1, 1, 1], 2)
    loads = [sum(c for c, o in zip([8, 1, 1, 1, 1, 1, 1, 1, 1], own) if o == r) for r in (0, 1)]
    assert sorted(loads) == [8, 8]
    assert set(sharding.partition_lpt([5.0] * 8, 8)) == set(range(8))
    assert sharding.partition_lpt([3, 2], 1) == [0, 0]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _oracle_eval(block):
    from oracle import gp_oracle as O
    ll, _ = O.cigp_ll_and_grads(block["X"], block["Y"], block["length_scales"], block["signal_variance"], block["log_beta"])
    return ll


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = dict(np.load(GOLDEN))
    F = int(g["F"])
    blocks = [dict(X=g[f"X{f}"], Y=g[f"Y{f}"], length_scales=g[f"length_scales{f}"],
                   signal_variance=g[f"signal_variance{f}"], log_beta=g[f"log_beta{f}"]) for f in range(F)]
    calls = []

    def ev(b):
        calls.append(1)
        return _oracle_eval(b)

    vec, total = sharding.joint_ll(blocks, evaluator=ev)
    q.put((rank, vec.tolist(), total, len(calls)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_joint_ll_two_ranks_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(2)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    g = dict(np.load(GOLDEN))
    F = int(g["F"])
    ref = [float(g[f"ll{f}"]) for f in range(F)]
    for rank, vec, total, ncalls in res:
        assert ncalls == F // 2                      # each rank evaluated only the blocks it owns
        assert np.allclose(vec, ref, rtol=1e-10)     # ... and still sees every block's value after the all-reduce
        assert abs(total - float(g["ll_sum"])) < 1e-9 * abs(float(g["ll_sum"]))


def test_joint_ll_single_process_matches():
    g = dict(np.load(GOLDEN))
    F = int(g["F"])
    blocks = [dict(X=g[f"X{f}"], Y=g[f"Y{f}"], length_scales=g[f"length_scales{f}"],
                   signal_variance=g[f"signal_variance{f}"], log_beta=g[f"log_beta{f}"]) for f in range(F)]
    vec, total = sharding.joint_ll(blocks, evaluator=_oracle_eval, reduce_device=torch.device("cpu"))
    assert abs(total - float(g["ll_sum"])) < 1e-9 * abs(float(g["ll_sum"]))


def _worker_hip(rank, world, port, q):
    """the REAL per-block evaluator (fused HIP NLML on cuda:0, shared by both ranks) under a 2-process gloo group"""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    torch.set_default_dtype(torch.float64)             # the evaluator builds its modules in the default dtype
    g = dict(np.load(GOLDEN))
    F = int(g["F"])
    blocks = [dict(X=g[f"X{f}"], Y=g[f"Y{f}"], length_scales=g[f"length_scales{f}"],
                   signal_variance=g[f"signal_variance{f}"], log_beta=g[f"log_beta{f}"]) for f in range(F)]
    vec, total = sharding.joint_ll(blocks)             # default evaluator: the library
    q.put((rank, vec.tolist(), total))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.timeout(300)
def test_joint_ll_two_ranks_gloo_hip_evaluator():
    """N > 1 with the shipped evaluator: two processes (one GPU between them on the test box, gloo for the F-vector
    all-reduce) each run the fused HIP NLML on the blocks they own; every rank ends up with every block's value"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_hip, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(2)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    g = dict(np.load(GOLDEN))
    F = int(g["F"])
    ref = [float(g[f"ll{f}"]) for f in range(F)]
    for rank, vec, total in res:
        assert np.allclose(vec, ref, rtol=1e-9, atol=0.0)
        assert abs(total - float(g["ll_sum"])) < 1e-9 * abs(float(g["ll_sum"]))
    assert res[0][1] == res[1][1]                      # both ranks hold the same vector after the all-reduce


def _trainer_run(F_blocks=4, steps=3, concurrent=True):
    """3 Adam steps of a 4-block model through ShardedTrainer on cuda:0 + the gathered posteriors (any world size)"""
    from fidelityfusion_amd import kernel
    from fidelityfusion_amd.cigp_v10 import cigp
    from oracle import gp_oracle as O
    torch.set_default_dtype(torch.float64)
    dev = torch.device("cuda", 0)
    shapes = [(420, 3, 2), (300, 3, 1), (515, 3, 4), (260, 3, 2)][:F_blocks]
    data = []
    for f, (n, D, d) in enumerate(shapes):
        X, Y = O.synthetic_xy(n, D, d, seed=30 + f)
        data.append((torch.tensor(X, device=dev), torch.tensor(Y, device=dev)))
    tr = sharding.ShardedTrainer(lambda f: cigp(kernel.ARDKernel(3), 0.5 + 0.1 * f).to(dev), data,
                                 [sharding.block_cost(n, d) for n, _, d in shapes], lr=5e-2, concurrent=concurrent)
    trace = [tr.step().tolist() for _ in range(steps)]
    xt = torch.tensor(O.synthetic_xy(7, 3, 1, seed=99)[0], device=dev)
    post = tr.gather_posteriors(xt)
    return trace, {f: (m.numpy(), v.numpy()) for f, (m, v) in post.items()}, sorted(tr.models)


def _worker_trainer(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    trace, post, owned = _trainer_run()
    q.put((rank, trace, post, owned))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.timeout(300)
def test_sharded_trainer_two_ranks_matches_single_process():
    """ShardedTrainer (the per-fidelity Adam loop of ResGP.py:82-88, sharded): a 2-rank gloo run on the test box's GPU
    reproduces the single-process run -- joint loss vector after every step, every block's posterior -- with the blocks
    split between the ranks; overlapped and sequential evaluation of a rank's own blocks agree as well"""
    ref_trace, ref_post, owned_all = _trainer_run(concurrent=True)
    seq_trace, _, _ = _trainer_run(concurrent=False)
    assert owned_all == [0, 1, 2, 3]
    assert np.allclose(ref_trace, seq_trace, rtol=1e-12, atol=0.0)
    assert ref_trace[0] != ref_trace[-1]                      # the parameters did move
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_trainer, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(2)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    owned = {}
    for rank, trace, post, own in res:
        owned[rank] = own
        assert np.allclose(trace, ref_trace, rtol=1e-12, atol=0.0)
        assert sorted(post) == [0, 1, 2, 3]                   # all-gathered: every rank holds every block's posterior
        for f in range(4):
            assert np.allclose(post[f][0], ref_post[f][0], rtol=1e-10, atol=1e-12)
            assert np.allclose(post[f][1], ref_post[f][1], rtol=1e-10, atol=1e-12)
    assert sorted(owned[0] + owned[1]) == [0, 1, 2, 3] and owned[0] and owned[1]


def _worker_nccl_one_rank(port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        g = dict(np.load(GOLDEN))
        F = int(g["F"])
        blocks = [dict(X=g[f"X{f}"], Y=g[f"Y{f}"], length_scales=g[f"length_scales{f}"],
                       signal_variance=g[f"signal_variance{f}"], log_beta=g[f"log_beta{f}"]) for f in range(F)]
        vec, total = sharding.joint_ll(blocks)                 # F-vector on the GPU, all-reduced by RCCL (1 rank)
        assert vec.is_cuda and dist.get_backend() == "nccl"
        trace, post, owned = _trainer_run(steps=2)             # ShardedTrainer.step's all-reduce + gather_posteriors' all-gather
        q.put((vec.cpu().tolist(), total, trace, owned))
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.timeout(300)
def test_joint_ll_and_trainer_one_rank_nccl_group_run_rccl():
    """`sharding.joint_ll` / `ShardedTrainer` with a 1-rank `nccl` process group on the 1-GPU box -- the
    collectives go through RCCL (reference site they replace: MFGP_ver2023May/ResGP.py:232-246), values as the fixture's"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_worker_nccl_one_rank, args=(_free_port(), q))
    p.start()
    vec, total, trace, owned = q.get(timeout=240)
    p.join(60)
    assert p.exitcode == 0
    g = dict(np.load(GOLDEN))
    F = int(g["F"])
    assert np.allclose(vec, [float(g[f"ll{f}"]) for f in range(F)], rtol=1e-9, atol=0.0)
    assert abs(total - float(g["ll_sum"])) < 1e-9 * abs(float(g["ll_sum"]))
    assert owned == [0, 1, 2, 3] and len(trace) == 2 and np.isfinite(trace).all()
