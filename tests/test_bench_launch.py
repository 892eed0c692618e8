"""bench.py's multi-rank path on CPU: `python bench.py --gpus 2 --backend gloo --dry` must launch its own two ranks (the
parent never imports torch), deal the blocks by LPT, all-reduce the F-vector and print ONE JSON line (rank 0)."""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*extra, env_extra=None):
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--backend", "gloo", "--dry", "--steps", "2", "--warmup",
                        "1"] + list(extra), env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    return json.loads(lines[0])


def _expected_joint(F_, n, D, d):
    sys.path.insert(0, ROOT)
    import bench
    return sum(float(np.square(bench.synthetic_xy(n, D, d, seed=f)[1]).sum()) + f for f in range(F_))


def test_self_launch_two_ranks_weak_and_sharded_legs():
    out = _run("--gpus", "2", "--n", "96")
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["config"]["blocks"] == 2
    assert out["collective"] == {"backend": "gloo", "ranks": 2}
    # every rank of a multi-rank run pins itself to its own slice of the host's CPUs before it touches the GPU (bench.pin_rank)
    aff = out["cpu_affinity"]
    assert aff is None or (aff["rank0_cpus"] >= 1 and aff["first"] <= aff["last"])
    for key in ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "higher_is_better", "vs_baseline", "dtype", "data",
                "roofline"):
        assert key in out
    assert abs(out["joint_nll"] - _expected_joint(2, 96, 16, 1)) < 1e-9 * abs(out["joint_nll"])
    # the fixed-F legs ride along: 4 and 8 blocks over 2 ranks, values summed by the all-reduce
    sh = out["sharded"]
    assert sh["cigar4"]["blocks"] == 4 and sh["cigar4"]["blocks_per_rank"] == 2 and sh["cigar4"]["scaling"] == "strong"
    assert sh["gar8"]["blocks"] == 8 and sh["gar8"]["blocks_per_rank"] == 4
    assert abs(sh["gar8"]["joint_nll"] - _expected_joint(8, 64, 8, 8)) < 1e-9 * abs(sh["gar8"]["joint_nll"])
    # ... and the HOGP wording of config 5 ("HOGP variant reported separately"), dealt and reduced the same way
    hg = sh["gar8_hogp"]
    assert hg["blocks"] == 8 and hg["blocks_per_rank"] == 4 and "HOGP" in hg["config"] and hg["value"] > 0
    assert abs(hg["sum_block_loss"] - _expected_joint(8, 64, 8, 8)) < 1e-9 * abs(hg["sum_block_loss"])
    assert hg["primary"] == "ms_per_step" and hg["value_canonical_syevd"] < hg["value"]


def test_fixed_blocks_same_joint_value_on_1_2_3_ranks():
    vals = []
    for g in (1, 2, 3):
        out = _run("--gpus", str(g), "--workload", "gar8", "--n", "80", "--d", "4")
        assert out["scaling"] == "strong" and out["config"]["blocks"] == 8 and out["n_gpus"] == g
        vals.append(out["joint_nll"])
    assert abs(vals[0] - _expected_joint(8, 80, 8, 4)) < 1e-9 * abs(vals[0])
    assert abs(vals[1] - vals[0]) < 1e-9 * abs(vals[0]) and abs(vals[2] - vals[0]) < 1e-9 * abs(vals[0])


def test_hogp_workload_flop_count_and_dry_run():
    sys.path.insert(0, ROOT)
    import bench
    n = 8192
    fl = bench.hogp_flops(n, (64, 64))
    assert abs(fl - (26.0 / 3.0 * n ** 3 + 4.0 * n * n * 4096 + 4.0 * n * 4096 * 128)) < 1e-6 * fl
    out = _run("--gpus", "2", "--workload", "gar8_hogp", "--n", "80", "--d", "4")
    assert out["scaling"] == "strong" and out["config"]["blocks"] == 8 and "gar8_hogp" in out["config"]["workload"]


def test_runs_as_a_rank_under_an_external_launcher():
    # what `python -m torch.distributed.run --nproc-per-node 1` sets up: WORLD_SIZE present -> no self-launch
    out = _run("--gpus", "1", "--n", "64", env_extra={"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1",
                                                      "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29591"})
    assert out["n_gpus"] == 1


def test_world_size_mismatch_is_an_error_not_a_hang():
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29592")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--dry"], env=env,
                       capture_output=True, text=True, timeout=120)
    assert p.returncode != 0 and "WORLD_SIZE" in (p.stderr + p.stdout)


def test_parent_of_a_self_launch_never_imports_torch():
    code = ("import sys, os; sys.argv = ['bench.py', '--gpus', '2', '--backend', 'gloo', '--dry', '--n', '64', '--steps', '1', "
            "'--warmup', '0', '--no-sharded']\n"
            "sys.path.insert(0, %r)\nimport bench\n"
            "try:\n    bench.main()\nexcept SystemExit as e:\n    assert e.code == 0, e.code\n"
            "assert 'torch' not in sys.modules, 'the launching parent imported torch'\nprint('parent clean')\n" % ROOT)
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "parent clean" in p.stdout, p.stderr[-2000:]


def test_more_ranks_than_visible_gpus_fails_fast_with_a_message():
    """`bench.py --gpus 2` on a box that shows fewer than 2 devices (here: none) must end at once with a message that names the
    count, not hang in a rendezvous or die in torch.cuda.set_device"""
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env,
                       capture_output=True, text=True, timeout=240)
    assert p.returncode != 0
    assert "GPU" in p.stderr and "visible" in p.stderr and "--gpus 2" in p.stderr, p.stderr[-1500:]


def test_importing_the_package_leaves_the_environment_alone():
    """the hardware-queue default is an explicit, logged call (`_lib.configure_queues`), not an import side effect"""
    code = ("import os\nos.environ.pop('GPU_MAX_HW_QUEUES', None)\nimport sys\nsys.path.insert(0, %r)\n"
            "import fidelityfusion_amd\nfrom fidelityfusion_amd import _lib, functional\n"
            "assert 'GPU_MAX_HW_QUEUES' not in os.environ\n"
            "import logging\nlogging.basicConfig(level=logging.INFO)\n"
            "assert _lib.configure_queues() == '8' and os.environ['GPU_MAX_HW_QUEUES'] == '8'\n"
            "os.environ['GPU_MAX_HW_QUEUES'] = '5'\nassert _lib.configure_queues(6) == '5'\nprint('ok')\n" % ROOT)
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=240)
    assert p.returncode == 0 and "ok" in p.stdout, p.stderr[-1500:]
    assert "GPU_MAX_HW_QUEUES=8 set for this process" in p.stderr


def test_pin_rank_gives_disjoint_slices():
    """bench.pin_rank: ranks that share a pool of CPUs get disjoint, non-empty slices; FFGP_BENCH_AFFINITY=0 leaves the mask alone"""
    sys.path.insert(0, ROOT)
    code = (
        "import os, sys, json; sys.path.insert(0, %r); import bench\n"
        "before = sorted(os.sched_getaffinity(0))\n"
        "mine = bench.pin_rank(int(sys.argv[1]), 2)\n"
        "print(json.dumps({'before': before, 'mine': mine, 'now': sorted(os.sched_getaffinity(0))}))\n" % ROOT)
    outs = []
    for r in (0, 1):
        p = subprocess.run([sys.executable, "-c", code, str(r)], capture_output=True, text=True, timeout=60)
        assert p.returncode == 0, p.stderr
        outs.append(json.loads(p.stdout.strip().splitlines()[-1]))
    if len(outs[0]["before"]) >= 2:
        assert outs[0]["mine"] and outs[1]["mine"] and not set(outs[0]["mine"]) & set(outs[1]["mine"])
        assert outs[0]["now"] == outs[0]["mine"]
    env = dict(os.environ, FFGP_BENCH_AFFINITY="0")
    p = subprocess.run([sys.executable, "-c", code, "0"], capture_output=True, text=True, timeout=60, env=env)
    o = json.loads(p.stdout.strip().splitlines()[-1])
    assert o["mine"] is None and o["now"] == o["before"]


def _fake_sysfs(root, ncards=8, nodes=2, cpus_per_node=64):
    """a sysfs tree of a 2-socket node with eight AMD GPUs, four per socket (what bench.gpu_numa_cpus reads)"""
    for c in range(ncards):
        pci = os.path.join(root, "devices", "pci0000:%02x" % (0x10 * c), "0000:%02x:00.0" % (0x10 * c + 1))
        os.makedirs(pci)
        open(os.path.join(pci, "vendor"), "w").write("0x1002\n")
        open(os.path.join(pci, "numa_node"), "w").write("%d\n" % (c * nodes // ncards))
        card = os.path.join(root, "class", "drm", "card%d" % c)
        os.makedirs(card)
        os.symlink(pci, os.path.join(card, "device"))
    for nd in range(nodes):     # socket 0: CPUs 0-63 and their SMT siblings 128-191; socket 1: 64-127, 192-255
        d = os.path.join(root, "devices", "system", "node", "node%d" % nd)
        os.makedirs(d)
        lo = nd * cpus_per_node
        open(os.path.join(d, "cpulist"), "w").write("%d-%d,%d-%d\n" % (lo, lo + cpus_per_node - 1, lo + nodes * cpus_per_node,
                                                                       lo + nodes * cpus_per_node + cpus_per_node - 1))


def test_eight_rank_cpu_slices_are_disjoint_on_a_two_socket_node(tmp_path):
    """pre-flight of the 8-GPU run that needs no GPU: on a 2-socket host with four GPUs per socket every rank's slice lies inside its
    GPU's NUMA node, the eight slices are pairwise disjoint and together cover both sockets"""
    sys.path.insert(0, ROOT)
    import bench
    root = str(tmp_path / "sys")
    _fake_sysfs(root)
    allowed = list(range(256))
    node_of = lambda c: (c % 128) // 64
    slices = [bench.rank_cpu_slice(r, 8, allowed, sysfs=root) for r in range(8)]
    for r, sl in enumerate(slices):
        assert len(sl) == 32 and {node_of(c) for c in sl} == {r // 4}, (r, sl)
    flat = [c for sl in slices for c in sl]
    assert len(flat) == len(set(flat)) == 256
    # a cgroup that only allows half of each socket: still disjoint, still on the right socket
    allowed = [c for c in range(256) if c % 2 == 0]
    slices = [bench.rank_cpu_slice(r, 8, allowed, sysfs=root) for r in range(8)]
    flat = [c for sl in slices for c in sl]
    assert len(flat) == len(set(flat)) == 128 and all({node_of(c) for c in sl} == {r // 4} for r, sl in enumerate(slices))
    # sysfs silent (no cards): an even split of the allowed CPUs
    empty = str(tmp_path / "none")
    os.makedirs(empty)
    slices = [bench.rank_cpu_slice(r, 8, list(range(16)), sysfs=empty) for r in range(8)]
    assert slices == [[2 * r, 2 * r + 1] for r in range(8)]


def test_eight_rank_dry_run_deals_one_block_per_gpu():
    """`bench.py --gpus 8 --dry --backend gloo`: the launch the driver makes on an 8-GPU node, on CPU.  gar8 (BASELINE configs[4]) puts
    exactly ONE block on every rank and every rank takes the single-block path (F.nlml -- the `blocks` leg of the 1-GPU line prices
    that unit); cigar4 (configs[3]) gives four ranks one block each and leaves four idle; the joint values are the 1-rank values"""
    out = _run("--gpus", "8", "--n", "64")
    assert out["n_gpus"] == 8 and out["collective"] == {"backend": "gloo", "ranks": 8}
    assert out["config"]["owner"] == list(range(8)) and out["config"]["rank_paths"] == ["single"] * 8
    g8, c4, hg = out["sharded"]["gar8"], out["sharded"]["cigar4"], out["sharded"]["gar8_hogp"]
    assert sorted(g8["owner"]) == list(range(8)) and g8["rank_paths"] == ["single"] * 8 and g8["blocks_per_rank"] == 1
    assert sorted(hg["owner"]) == list(range(8)) and hg["rank_paths"] == ["single"] * 8
    assert len(set(c4["owner"])) == 4 and sorted(c4["rank_paths"]) == ["idle"] * 4 + ["single"] * 4
    assert abs(g8["joint_nll"] - _expected_joint(8, 64, 8, 8)) < 1e-9 * abs(g8["joint_nll"])
    assert abs(c4["joint_nll"] - _expected_joint(4, 64, 8, 8)) < 1e-9 * abs(c4["joint_nll"])
    # four ranks: cigar4 one block per GPU, gar8 two per GPU through ONE shared chain
    out = _run("--gpus", "4", "--n", "64")
    assert out["sharded"]["cigar4"]["rank_paths"] == ["single"] * 4
    assert out["sharded"]["gar8"]["rank_paths"] == ["chain"] * 4 and out["sharded"]["gar8_hogp"]["rank_paths"] == ["threads"] * 4


def test_block_path_is_what_the_step_runs():
    sys.path.insert(0, ROOT)
    import bench
    assert [bench.block_path(k) for k in (0, 1, 2, 8)] == ["idle", "single", "chain", "chain"]
    assert bench.block_path(3, no_chain_batch=True) == "streams" and bench.block_path(1, no_chain_batch=True) == "single"
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert 'if path == "single":' in src and 'elif path == "chain":' in src and 'elif path == "streams":' in src


import pytest  # noqa: E402


@pytest.mark.gpu
def test_roofline_flops_count_every_block_of_a_shared_chain():
    """the trailing-update statistics behind `roofline.achieved` must count all F blocks a shared-chain launch covers: the same
    workload with one block (single call) and with four (one chain) reports the same launches per step and four times the flops"""
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("no GPU")

    def run(blocks):
        env = dict(os.environ)
        for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
            env.pop(k, None)
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "cigar4", "--blocks", str(blocks), "--steps", "2",
                            "--warmup", "1", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-2000:]
        return json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][0])
    one, four = run(1), run(4)
    r1, r4 = one["roofline"], four["roofline"]
    assert r1["launches"] == r4["launches"] > 0
    assert abs(r4["avg_launch_gflop"] - 4.0 * r1["avg_launch_gflop"]) <= 1e-3 * r4["avg_launch_gflop"]     # (the line rounds to 3 decimals)
    assert 0.2 < r4["frac"] < 1.0


@pytest.mark.gpu
def test_two_ranks_share_the_one_gpu_real_step_over_gloo():
    """`--gpus 2 --backend gloo` with the REAL (not dry) step, both ranks on cuda:0 (FFGP_BENCH_ONE_GPU=1): self-launch, per-rank
    CPU pinning before the first GPU call, LPT partition, the fused HIP blocks, the F-vector all-reduce and the JSON line -- the
    multi-rank code path end to end on the one-GPU box; the joint value must be the single-rank run's"""
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("no GPU")

    def run(gpus):
        env = dict(os.environ, FFGP_BENCH_ONE_GPU="1")
        for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
            env.pop(k, None)
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--backend", "gloo", "--workload", "cigar4",
                            "--n", "1536", "--d", "64", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"], env=env,
                           capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-2000:]
        return json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][0])
    one, two = run(1), run(2)
    assert two["n_gpus"] == 2 and two["collective"] == {"backend": "gloo", "ranks": 2} and two["config"]["blocks"] == 4
    assert two["cpu_affinity"] is None or two["cpu_affinity"]["rank0_cpus"] >= 1
    assert abs(two["joint_nll"] - one["joint_nll"]) <= 1e-12 * abs(one["joint_nll"])


@pytest.mark.gpu
@pytest.mark.parametrize("queues", [1, 2])
def test_handles_with_lookahead_side_by_side_on_shared_hardware_queues(queues):
    """the look-ahead's value waits are kernels at the head of their hardware queue, and the streams of several handles share queues
    once a process has more streams than queues: every wait must be enqueued after the launch that satisfies it (ffgp_potrf_impl's
    ordering rule), or two handles end up in front of each other's producers.  Three host threads with a handle each factor blocks that
    all use the side stream, with ONE (and two) hardware queues per priority for the whole process -- tools/handoff_stress.py hangs
    within seconds when that rule is broken, and checks every value against the block's value alone"""
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    env = dict(os.environ, FFGP_STRESS_DEADLINE_S="100")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "handoff_stress.py"), "3", "25", str(queues), str(queues)], env=env,
                       capture_output=True, text=True, timeout=240)
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("handoff_stress")]
    assert p.returncode == 0 and line and "HUNG" not in line[0], (p.stdout[-1500:], p.stderr[-1500:])
