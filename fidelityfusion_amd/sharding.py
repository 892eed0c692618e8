"""Per-fidelity sharding of the joint log marginal likelihood.

The joint likelihood of the multi-fidelity models is a plain sum over fidelities of independent GP blocks with
disjoint hyper-parameters -- `loss += cigp_list[f].compute_loss(...)` in MFGP_ver2023May/ResGP.py:232-246, and the
2024 trainers optimise `gpr_list[i]` one after another (FidelityFusion_Models/ResGP.py:78-112, CIGAR.py:99-134).
So the blocks shard embarrassingly: fidelity f -> rank (longest-processing-time first on N_f^3), every rank
runs the fused HIP NLML(+gradient) on the blocks it owns, keeps their parameters and optimiser state, and the only
exchange is ONE all-reduce(SUM) of the F-vector of per-block values per step (8*F bytes: latency-bound over xGMI,
`torch.distributed` backend "nccl" = RCCL) so that every rank can log the joint NLML.  Prediction all-gathers
the per-fidelity posteriors.  There is no data-path collective.

The non-subset "fill" mode of the reference (MF_data.py:253-303) makes fidelity f's targets depend on the trained
posterior of f-1 and therefore does not shard ("replicas only"); the aligned/subset regime does.
"""
import torch
import torch.distributed as dist


def partition_lpt(costs, world_size):
    """Longest-processing-time-first assignment.  costs[f] ~ N_f^3 (+ N_f^2 d_f).  Returns owner[f] (rank)."""
    order = sorted(range(len(costs)), key=lambda f: (-costs[f], f))
    load = [0.0] * world_size
    owner = [0] * len(costs)
    for f in order:
        r = min(range(world_size), key=lambda i: (load[i], i))
        owner[f] = r
        load[r] += costs[f]
    return owner


def block_cost(n, d):
    return float(n) ** 3 / 3.0 + float(n) * float(n) * float(d)


def _rank_world(group=None):
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def _has_group():
    """Collectives are issued whenever a process group exists -- also a ONE-rank group, where they are arithmetically a no-op: a
    1-rank `nccl` group on a single-GPU box then drives RCCL through exactly the calls the 8-GPU run makes (tests/test_sharding_gloo.py,
    `-m gpu`), so that the first multi-GPU run is not the first time those code paths execute.  Without a process group: none."""
    return dist.is_available() and dist.is_initialized()


def _f64(a):
    """values of a block description as a float64 tensor (a Python float must not pass through torch's float32 default)"""
    import numpy as np
    if isinstance(a, torch.Tensor):
        return a.detach().to(dtype=torch.float64, device="cpu").reshape(-1)
    return torch.from_numpy(np.asarray(a, dtype=np.float64).reshape(-1))


def _block_model(block, dev, kernel, cigp):
    """cigp(ARDKernel) with the block's hyper-parameters, built in fp64 whatever torch's default dtype is"""
    k = kernel.ARDKernel(len(block["length_scales"]))
    m = cigp(k, 0.0).double()
    with torch.no_grad():
        k.length_scales.copy_(_f64(block["length_scales"]))
        k.signal_variance.copy_(_f64(block["signal_variance"])[:1])
        m.log_beta.copy_(_f64(block["log_beta"])[:1])
    return m.to(dev)


def hip_block_evaluator(device=None):
    """Default evaluator: cigp (ARD kernel) on the fused HIP path.  block = dict(X, Y, length_scales,
    signal_variance, log_beta); returns the block's +LL (what cigp.negative_log_likelihood returns) as a float."""
    from . import kernel
    from .cigp_v10 import cigp

    def evaluate(block):
        dev = device or torch.device("cuda", torch.cuda.current_device())
        m = _block_model(block, dev, kernel, cigp)
        X = torch.as_tensor(block["X"], dtype=torch.float64, device=dev)
        Y = torch.as_tensor(block["Y"], dtype=torch.float64, device=dev)
        with torch.no_grad():
            return float(m.negative_log_likelihood(X, Y))

    return evaluate


def hip_blocks_evaluator_concurrent(device=None, nslots=2):
    """Evaluates a LIST of owned blocks together: they share ONE factorisation chain (cigp_v10.negative_log_likelihood_many ->
    ffgp_nlml_fused_batch; blocks of different sizes up to 12288 rows: the ragged chain, a member leaves it when its columns are used
    up); sets the chain does not take (a member above 12288 rows beside a different one, small and large members mixed) overlap on the
    GPU through streams (functional.concurrent_blocks: one block's latency-bound panel chain runs under another's trailing updates)."""
    from . import functional as F
    from . import kernel
    from .cigp_v10 import cigp

    def evaluate_many(owned):
        dev = device or torch.device("cuda", torch.cuda.current_device())
        models = [_block_model(block, dev, kernel, cigp) for block in owned]
        Xs = [torch.as_tensor(block["X"], dtype=torch.float64, device=dev).contiguous() for block in owned]
        Ys = [torch.as_tensor(block["Y"], dtype=torch.float64, device=dev).contiguous() for block in owned]
        with torch.no_grad():
            if F.many_batchable([(x.shape[0], y.shape[1]) for x, y in zip(Xs, Ys)]):
                # blocks of one shape (or all small): ONE library call -- equal-size blocks share one factorisation chain
                # (4 blocks of N = 8192, d = 1024: 20.5 ms against 24.1 ms overlapped through streams and 28.6 ms one after the other)
                from .cigp_v10 import negative_log_likelihood_many
                return [float(v) for v in negative_log_likelihood_many(models, Xs, Ys)]
            outs = []
            with F.concurrent_blocks(nslots=max(1, min(nslots, len(owned))), device_index=dev.index) as cb:
                for i, (m, X, Y) in enumerate(zip(models, Xs, Ys)):
                    with cb.slot(i):
                        outs.append(m.negative_log_likelihood(X, Y))
        return [float(o) for o in outs]

    return evaluate_many


def joint_ll(blocks, evaluator=None, group=None, reduce_device=None):
    """Every rank evaluates the blocks it owns; one all-reduce(SUM) of the F-vector.
    Returns (ll_per_block [F] tensor, joint LL float) -- identical on every rank.
    evaluator: block -> float (injected in CPU tests); default = the fused HIP path, owned blocks overlapped."""
    rank, world = _rank_world(group)
    costs = [block_cost(len(b["X"]), b["Y"].shape[1] if hasattr(b["Y"], "shape") else 1) for b in blocks]
    owner = partition_lpt(costs, world)
    if reduce_device is None:
        use_cuda = dist.is_initialized() and dist.get_backend(group) == "nccl"
        reduce_device = torch.device("cuda", torch.cuda.current_device()) if use_cuda else torch.device("cpu")
    vec = torch.zeros(len(blocks), dtype=torch.float64, device=reduce_device)
    mine = [f for f in range(len(blocks)) if owner[f] == rank]
    if evaluator is None:
        vals = hip_blocks_evaluator_concurrent()([blocks[f] for f in mine]) if mine else []
    else:
        vals = [evaluator(blocks[f]) for f in mine]
    for f, v in zip(mine, vals):
        vec[f] = v
    if _has_group():
        dist.all_reduce(vec, op=dist.ReduceOp.SUM, group=group)
    return vec, float(vec.sum())


def joint_nll_single_process(blocks, device=None):
    """All blocks on one GPU (the N = 1 case of joint_ll); returns the list of per-block +LL."""
    ev = hip_block_evaluator(torch.device(device) if device is not None else None)
    return [ev(b) for b in blocks]


class ShardedTrainer:
    """One Adam step per call on every owned block (the reference's per-fidelity loop body,
    FidelityFusion_Models/ResGP.py:82-88), then the scalar all-reduce for the joint value.
    `make_model(f)` -> nn.Module with `.negative_log_likelihood(x, y)`; `data[f]` = (x, y)."""

    def __init__(self, make_model, data, costs, lr=1e-2, group=None, concurrent=True, nslots=2, slot_lookahead=False):
        self.group = group
        self.concurrent = concurrent
        self.nslots, self.slot_lookahead = nslots, slot_lookahead
        self.rank, self.world = _rank_world(group)
        self.owner = partition_lpt(costs, self.world)
        self.F = len(costs)
        self.models, self.opts, self.data = {}, {}, {}
        for f in range(self.F):
            if self.owner[f] == self.rank:
                self.models[f] = make_model(f)
                self.opts[f] = torch.optim.Adam(self.models[f].parameters(), lr=lr)
                self.data[f] = data[f]

    def step(self, reduce_device=None):
        dev = reduce_device
        if dev is None:
            use_cuda = dist.is_initialized() and dist.get_backend(self.group) == "nccl"
            dev = torch.device("cuda", torch.cuda.current_device()) if use_cuda else torch.device("cpu")
        vec = torch.zeros(self.F, dtype=torch.float64, device=dev)
        losses = {}
        on_gpu = torch.cuda.is_available() and len(self.models) > 1 and self.concurrent
        if on_gpu and self._chainable():
            # the rank's blocks share ONE factorisation chain (cigp_v10.negative_log_likelihood_many -> ffgp_nlml_fused_batch; blocks of
            # different sizes: the ragged chain) -- forward and the gradient stages; one backward for the independent blocks' sum
            from .cigp_v10 import negative_log_likelihood_many
            order = list(self.models)
            for f in order:
                self.opts[f].zero_grad()
            vals = negative_log_likelihood_many([self.models[f] for f in order], [self.data[f][0] for f in order],
                                                [self.data[f][1] for f in order])
            (-vals).sum().backward()
            for i, f in enumerate(order):
                self.opts[f].step()
                vec[f] = -vals[i].detach().to(dev)
            if _has_group():
                dist.all_reduce(vec, op=dist.ReduceOp.SUM, group=self.group)
            return vec
        if on_gpu:   # owned blocks overlap on the GPU; gradients were produced by the same fused calls
            from . import functional as F
            with F.concurrent_blocks(nslots=min(self.nslots, len(self.models)), lookahead=self.slot_lookahead) as cb:
                for i, (f, m) in enumerate(self.models.items()):
                    self.opts[f].zero_grad()
                    with cb.slot(i):
                        losses[f] = -m.negative_log_likelihood(*self.data[f])
        else:
            for f, m in self.models.items():
                self.opts[f].zero_grad()
                losses[f] = -m.negative_log_likelihood(*self.data[f])
        for f, m in self.models.items():
            losses[f].backward()
            self.opts[f].step()
            vec[f] = losses[f].detach().to(dev)
        if _has_group():
            dist.all_reduce(vec, op=dist.ReduceOp.SUM, group=self.group)
        return vec

    def _chainable(self):
        """the owned models are `cigp` modules on one GPU in fp64 whose shapes share a factorisation chain"""
        from . import functional as F
        ms = list(self.models.values())
        if not all(hasattr(m, "kernel") and hasattr(m, "log_beta") for m in ms):
            return False
        shapes = []
        for f, m in self.models.items():
            x, y = self.data[f]
            if isinstance(y, list) or not isinstance(x, torch.Tensor) or not x.is_cuda or F.raw_many_ok(m.kernel, x, y, m.log_beta) is None:
                return False
            shapes.append((x.shape[0], y.shape[1]))
        return F.many_batchable(shapes) and all(n > F.SMALL_BATCH_MAX_N for n, _ in shapes)

    def gather_posteriors(self, x_test):
        """all-gather of the per-fidelity posterior means / variances at x_test (each rank computes its own
        blocks; the cheap residual chain is applied by the caller, FidelityFusion_Models/CIGAR.py:75-76)."""
        local = {}
        with torch.no_grad():
            for f, m in self.models.items():
                mean, var = m(self.data[f][0], self.data[f][1], x_test)
                local[f] = (mean.cpu(), var.cpu())
        if not _has_group():
            return local
        out = [None] * self.world
        dist.all_gather_object(out, local, group=self.group)
        merged = {}
        for part in out:
            merged.update(part)
        return merged
