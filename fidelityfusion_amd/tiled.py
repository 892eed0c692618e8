"""One GP block across several GPUs: a column-panel (1-D block-cyclic) distributed Cholesky and the likelihood on top of
it -- SURVEY.md section 8(f) row 4, for a single block whose N x N covariance no longer fits one GPU's 288 GB
(N >~ 180 000 in fp64).  The per-fidelity sharding of `sharding.py` needs no data-path collective; this does: one
panel broadcast per outer step.

Layout.  Sigma is cut into column panels of width `nb` (default 512 = the single-GPU outer block); panel k belongs to rank
k mod R and is stored on its owner only, from its diagonal block down: a dense [(n - k nb) x w_k] row-major tensor whose
top w_k x w_k block is the diagonal block.  Memory per rank ~ 8 N^2 / (2 R) bytes.  Every rank assembles its own panels
from X (the inputs are replicated: N x D is tiny next to N x N), so the covariance never travels.

Factorisation (right-looking, one panel of look-ahead):

    step k   owner(k):  potrf_rows(panel k)      L_kk and L[k+1:, k] = A[k+1:, k] L_kk^-T in ONE call -- the rows below the
                                                 diagonal block are the single-GPU path's "passenger rows" (ffgp_potrf_rows)
             everyone:  broadcast(panel k)       RCCL broadcast from owner(k), (n - k nb) x w_k doubles, posted asynchronously
             owner(k+1) first updates, factors and posts panel k+1, THEN updates its remaining panels with panel k --
             the broadcast of panel k+1 runs under everybody's trailing updates of step k
             every rank: panel j -= P[j-rows:, :] P[j-block, :]^T for its own panels j > k   (local MFMA GEMM, K = w_k)

Per step the wire carries 8 (n - k nb) w_k bytes to each rank (ring / tree broadcast over xGMI) against
2 (n - k nb)^2 w_k / R flops of local update per rank: at N = 262 144, nb = 512, R = 8 the first step moves 1.07 GB
(~10 ms at ~100 GB/s per link) under 8.8 TFLOP of update per rank (~150 ms) -- the broadcast hides; the scheme turns
communication-bound only when (n - k nb) / R falls below ~3 000 rows, i.e. in the last few percent of the flops.

Likelihood.  Gamma = L^-1 Y by a left-looking block substitution that keeps the partial sums local: every rank holds
Z^(r) = sum over its own panels j of L[:, j] Gamma_j; at step k ONE reduce of the w_k x d block Z_k to owner(k) completes
(Y_k - Z_k), the owner solves Gamma_k = L_kk^-1 (.) and adds L[k+1:, k] Gamma_k to its Z.  log-det and ||Gamma||^2 are
local sums followed by one 2-scalar all-reduce.  nll = 1/2 ||Gamma||^2 + d sum log L_ii + 1/2 N d log(2 pi~) -- the V1
formula of the single-GPU path (cigp_v10.py:67-68), same constants.

The arithmetic is injected (`ops`): `HipOps` drives libffgp (ffgp_assemble / ffgp_potrf_rows / ffgp_gemm /
ffgp_trsm_lower) on this rank's GPU; the CPU tests inject a torch double of the four operations (tests/tiled_torch_ops.py) to prove the distributed algorithm over gloo.  No
multi-GPU hardware number is claimed for this module: the pool has 1-GPU boxes.
"""
import math

import torch
import torch.distributed as dist

PI_TRUNC = 3.1415   # GaussianProcess/cigp_v10.py:15


def _rank_world(group=None):
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


class PanelLayout:
    """Column panels of width nb, panel k owned by rank k mod world."""

    def __init__(self, n, nb, world):
        if n <= 0 or nb <= 0 or nb % 2:
            raise ValueError("need n > 0 and an even panel width nb > 0 (the device path loads 16-byte pairs)")
        self.n, self.nb, self.world = int(n), int(nb), int(world)
        self.npanels = (n + nb - 1) // nb

    def start(self, k):
        return k * self.nb

    def width(self, k):
        return min(self.nb, self.n - k * self.nb)

    def rows(self, k):
        return self.n - k * self.nb

    def owner(self, k):
        return k % self.world

    def owned(self, rank):
        return [k for k in range(self.npanels) if self.owner(k) == rank]

    def bytes_per_rank(self, rank):
        return sum(8 * self.rows(k) * self.width(k) for k in self.owned(rank))


class HipOps:
    """The same operations on libffgp (one rank = one GPU).  Panels are contiguous fp64 tensors on the rank's device."""

    def __init__(self, device=None):
        from . import _lib
        from . import functional as F
        self._lib, self._F = _lib, F
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())

    def _h(self):
        h = self._lib.handle(self.device.index)
        self._lib.bind_stream(h, self.device.index)
        return h

    def kernel_panel(self, Xr, Xc, w, amp, clamp):
        return self._F.kernel_matrix(Xr, Xc, w, amp, clamp)

    def potrf_rows(self, T, w):
        import ctypes as C
        rc = self._lib.check(self._lib.lib.ffgp_potrf_rows(self._h(), C.c_void_p(T.data_ptr()), w, T.shape[0], T.stride(0)),
                             "ffgp_potrf_rows")
        return rc

    def update(self, C_, A, B):
        import ctypes as C
        m, n, k = A.shape[0], B.shape[0], A.shape[1]
        if m and n:
            self._lib.check(self._lib.lib.ffgp_gemm(self._h(), 0, 0, 0, 0, C.c_void_p(A.data_ptr()), A.stride(0), C.c_void_p(B.data_ptr()),
                                                    B.stride(0), C.c_void_p(C_.data_ptr()), C_.stride(0), m, n, k, -1.0, 1.0), "ffgp_gemm")

    def trsm_lower(self, Lkk, B):
        import ctypes as C
        self._lib.check(self._lib.lib.ffgp_trsm_lower(self._h(), C.c_void_p(Lkk.data_ptr()), Lkk.shape[1], Lkk.stride(0),
                                                      C.c_void_p(B.data_ptr()), B.shape[1], B.stride(0)), "ffgp_trsm_lower")

    def gemm_acc(self, Z, A, G):
        import ctypes as C
        m, n, k = A.shape[0], G.shape[1], A.shape[1]
        if m and n:
            self._lib.check(self._lib.lib.ffgp_gemm(self._h(), 0, 1, 0, 0, C.c_void_p(A.data_ptr()), A.stride(0), C.c_void_p(G.data_ptr()),
                                                    G.stride(0), C.c_void_p(Z.data_ptr()), Z.stride(0), m, n, k, 1.0, 1.0), "ffgp_gemm")


class TiledCholesky:
    """Distributed factor of ONE covariance.  `panels[k]` exists on owner(k) only."""

    def __init__(self, n, nb=512, group=None, ops=None, comm_device=None):
        self.group = group
        self.rank, self.world = _rank_world(group)
        # collectives are issued whenever a process group exists, also a ONE-rank group (arithmetically a no-op there): a 1-rank
        # `nccl` group on a single-GPU box runs the panel broadcast / reduce-to-owner / status all-reduce through RCCL exactly as
        # the multi-GPU run will (tests/test_tiled_gloo.py, `-m gpu`)
        self.comm = dist.is_available() and dist.is_initialized()
        self.layout = PanelLayout(n, nb, self.world)
        self.ops = ops if ops is not None else HipOps()
        if isinstance(self.ops, HipOps) and n % 2:
            raise ValueError("the device path needs an even n (16-byte operand loads): pad the block by one point")
        self.dev = self.ops.device
        # collectives run on the compute device with RCCL ("nccl"), through host staging with gloo
        if comm_device is None:
            use_dev = self.comm and dist.get_backend(group) == "nccl"
            comm_device = self.dev if (use_dev or not self.comm) else torch.device("cpu")
        self.comm_dev = torch.device(comm_device)
        self.panels = {}
        self.info = 0

    # ---- assembly: every rank builds its own panels from the replicated inputs ---------------------------------------
    def assemble(self, X, w, amp, diag_add, clamp=float("-inf")):
        """Sigma = amp exp(-1/2 max(||(x - x') o w||^2, clamp)) + diag_add I, panel by panel, owners only."""
        lay = self.layout
        X = X.to(device=self.dev, dtype=torch.float64)
        w = w.to(device=self.dev, dtype=torch.float64).reshape(-1)
        amp = amp.to(device=self.dev, dtype=torch.float64).reshape(-1)[:1]
        dadd = float(diag_add.detach().double().reshape(-1)[0]) if isinstance(diag_add, torch.Tensor) else float(diag_add)
        for k in lay.owned(self.rank):
            k0, wk = lay.start(k), lay.width(k)
            T = self.ops.kernel_panel(X[k0:], X[k0:k0 + wk], w, amp, clamp).contiguous()
            T[:wk, :wk].diagonal().add_(dadd)
            self.panels[k] = T
        return self

    def load_dense(self, Sigma):
        """(tests) take the owned panels out of a dense matrix every rank holds."""
        lay = self.layout
        for k in lay.owned(self.rank):
            k0, wk = lay.start(k), lay.width(k)
            self.panels[k] = Sigma[k0:, k0:k0 + wk].to(device=self.dev, dtype=torch.float64).contiguous()
        return self

    # ---- communication -----------------------------------------------------------------------------------------------
    def _post_broadcast(self, k, buf):
        """post the broadcast of panel k into `buf` ([rows(k), nb] on comm_dev); returns (work, view)"""
        lay = self.layout
        if not self.comm:
            return None, self.panels[k]
        view = buf[:lay.rows(k), :lay.width(k)]
        src = lay.owner(k)
        if self.rank == src:
            if self.panels[k].device == view.device:
                view = self.panels[k]                    # the owner sends its panel as it lies (dense [rows, w])
            else:
                view.copy_(self.panels[k])
        root = dist.get_global_rank(self.group, src) if self.group is not None else src
        work = dist.broadcast(view, src=root, group=self.group, async_op=True)
        return work, view

    def _landed(self, work, view, k):
        if work is not None:
            work.wait()
        if not self.comm or self.layout.owner(k) == self.rank:
            return self.panels[k]          # the owner reads its own copy
        return view if view.device == self.dev else view.to(self.dev)

    # ---- factorisation -------------------------------------------------------------------------------------------------
    def _update_panel(self, j, k, P):
        """panel j -= P[rows >= j0, :] P[j-block, :]^T with P = panel k (rows counted from k0)"""
        lay = self.layout
        off = lay.start(j) - lay.start(k)
        self.ops.update(self.panels[j], P[off:], P[off:off + lay.width(j)])

    def _factor_panel(self, k):
        rc = self.ops.potrf_rows(self.panels[k], self.layout.width(k))
        if rc and not self.info:
            self.info = self.layout.start(k) + int(rc)

    def factor(self):
        """In-place right-looking factorisation with one panel of look-ahead.  Returns 0, or the 1-based index of the first
        non-positive pivot (every rank returns the same value: one MAX all-reduce of the status at the end)."""
        lay = self.layout
        K = lay.npanels
        mine = set(lay.owned(self.rank))
        # contiguous [rows, nb] staging: the broadcast views must be dense, so each buffer is re-viewed per panel
        bufs = [torch.empty((lay.rows(0) * lay.nb,), dtype=torch.float64, device=self.comm_dev) for _ in range(2 if self.comm else 0)]

        def staging(k):
            return bufs[k & 1][:lay.rows(k) * lay.width(k)].view(lay.rows(k), lay.width(k)) if self.comm else None

        if 0 in mine:
            self._factor_panel(0)
        work, view = self._post_broadcast(0, staging(0)) if self.comm else (None, None)
        for k in range(K):
            P = self._landed(work, view, k)
            nxt = k + 1
            if nxt < K:
                if nxt in mine:                      # look-ahead: the next panel first, so its broadcast can start
                    self._update_panel(nxt, k, P)
                    self._factor_panel(nxt)
                work, view = self._post_broadcast(nxt, staging(nxt)) if self.comm else (None, None)
            for j in sorted(mine):
                if j > nxt:
                    self._update_panel(j, k, P)
        if self.comm:
            st = torch.tensor([self.info if self.info else 2 ** 31 - 1], dtype=torch.int64, device=self.comm_dev)
            dist.all_reduce(st, op=dist.ReduceOp.MIN, group=self.group)     # the FIRST failing pivot over all ranks
            self.info = 0 if int(st) == 2 ** 31 - 1 else int(st)
        return self.info

    # ---- likelihood ------------------------------------------------------------------------------------------------------
    def nll_v1(self, Y, pi_const=PI_TRUNC):
        """1/2 ||L^-1 Y||^2 + d sum log L_ii + 1/2 N d log(2 pi_const) on the factored panels (identical on every rank)."""
        lay = self.layout
        n = lay.n
        Y = Y.to(device=self.dev, dtype=torch.float64)
        d = Y.shape[1]
        Z = torch.zeros((n, d), dtype=torch.float64, device=self.dev)      # this rank's partial sums  sum_j L[:, j] Gamma_j
        quad = torch.zeros((), dtype=torch.float64, device=self.dev)
        logdet = torch.zeros((), dtype=torch.float64, device=self.dev)
        for k in range(lay.npanels):
            k0, wk = lay.start(k), lay.width(k)
            own = lay.owner(k)
            zk = Z[k0:k0 + wk]
            if self.comm:
                zc = zk.to(self.comm_dev).contiguous()
                root = dist.get_global_rank(self.group, own) if self.group is not None else own
                dist.reduce(zc, dst=root, op=dist.ReduceOp.SUM, group=self.group)
                if self.rank == own:
                    zk = zc.to(self.dev)
            if self.rank == own:
                T = self.panels[k]
                G = (Y[k0:k0 + wk] - zk).contiguous()
                self.ops.trsm_lower(T[:wk, :wk], G)                         # Gamma_k
                quad += (G * G).sum()
                logdet += torch.log(T[:wk, :wk].diagonal()).sum()
                if T.shape[0] > wk:
                    self.ops.gemm_acc(Z[k0 + wk:], T[wk:], G)
        pair = torch.stack([quad, logdet]).to(self.comm_dev)
        if self.comm:
            dist.all_reduce(pair, op=dist.ReduceOp.SUM, group=self.group)
        return float(0.5 * pair[0] + d * pair[1] + 0.5 * n * d * math.log(2.0 * pi_const))

    def gather_dense_factor(self):
        """(tests, small n) the lower factor as one dense matrix on every rank."""
        lay = self.layout
        L = torch.zeros((lay.n, lay.n), dtype=torch.float64)
        for k in range(lay.npanels):
            k0, wk = lay.start(k), lay.width(k)
            blk = torch.zeros((lay.rows(k), wk), dtype=torch.float64)
            if lay.owner(k) == self.rank:
                blk.copy_(self.panels[k].cpu())
            if self.comm:
                root = dist.get_global_rank(self.group, lay.owner(k)) if self.group is not None else lay.owner(k)
                wire = blk.to(self.comm_dev)                    # (RCCL moves device buffers only)
                dist.broadcast(wire, src=root, group=self.group)
                blk = wire.cpu()
            L[k0:, k0:k0 + wk] = blk
        return torch.tril(L)


def tiled_nll(X, Y, w, amp, diag_add, clamp=float("-inf"), nb=512, group=None, ops=None, pi_const=PI_TRUNC):
    """V1 negative log marginal likelihood of ONE block spread over the ranks of `group`.  Raises LinAlgError on every rank
    if Sigma is not positive definite (the index is the single-GPU path's: first non-positive pivot, 1-based)."""
    tc = TiledCholesky(X.shape[0], nb=nb, group=group, ops=ops)
    tc.assemble(X, w, amp, diag_add, clamp)
    rc = tc.factor()
    if rc > 0:
        raise torch.linalg.LinAlgError("linalg.cholesky: the leading minor of order %d is not positive-definite" % rc)
    return tc.nll_v1(Y, pi_const)
