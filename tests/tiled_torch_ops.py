"""CPU test double of `fidelityfusion_amd.tiled.HipOps`: the four local operations of the multi-GPU tiled Cholesky on torch tensors,
injected by tests/test_tiled_gloo.py to prove the distributed algorithm on CPU over gloo (also a readable specification of
HipOps).  Test infrastructure only -- the product package has no CPU path."""
import torch


class TorchOps:
    """The four local operations on torch tensors (CPU tests; also a readable specification of HipOps)."""

    def __init__(self, device="cpu"):
        self.device = torch.device(device)

    def kernel_panel(self, Xr, Xc, w, amp, clamp):
        d = (Xr * w).unsqueeze(1) - (Xc * w).unsqueeze(0)
        return amp * torch.exp(-0.5 * torch.clamp((d * d).sum(-1), min=clamp))

    def potrf_rows(self, T, w):
        L, info = torch.linalg.cholesky_ex(T[:w, :w])
        if int(info) > 0:
            return int(info)
        T[:w, :w] = L
        if T.shape[0] > w:
            T[w:] = torch.linalg.solve_triangular(L, T[w:].T, upper=False).T
        return 0

    def update(self, C, A, B):
        """C -= A B^T"""
        C -= A @ B.T

    def trsm_lower(self, Lkk, B):
        B.copy_(torch.linalg.solve_triangular(torch.tril(Lkk), B, upper=False))

    def gemm_acc(self, Z, A, G):
        """Z += A G"""
        Z += A @ G
