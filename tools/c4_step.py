"""Config 4 on ONE GPU: a CIGAR-sized set of F = 4 independent blocks (N = 8192, D = 8, d = 1024), one Adam step per block
through sharding.ShardedTrainer -- sequential vs overlapped (functional.concurrent_blocks)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from fidelityfusion_amd import kernel
from fidelityfusion_amd.cigp_v10 import cigp
from fidelityfusion_amd.sharding import ShardedTrainer, block_cost

torch.set_default_dtype(torch.float64)
dev = "cuda:0"
F_, n, D, d = 4, int(sys.argv[1]) if len(sys.argv) > 1 else 8192, 8, int(sys.argv[2]) if len(sys.argv) > 2 else 1024
g = torch.Generator(device=dev).manual_seed(0)
data = [(torch.rand((n, D), generator=g, device=dev), torch.randn((n, d), generator=g, device=dev)) for _ in range(F_)]
for conc, nslots, la in ((False, 1, True), (True, 3, True), (True, 2, False), (True, 4, False)):
    tr = ShardedTrainer(lambda f: cigp(kernel.ARDKernel(D), 1.0).to(dev), data, [block_cost(n, d)] * F_, lr=1e-2, concurrent=conc,
                        nslots=nslots, slot_lookahead=la)
    for _ in range(2):
        tr.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        v = tr.step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 3 * 1e3
    flops = F_ * (n ** 3 + 2.0 * n * n * d + 4.0 * n * n * D)
    print("F=%d N=%d d=%d, one training step of every block, %s: %.1f ms (%.1f TFLOP/s on N^3 + 2N^2 d + 4N^2 D per block)"
          % (F_, n, d, ("overlapped, %d slots, look-ahead %s" % (nslots, "on" if la else "off")) if conc else "sequential", ms, flops / ms / 1e9))
