"""One fwd+grad call of negative_log_likelihood_many on (300, 300, 250) under rocprofv3 --kernel-trace: prints the kernels of the LAST
call with start offsets and durations (us).  rocprofv3 --kernel-trace -d out -o t --output-format csv -- python3 tools/ragged_grad_trace.py;
python3 tools/ragged_grad_trace.py --report out/t_kernel_trace.csv"""
import csv
import os
import sys

if len(sys.argv) > 2 and sys.argv[1] == "--report":
    rows = list(csv.DictReader(open(sys.argv[2])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # the last call: everything after the marker kernel (a torch.cumsum)
    idx = max(i for i, r in enumerate(rows) if "scan" in r["Kernel_Name"].lower() or "cumsum" in r["Kernel_Name"].lower())
    last = rows[idx + 1:]
    t0 = int(last[0]["Start_Timestamp"])
    prev_end = t0
    for r in last:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        print("%9.1f  +gap %6.1f  dur %7.1f  %s" % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, r["Kernel_Name"][:90]))
        prev_end = max(prev_end, e)
    print("kernels %d, span %.1f us" % (len(last), (prev_end - t0) / 1e3))
    sys.exit(0)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from bench import synthetic_xy
from fidelityfusion_amd import kernel
from fidelityfusion_amd.cigp_v10 import cigp, negative_log_likelihood_many

torch.set_default_dtype(torch.float64)
dev = torch.device("cuda", 0)
models, xs, ys = [], [], []
for f, n in enumerate((300, 300, 250)):
    X, Y = synthetic_xy(n, 8, 1, seed=f)
    models.append(cigp(kernel.ARDKernel(8), 1.0).to(dev))
    xs.append(torch.tensor(X, device=dev))
    ys.append(torch.tensor(Y, device=dev))
for rep in range(4):
    if rep == 3:
        torch.cuda.synchronize()
        marker = torch.cumsum(torch.ones(4096, device=dev), 0)      # (a kernel name nothing else in the run has)
        torch.cuda.synchronize()
    vals = negative_log_likelihood_many(models, xs, ys)
    vals.sum().backward()
torch.cuda.synchronize()
