"""Does the ORDER of workloads inside one process change their speed?  Runs bench.py workloads one after another in this process and
prints each one's ms/step.  usage: python tools/order_probe.py cigar4:5:2,gar8_hogp:3:1   (workload:steps:warmup; also sleepN,
streamsN = touch N fresh torch streams, emptycache, reserve = _lib.reserve_block_streams).  This is how the slow `sharded.gar8_hogp` leg
of the default run was traced to hardware-queue assignment by first use (docs/concurrency.md)."""
import io, json, sys, time, runpy, contextlib, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
def run(args):
    sys.argv = ["bench.py"] + args + ["--no-cpu-baseline", "--no-sharded"]
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        try:
            runpy.run_path(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"), run_name="__main__")
        except SystemExit:
            pass
    o = json.loads(buf.getvalue().strip().splitlines()[-1])
    return o["ms_per_step"]
seq = sys.argv[1].split(",")
for item in seq:
    if item.startswith("streams"):
        import torch
        k = int(item[7:]); ss = [torch.cuda.Stream(0) for _ in range(k)]
        x = torch.ones(1024, device="cuda:0")
        for st in ss:
            with torch.cuda.stream(st):
                x.add_(1.0)
        torch.cuda.synchronize(); print("  (touched %d fresh torch streams)" % k, flush=True); continue
    if item == "reserve":
        from fidelityfusion_amd import functional as F
        F.reserve_block_streams(0, 4); print("  (worker streams reserved explicitly)", flush=True); continue
    if item == "emptycache":
        import torch
        torch.cuda.synchronize(); torch.cuda.empty_cache(); print("  (empty_cache)", flush=True); continue
    if item.startswith("sleep"):
        time.sleep(float(item[5:])); print("  (idle %s s)" % item[5:], flush=True); continue
    wl, steps, warm = item.split(":")
    print("  %-10s %8.1f ms/step" % (wl, run(["--workload", wl, "--steps", steps, "--warmup", warm])), flush=True)
