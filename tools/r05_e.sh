#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r05e
mkdir -p $OUT
cd $ROOT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log
tail -8 $OUT/pytest.log
timeout -k 10 300 python tools/ragged_probe.py 300,300,250 1 > $OUT/ragged.txt 2>&1
timeout -k 10 300 python tools/ragged_probe.py 8192,4096,2048,1024 1 >> $OUT/ragged.txt 2>&1
cat $OUT/ragged.txt
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"
python - <<'PY'
import json
o=json.loads([l for l in open("gpurun_out/r05e/bench_default.json") if l.startswith("{")][0])
print({k:o[k] for k in ("value","ms_per_step","ms_per_step_without_launch_events")})
print("roofline", {k:o["roofline"][k] for k in ("achieved","frac","avg_launch_ms","launches")})
print("sharded", {k:(v["ms_per_step"],v["value"]) for k,v in o.get("sharded",{}).items()})
print("train_step", {k:(v["ms_per_step"],v["frac_of_mfma_peak"]) for k,v in o.get("train_step",{}).items()})
print("stage_ms", o["stage_ms"])
PY
