#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r05l
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -i -E "TCC_EA0?_RDREQ|TCC_EA0?_WRREQ|TCC_REQ|TCC_READ|FETCH_SIZE|WRITE_SIZE" | head -40 > $OUT/counters.txt
for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  name=$(echo $pass | cut -d' ' -f1)
  rocprofv3 --pmc $pass -d $OUT/pmc_$name -o pmc --output-format csv -- python3 $ROOT/tools/traffic_calib.py > $OUT/calib_$name.log 2> $OUT/calib_$name.err
done
cd $ROOT
python3 - <<'PY'
import csv, glob
for f in sorted(glob.glob("gpurun_out/r05l/pmc_*/**/*counter_collection.csv", recursive=True)):
    rows=[r for r in csv.DictReader(open(f)) if "ffgp_gemm_f64<0, 0, 1, 0, 128, 128>" in r["Kernel_Name"]]
    rows.sort(key=lambda r:int(r["Dispatch_Id"]))
    print(f.split("/")[2], [(r["Counter_Name"], r["Counter_Value"]) for r in rows])
PY
cat $OUT/counters.txt | head -30; tail -2 $OUT/calib_FETCH_SIZE.log
find $OUT -name "*counter_collection.csv" -size +2M -delete
