#!/bin/bash
# round 5, first GPU call: baseline suite, ragged probe, chain timeline of one N = 8192, d = 1024 block
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r05a
mkdir -p $OUT
cd $ROOT
python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log
tail -3 $OUT/pytest.log
python tools/ragged_probe.py 8192,4096,2048,1024 1 > $OUT/ragged.txt 2>&1
python tools/ragged_probe.py 300,300,250 1 >> $OUT/ragged.txt 2>&1
python tools/ragged_probe.py 300,300,250 1 grad >> $OUT/ragged.txt 2>&1
python tools/ragged_probe.py 8192,4096,2048,1024 1 grad >> $OUT/ragged.txt 2>&1
cat $OUT/ragged.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $OUT/tr -o tr --output-format csv -- python3 $ROOT/bench.py --workload headline --n 8192 --D 8 --d 1024 --steps 3 --warmup 1 --no-cpu-baseline --no-sharded > $OUT/b8192.json 2> $OUT/b8192.err
cd $ROOT
python tools/chain_timeline.py "$OUT/tr/**/*kernel_trace.csv" 0 400 > $OUT/timeline_8192_d1024.txt 2>&1
python tools/gap_report.py "$OUT/tr/**/*kernel_trace.csv" > $OUT/gap_8192_d1024.txt 2>&1
find $OUT -name "*kernel_trace.csv" -delete
tail -2 $OUT/b8192.json | cut -c1-300
