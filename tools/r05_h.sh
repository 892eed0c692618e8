#!/bin/bash
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/r05h
mkdir -p $OUT
cd $ROOT
timeout -k 10 300 python3 -m pytest tests/test_gpu_kernels.py -q -x -k "potrf" > $OUT/pytest.log 2>&1
echo "pytest rc=$?"; tail -3 $OUT/pytest.log
cd /tmp && export TMPDIR=/tmp
for n in 2048; do
for t in 0 1 2 4 15; do
  rocprofv3 --kernel-trace --stats -d $OUT/n${n}_f$t -o st --output-format csv -- python3 $ROOT/bench.py --n $n --D 16 --d 1 --steps 5 --warmup 2 --no-cpu-baseline --no-sharded --opt trsm128_dbg=$t > $OUT/n${n}_f$t.json 2> $OUT/n${n}_f$t.err
  find $OUT/n${n}_f$t -name "*kernel_stats.csv" -exec cp {} $OUT/stats_n${n}_f$t.csv \;
  find $OUT -name "*kernel_trace.csv" -delete
  echo "dbg=$t $(grep trsm128 $OUT/stats_n${n}_f$t.csv | cut -d, -f9-14)"
done
done
