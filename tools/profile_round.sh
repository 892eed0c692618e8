#!/bin/bash
# Round profile on the GPU box: kernel-trace stats of the default bench, then the PMC passes (counters in their own
# runs, never combined with tracing domains; under counter collection the library's handles fall back to event hand-offs by themselves:
# rocprofv3 --pmc serialises the dispatches of all queues, and a value wait is a polling kernel -- see include/ffgp.h, "ho_values").  Usage: bash tools/profile_round.sh <tag>   (writes under gpurun_out/<tag>/)
set -u
TAG=${1:-prof}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $ROOT/bench.py --steps 10 --warmup 2 > $OUT/bench_c3.json 2> $OUT/bench_c3.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/stats -o st --output-format csv -- python3 $ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-sharded > $OUT/stats_bench.json 2> $OUT/stats.err
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  name=$(echo $pass | cut -d' ' -f1)
  timeout -k 10 300 rocprofv3 --pmc $pass -d $OUT/pmc_$name -o pmc --output-format csv -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-sharded > /dev/null 2> $OUT/pmc_$name.err
done
cd $ROOT
python3 tools/pmc_summary.py $OUT/pmc_summary.json $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/pmc_SQ_VALU_MFMA_BUSY_CYCLES
cp $OUT/stats/*kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null || find $OUT/stats -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
find $OUT -name "*counter_collection.csv" -delete
find $OUT -name "*kernel_trace.csv" -delete
# the symmetric eigensolver at N = 8192 (two calls of ffgp_syevd)
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/stats_syevd -o st --output-format csv -- python3 $ROOT/tools/eigh_prof.py 8192 > $OUT/stats_syevd.log 2> $OUT/stats_syevd.err
cd $ROOT
find $OUT/stats_syevd -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_syevd.csv \;
find $OUT -name "*kernel_trace.csv" -delete
tail -1 $OUT/bench_c3.json | cut -c1-400
head -8 $OUT/kernel_stats.csv | cut -c1-160
