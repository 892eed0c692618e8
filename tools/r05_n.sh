#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r05n
mkdir -p $OUT
cd $ROOT
timeout -k 10 300 python tools/train_bench.py 200 > $OUT/train_default.txt 2>&1
FFGP_OPTS=small_finish=1 timeout -k 10 300 python tools/train_bench.py 200 > $OUT/train_small_finish.txt 2>&1
cat $OUT/train_default.txt $OUT/train_small_finish.txt
