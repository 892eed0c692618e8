#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r05c
mkdir -p $OUT
cd $ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_train.py -x -q > $OUT/pytest_train.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_train.log
tail -25 $OUT/pytest_train.log
timeout -k 10 300 python tools/train_bench.py 200 > $OUT/train_bench.txt 2>&1
cat $OUT/train_bench.txt
