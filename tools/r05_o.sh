#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r05o
mkdir -p $OUT
cd $ROOT
timeout -k 10 300 python tools/train_bench.py 200 > $OUT/train_bench.txt 2>&1
cat $OUT/train_bench.txt
timeout -k 10 1000 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log
tail -6 $OUT/pytest.log
