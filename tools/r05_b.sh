#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r05b
mkdir -p $OUT
cd $ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "ragged or falls_back or equal_shape or gradient_stage" > $OUT/pytest_ragged.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_ragged.log
tail -15 $OUT/pytest_ragged.log
timeout -k 10 300 python tools/ragged_probe.py 8192,4096,2048,1024 1 > $OUT/ragged.txt 2>&1
timeout -k 10 300 python tools/ragged_probe.py 300,300,250 1 >> $OUT/ragged.txt 2>&1
timeout -k 10 300 python tools/ragged_probe.py 300,300,250 1 grad >> $OUT/ragged.txt 2>&1
timeout -k 10 300 python tools/ragged_probe.py 8192,4096,2048,1024 1 grad >> $OUT/ragged.txt 2>&1
cat $OUT/ragged.txt
