#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r05d
mkdir -p $OUT
cd $ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_train.py -x -q -k "ragged or falls_back or equal_shape or gradient_stage or full_size_blocks or train_many" > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log
tail -15 $OUT/pytest.log
timeout -k 10 300 python tools/ragged_probe.py 300,300,250 1 > $OUT/ragged.txt 2>&1
timeout -k 10 300 python tools/ragged_probe.py 8192,4096,2048,1024 1 >> $OUT/ragged.txt 2>&1
timeout -k 10 300 python tools/ragged_probe.py 300,300,250 1 grad >> $OUT/ragged.txt 2>&1
cat $OUT/ragged.txt
timeout -k 10 300 python tools/train_bench.py 200 > $OUT/train_bench.txt 2>&1
cat $OUT/train_bench.txt
timeout -k 10 300 python tools/batch_chain_bench.py 4096 8 > $OUT/batch8.txt 2>&1
cat $OUT/batch8.txt
