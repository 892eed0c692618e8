#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r05p
mkdir -p $OUT
cd $ROOT
FFGP_TEST_NOISE=1 timeout -k 10 1100 python -m pytest tests -m gpu -q > $OUT/pytest_noise.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_noise.log
tail -6 $OUT/pytest_noise.log
