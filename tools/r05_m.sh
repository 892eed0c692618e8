#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r05m
mkdir -p $OUT
cd $ROOT
for N in 16384 8192; do
  echo "N=$N" >> $OUT/ab.txt
  AB_N=$N timeout -k 10 500 python tools/ab_forward.py "" "syrk_direct=1" >> $OUT/ab.txt 2>&1
done
cat $OUT/ab.txt
