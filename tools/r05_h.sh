#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r05h
mkdir -p $OUT
cd $ROOT
for N in 8192 4096; do
  echo "N=$N" >> $OUT/ab.txt
  AB_N=$N timeout -k 10 400 python tools/ab_forward.py "" "polite_m=9000" "polite_m=9000,polite_pad_kb=17" "polite_pad_kb=17" "polite_pad_kb=24" "tile32_threshold=512" "polite_m=9000,tile32_threshold=512" "polite_m=0" >> $OUT/ab.txt 2>&1
done
cat $OUT/ab.txt
