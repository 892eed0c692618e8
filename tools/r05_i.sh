#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r05i
mkdir -p $OUT
cd $ROOT
for N in 8192 4096; do
  echo "N=$N" >> $OUT/ab.txt
  AB_N=$N timeout -k 10 500 python tools/ab_forward.py "" "tail_mask_m=4096" "tail_mask_m=6144" "tail_mask_m=9000" "tail_mask_m=6144,tail_mask_cus=4" "tail_mask_m=6144,tail_mask_cus=12" "tail_mask_m=9000,tail_mask_cus=4" "tail_mask_m=9000,tail_mask_cus=16" >> $OUT/ab.txt 2>&1
done
cat $OUT/ab.txt
