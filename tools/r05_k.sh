#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r05k
mkdir -p $OUT
cd $ROOT
for N in 16384 8192; do
  echo "N=$N" >> $OUT/ab.txt
  AB_N=$N timeout -k 10 500 python tools/ab_forward.py "" "syrk_h64=1" "syrk_h64=1,polite_m=0" >> $OUT/ab.txt 2>&1
done
cat $OUT/ab.txt
python tools/gemm_bench.py syrk > $OUT/syrk.txt 2>&1; tail -20 $OUT/syrk.txt
