#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r05j
mkdir -p $OUT
cd $ROOT
for N in 8192 4096 6144 2048; do
  echo "N=$N" >> $OUT/ab.txt
  AB_N=$N timeout -k 10 500 python tools/ab_forward.py "" "nb_outer=256" "nb_outer=768" "nb_outer=1024" "nb_outer=1024,la_min_n=1024" "nb_outer=2048" >> $OUT/ab.txt 2>&1
done
cat $OUT/ab.txt
