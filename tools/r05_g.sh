#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r05g
mkdir -p $OUT
cd $ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_kernels.py -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log
tail -8 $OUT/pytest.log
for d in 1024 4096; do
  for ps in 0 256; do
    echo "N=8192 d=$d pass_split_min=$ps" >> $OUT/blocks.txt
    timeout -k 10 300 python bench.py --workload headline --n 8192 --D 8 --d $d --steps 10 --warmup 3 --no-cpu-baseline --no-sharded --opt pass_split_min=$ps 2>/dev/null | python -c "import sys,json; o=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(o['ms_per_step'], o['value'], o['whole_path_frac_of_mfma_peak'], o['nll'])" >> $OUT/blocks.txt
  done
done
for wl in cigar4 gar8; do
  for ps in 0 256; do
    echo "$wl pass_split_min=$ps" >> $OUT/blocks.txt
    timeout -k 10 300 python bench.py --workload $wl --steps 6 --warmup 3 --no-cpu-baseline --opt pass_split_min=$ps 2>/dev/null | python -c "import sys,json; o=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(o['ms_per_step'], o['value'], o['whole_path_frac_of_mfma_peak'], o['joint_nll'])" >> $OUT/blocks.txt
  done
done
cat $OUT/blocks.txt
