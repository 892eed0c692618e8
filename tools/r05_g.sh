#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05g
mkdir -p $OUT
{
for w in cigar4 gar8; do for t in 0 4096 8192 16384 32768; do
  python3 bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline --opt trsm128_max_m=$t 2>/dev/null | python3 -c "import json,sys; o=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w max_m=$t', o['ms_per_step'])"
done; done
for t in 0 4096 8192 12288 16384; do
  python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-sharded --opt trsm128_max_m=$t 2>/dev/null | python3 -c "import json,sys; o=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c3 max_m=$t', o['ms_per_step'])"
  python3 bench.py --n 8192 --D 8 --d 1024 --steps 10 --warmup 3 --no-cpu-baseline --no-sharded --opt trsm128_max_m=$t 2>/dev/null | python3 -c "import json,sys; o=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c4 block max_m=$t', o['ms_per_step'])"
done
} > $OUT/ab4.txt 2>&1
cat $OUT/ab4.txt
