#!/bin/bash
# C2 / N = 8192 forward against a few chain options after the diagonal-block kernel changed (round 6): tools/c2_sweep.sh
cd "$(dirname "$0")/.."
run() {
  w=$1; shift
  python3 bench.py --workload c2 $w --steps 40 --warmup 10 --no-cpu-baseline --no-sharded "$@" 2>/dev/null | python3 -c "import json,sys; o=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w $*', o['ms_per_step'])"
}
for rep in 1 2; do
for w in "" "--n 8192"; do
run "$w"
run "$w" --opt nb_outer=256
run "$w" --opt nb_outer=384
run "$w" --opt diag_excl_rows=0
run "$w" --opt polite32_pad_kb=0
run "$w" --opt trsm128=0
done
done
