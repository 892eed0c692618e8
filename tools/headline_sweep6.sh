#!/bin/bash
# headline step against the outer block and the look-ahead thresholds after round 6's diagonal-block kernel: tools/headline_sweep6.sh
cd "$(dirname "$0")/.."
run() {
  python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-sharded "$@" 2>/dev/null | python3 -c "import json,sys; o=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', o['ms_per_step'], o['roofline']['frac'], o['roofline']['avg_launch_ms'], o['roofline']['launches'])"
}
for rep in 1 2; do
run
run --opt nb_outer=640
run --opt nb_outer=768
run --opt nb_outer=1024
run --opt la_carry_rows=6144
run --opt la_carry_rows=10240
run --opt polite_m=4096
run --opt polite_m=8192
done
