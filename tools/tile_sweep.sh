#!/bin/bash
# headline step against the tile-shape thresholds of the chain's K = 128 products (round 6): tools/tile_sweep.sh
cd "$(dirname "$0")/.."
run() {
  python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-sharded "$@" 2>/dev/null | python3 -c "import json,sys; o=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', o['ms_per_step'], o['roofline']['frac'], o['roofline']['avg_launch_ms'])"
}
for rep in 1 2; do
run
run --opt tile32_threshold=0
run --opt tile32_threshold=256
run --opt tile32_threshold=512
run --opt small_tile_threshold=320
run --opt small_tile_threshold=320 --opt tile32_threshold=256
run --opt small_tile_threshold=160 --opt tile32_threshold=0
done
