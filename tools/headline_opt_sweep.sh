cd $GRAFT_REPO_ROOT
for i in 1 2; do for t in 0 6144 8192 10240 12288; do
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-sharded --opt la_carry_rows=$t 2>/dev/null | python3 -c "import json,sys; o=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('headline la_carry_rows=$t', o['ms_per_step'], o.get('ms_per_step_without_launch_events'), o['roofline']['frac'], o['roofline']['avg_launch_ms'], o['roofline']['avg_launch_gflop'])"
done; done
