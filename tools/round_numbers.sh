#!/bin/bash
# Every number DESIGN.md quotes for the round, in one GPU call.  Usage: bash tools/round_numbers.sh <tag>
TAG=${1:-numbers}
PART=${2:-all}      # bench | tools | all  (a gpurun call is limited to 20 minutes: the two halves fit one call each)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
if [ "$PART" != "tools" ]; then
python3 bench.py --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench.err
python3 bench.py --workload c2 --steps 50 --warmup 10 --no-cpu-baseline > $OUT/bench_c2.json 2>> $OUT/bench.err
python3 bench.py --with-grad --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_c3_grad.json 2>> $OUT/bench.err
python3 bench.py --workload c2 --with-grad --steps 50 --warmup 10 --no-cpu-baseline > $OUT/bench_c2_grad.json 2>> $OUT/bench.err
python3 bench.py --workload cigar4 --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_cigar4.json 2>> $OUT/bench.err
python3 bench.py --workload gar8 --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_gar8.json 2>> $OUT/bench.err
python3 bench.py --workload cigar4 --with-grad --steps 6 --warmup 2 --no-cpu-baseline > $OUT/bench_cigar4_grad.json 2>> $OUT/bench.err
python3 bench.py --n 8192 --D 8 --d 1024 --steps 10 --warmup 3 --no-cpu-baseline --no-sharded > $OUT/bench_c4_block.json 2>> $OUT/bench.err
python3 bench.py --n 8192 --D 8 --d 4096 --steps 10 --warmup 3 --no-cpu-baseline --no-sharded > $OUT/bench_c5_block.json 2>> $OUT/bench.err
python3 bench.py --n 32768 --steps 3 --warmup 1 --no-cpu-baseline --no-sharded > $OUT/bench_n32768.json 2>> $OUT/bench.err
python3 bench.py --workload gar8_hogp --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_gar8_hogp.json 2>> $OUT/bench.err
fi
if [ "$PART" != "bench" ]; then
# (the development library -- switches of rejected experiments that diag_bench / eigh_bench / raw_graph_bench A/B -- does not travel to
#  the box: built here, ~1 min; tools/_devlib.py picks it up)
make -C fidelityfusion_amd/csrc -j16 dev > $OUT/make_dev.log 2>&1
{
echo "## tools/train_bench.py 200"; timeout 300 python3 tools/train_bench.py 200 2>&1 | grep -v amdgpu.ids | tail -8
for sz in "300,300,250 1" "300,300,250 1 grad" "8192,4096,2048,1024 1" "8192,4096,2048,1024 1 grad" "8192,4096,2048,1024 256"; do
  echo "## tools/ragged_probe.py $sz"; timeout 300 python3 tools/ragged_probe.py $sz 2>&1 | grep -v amdgpu.ids | tail -7
done
echo "## tools/batch_chain_bench.py 4096 8"; timeout 300 python3 tools/batch_chain_bench.py 4096 8 2>&1 | grep -v amdgpu.ids | tail -8
for t in posterior_bench hogp_bench c4_step small_n_latency small_kernel_bench v2_bench diag_bench assemble_bench chase_dbg step_stages raw_graph_bench; do
  echo "## tools/$t.py"; timeout 300 python3 tools/$t.py 2>&1 | grep -v amdgpu.ids | tail -12
done
echo "## tools/eigh_bench.py full big"; timeout 600 python3 tools/eigh_bench.py full big 2>&1 | grep -v amdgpu.ids | tail -12
echo "## tools/pair_bench.py 16384 16"; timeout 300 python3 tools/pair_bench.py 16384 16 2>&1 | grep -v amdgpu.ids | tail -8
echo "## tools/small_n_breakdown.py 128 400"; timeout 300 python3 tools/small_n_breakdown.py 128 400 2>&1 | grep -v amdgpu.ids | tail -4
} > $OUT/tools_output.txt
fi
for f in $OUT/bench_*.json; do python3 - "$f" <<'P'
import json, sys
try:
    o = json.load(open(sys.argv[1]))
    print(sys.argv[1].split('/')[-1], o["ms_per_step"], "ms", o["value"], o["unit"], "|", o["config"]["workload"][:90])
except Exception as e:
    print(sys.argv[1], "unreadable", e)
P
done
