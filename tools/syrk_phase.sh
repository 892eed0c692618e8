#!/bin/bash
# Review item "trailing update in situ": kernel trace of the headline forward -> tools/syrk_phase_account.py, then the SYRK alone at the
# same 22 sizes.  Usage (GPU box): bash tools/syrk_phase.sh <tag>
TAG=${1:-syrk_phase}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace -d $OUT/trace -o t --output-format csv -- python3 $ROOT/tools/forward_trace_target.py 16384 16 5 > $OUT/trace.log 2>&1 || exit 1
cd $ROOT
{
python3 tools/syrk_phase_account.py "$OUT/trace/**/*kernel_trace.csv" 16384 512
echo
python3 tools/gap_report.py "$OUT/trace/**/*kernel_trace.csv"
echo
timeout -k 10 300 python3 tools/syrk_standalone.py 16384 512 22 2>&1 | grep -v amdgpu.ids
} > $OUT/syrk_phase.txt 2>&1
rm -rf $OUT/trace
tail -50 $OUT/syrk_phase.txt
