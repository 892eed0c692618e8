R=$PWD; mkdir -p gpurun_out/r06y; cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace -d $R/gpurun_out/r06y/t -o t --output-format csv -- python3 $R/tools/forward_trace_target.py 8192 8 4 4096 > $R/gpurun_out/r06y/log.txt 2>&1 || exit 1
cd $R
python3 tools/queue_windows.py "gpurun_out/r06y/t/**/*kernel_trace.csv" 500 > gpurun_out/r06y/windows.txt
python3 tools/stream_busy.py "gpurun_out/r06y/t/**/*kernel_trace.csv" > gpurun_out/r06y/busy.txt
python3 tools/chain_timeline.py "gpurun_out/r06y/t/**/*kernel_trace.csv" 0 3000 > gpurun_out/r06y/timeline.txt
rm -rf gpurun_out/r06y/t
