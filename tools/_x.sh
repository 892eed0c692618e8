cd $GRAFT_REPO_ROOT
timeout -k 10 600 python3 -m pytest tests/test_gpu_syevd.py tests/test_gpu_eigh.py -q -x 2>&1 | tail -3
python3 tools/eigh_bench.py full 2>&1 | grep -v amdgpu.ids | tail -8
