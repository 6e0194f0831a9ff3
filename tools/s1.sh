cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s22
timeout 900 python tools/pers_bench.py bs8 --write > gpurun_out/s22/pers_bs8.txt 2>&1
timeout 1500 python -m pytest tests/test_gpu_forward.py tests/test_gpu_swin.py -x -q 2>&1 | tail -5 > gpurun_out/s22/pytest_fwd.txt
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/s22/bench.txt 2>&1
