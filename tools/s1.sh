cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s2
timeout 900 python -m pytest tests/test_gpu_train.py -q -k "side_stream" 2>&1 | tail -15 > gpurun_out/s2/pytest.txt
timeout 300 python tools/chain_trace.py 1 34 3 > gpurun_out/s2/chain_bs1_34.txt 2>&1
timeout 300 python tools/chain_trace.py 8 34 2 > gpurun_out/s2/chain_bs8_34.txt 2>&1
timeout 300 python tools/mem_check.py > gpurun_out/s2/mem.txt 2>&1
tail -5 gpurun_out/s2/pytest.txt gpurun_out/s2/mem.txt
