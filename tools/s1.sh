cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s6
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_pipeline.py -x -q -k "bn_backward_sums or dgrad_staging or in_flight" 2>&1 | tail -15 > gpurun_out/s6/pytest.txt
YM_FORCE_STAGES=43 timeout 900 python -m pytest tests/test_gpu_train.py -x -q -k "matches_reference_golden or 256_well or losses_128" 2>&1 | tail -15 > gpurun_out/s6/pytest_force43.txt
timeout 900 python tools/pers_bench.py train > gpurun_out/s6/pers_train.txt 2>&1
