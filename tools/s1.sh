cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s7
timeout 1500 python -m pytest tests/test_gpu_train_fullsize.py tests/test_gpu_pipeline.py -x -q -s 2>&1 | tail -30 > gpurun_out/s7/pytest_new.txt
timeout 900 python tools/pers_bench.py bs8 --write > gpurun_out/s7/pers_bs8.txt 2>&1
timeout 900 python tools/pers_bench.py train --write > gpurun_out/s7/pers_train.txt 2>&1
timeout 1500 python -m pytest tests/test_gpu_forward.py tests/test_gpu_train.py -x -q -k "544 or digest or golden" 2>&1 | tail -15 > gpurun_out/s7/pytest_table.txt
timeout 1200 python bench.py --steps 100 > gpurun_out/s7/bench_full.txt 2>&1
