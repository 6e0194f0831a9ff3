set -x
mkdir -p gpurun_out/r9j
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_forward.py tests/test_gpu_train_fullsize.py -q -m gpu 2>&1 | tail -6 | tee gpurun_out/r9j/pytest.txt
timeout 1500 python bench.py --steps 20 --warmup 5 > gpurun_out/r9j/bench_driver_args.json 2> gpurun_out/r9j/bench.err
timeout 1500 python bench.py > gpurun_out/r9j/bench_default.json 2> gpurun_out/r9j/bench2.err
tail -c 300 gpurun_out/r9j/bench_default.json
