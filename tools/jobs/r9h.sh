set -x
mkdir -p gpurun_out/r9h
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_forward.py -q -m gpu -k "autotune or other_image" 2>&1 | tail -8 | tee gpurun_out/r9h/pytest_new.txt
timeout 1500 python bench.py --steps 20 --warmup 5 > gpurun_out/r9h/bench_driver_args.json 2> gpurun_out/r9h/bench.err
tail -c 600 gpurun_out/r9h/bench_driver_args.json
