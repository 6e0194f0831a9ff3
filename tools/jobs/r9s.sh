set -x
mkdir -p gpurun_out/r9s
timeout 2400 python -m pytest tests/ -q -m gpu 2>&1 | tail -6 | tee gpurun_out/r9s/pytest_gpu.txt
timeout 1500 python bench.py --steps 20 --warmup 5 > gpurun_out/r9s/bench_driver_args.json 2> gpurun_out/r9s/bench1.err
timeout 1500 python bench.py > gpurun_out/r9s/bench_default.json 2> gpurun_out/r9s/bench2.err
bash tools/profile_round.sh r06 > gpurun_out/r9s/profile_round.log 2>&1
bash tools/pmc_round.sh r06 > gpurun_out/r9s/pmc_round.log 2>&1
ls gpurun_out/profiles_r06 | head -50
