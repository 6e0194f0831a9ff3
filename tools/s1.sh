cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s23
timeout 2400 python -m pytest tests -q -m gpu --durations=8 2>&1 | tail -20 > gpurun_out/s23/pytest_all.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/s23/smoke.txt 2>&1
bash tools/profile_round.sh r03 > gpurun_out/profile_r03.log 2>&1
bash tools/pmc_round.sh r03 > gpurun_out/pmc_r03.log 2>&1
timeout 1500 python bench.py > gpurun_out/s23/bench_full.txt 2>&1
