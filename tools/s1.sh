cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s14
timeout 1500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_train.py -x -q -k "persistent or race_free or bn_backward_sums or dgrad_staging or golden" 2>&1 | tail -6 > gpurun_out/s14/pytest.txt
timeout 900 python bench.py --no-extra --no-cpu-baseline > gpurun_out/s14/bench.txt 2>&1
R="$(pwd)"; OUT="$R/gpurun_out/profiles_r03"; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/raw_train
rocprofv3 --kernel-trace -d $OUT/raw_train -o train -- python $R/tools/train_profile.py --steps 10 > $OUT/train.stdout 2> $OUT/train.stderr
db=$(find $OUT/raw_train -name '*.db' | head -1)
[ -n "$db" ] && python $R/tools/prof_summary.py "$db" "$OUT/r03_train_res101_bs8_kernel_stats.md" > /dev/null && python $R/tools/gap_summary.py "$db" 30 k_sgd > "$OUT/r03_train_res101_bs8_gaps.txt"
rm -rf $OUT/raw_train
C="SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
env YM_WGRAD_STREAM=0 rocprofv3 --kernel-trace --pmc $C -d $OUT/raw_pt -o p --output-format csv -- python $R/tools/train_profile.py --steps 4 > /dev/null 2>&1
f=$(find $OUT/raw_pt -name '*counter_collection.csv' | head -1)
[ -n "$f" ] && python $R/tools/pmc_mfma_summary.py "$f" "$OUT/r03_pmc_mfma_train_bs8_res101.json" "rocprofv3 --pmc $C -- YM_WGRAD_STREAM=0 python tools/train_profile.py --steps 4"
rm -rf $OUT/raw_pt
