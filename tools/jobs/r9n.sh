R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r9n
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d /tmp/prof_miou -o miou -- python $R/tools/micro/mask_iou_probe.py > $R/gpurun_out/r9n/probe.out 2> $R/gpurun_out/r9n/probe.err
db=$(find /tmp/prof_miou -name '*.db' | head -1)
python $R/tools/prof_summary.py "$db" $R/gpurun_out/r9n/mask_iou_kernel_stats.md | head -4
cat $R/gpurun_out/r9n/probe.out
cd $R
python tools/micro/mask_iou_probe.py --n 37 --g 3
python tools/micro/mask_iou_probe.py --n 200 --g 140
timeout 600 python -m pytest tests/test_gpu_metrics.py -q 2>&1 | tail -3
