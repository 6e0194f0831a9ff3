set -x
mkdir -p gpurun_out/r9f
cd /tmp && mkdir -p rl && cd rl
python $GRAFT_REPO_ROOT/tests/run_reference_loop.py --cfg swin_tiny_coco --img_size 128 --train_bs 2 --steps 3 --seed 73 --out /tmp/rl/x.npz --no_drop_path > $GRAFT_REPO_ROOT/gpurun_out/r9f/swin_default.out 2> $GRAFT_REPO_ROOT/gpurun_out/r9f/swin_default.err
echo rc=$?
YM_TUNED_NEAREST=0 python $GRAFT_REPO_ROOT/tests/run_reference_loop.py --cfg swin_tiny_coco --img_size 128 --train_bs 2 --steps 3 --seed 73 --out /tmp/rl/y.npz --no_drop_path > $GRAFT_REPO_ROOT/gpurun_out/r9f/swin_off.out 2> $GRAFT_REPO_ROOT/gpurun_out/r9f/swin_off.err
echo rc=$?
cd $GRAFT_REPO_ROOT
grep -v "^\[W\|Warning\|warn" gpurun_out/r9f/swin_default.err | tail -30
timeout 2400 python -m pytest tests/ -q -m gpu 2>&1 | tail -15 | tee gpurun_out/r9f/pytest_gpu.txt
