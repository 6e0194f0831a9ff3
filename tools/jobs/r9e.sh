set -x
mkdir -p gpurun_out/r9e
timeout 2400 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -8 | tee gpurun_out/r9e/pytest_gpu.txt
export GPU_MAX_HW_QUEUES=4
for S in 320 384 480; do
  python tools/train_size_bench.py --size $S --tag nearest_v2 2>/dev/null | tail -1 >> gpurun_out/r9e/train_sizes.jsonl
done
YM_TUNED_NEAREST=0 python tools/train_size_bench.py --size 384 --tag heuristic 2>/dev/null | tail -1 >> gpurun_out/r9e/train_sizes.jsonl
YM_TUNED_NEAREST=0 python tools/train_size_bench.py --size 480 --tag heuristic 2>/dev/null | tail -1 >> gpurun_out/r9e/train_sizes.jsonl
cat gpurun_out/r9e/train_sizes.jsonl
