set -x
mkdir -p gpurun_out/r9d
timeout 1500 python -m pytest tests/test_gpu_forward.py tests/test_gpu_postproc.py tests/test_gpu_ops.py -x -q 2>&1 | tail -8 | tee gpurun_out/r9d/pytest_fwd.txt
export GPU_MAX_HW_QUEUES=4
for S in 320 416 640; do
  YM_TUNED_NEAREST=0 python tools/train_size_bench.py --size $S --tag heuristic 2>/dev/null | tail -1 >> gpurun_out/r9d/train_sizes.jsonl
  python tools/train_size_bench.py --size $S --tag nearest 2>/dev/null | tail -1 >> gpurun_out/r9d/train_sizes.jsonl
done
python tools/train_size_bench.py --size 544 --tag table 2>/dev/null | tail -1 >> gpurun_out/r9d/train_sizes.jsonl
YM_TUNED_NEAREST=only python tools/train_size_bench.py --size 544 --tag transfers_only 2>/dev/null | tail -1 >> gpurun_out/r9d/train_sizes.jsonl
YM_NO_TUNED=1 python tools/train_size_bench.py --size 544 --tag no_table 2>/dev/null | tail -1 >> gpurun_out/r9d/train_sizes.jsonl
cat gpurun_out/r9d/train_sizes.jsonl
