set -x
mkdir -p gpurun_out/r9b
export GPU_MAX_HW_QUEUES=8
S=320,416,480,544,640,736
python tools/size_bench.py --tag shipped+nearest --sizes $S > gpurun_out/r9b/shipped_nearest.jsonl 2> gpurun_out/r9b/err1.txt
YM_TUNED_PATH=tools/jobs/tuned_merged_r9a.json YM_TUNED_NEAREST=only python tools/size_bench.py --tag merged_leave_one_out --sizes $S > gpurun_out/r9b/loo.jsonl 2> gpurun_out/r9b/err2.txt
YM_TUNED_PATH=tools/jobs/tuned_merged_r9a.json python tools/size_bench.py --tag merged+nearest --sizes 288,352,384,448,512,576,608,672,704,768,800 > gpurun_out/r9b/merged_other.jsonl 2> gpurun_out/r9b/err3.txt
YM_TUNED_NEAREST=0 python tools/size_bench.py --tag shipped_heuristic --sizes 288,384,512,608,704,800 > gpurun_out/r9b/heur_other.jsonl 2> gpurun_out/r9b/err4.txt
cat gpurun_out/r9b/*.jsonl
timeout 600 python -m pytest tests/test_gpu_forward.py -x -q -k "other_image_sizes or matches_golden" 2>&1 | tail -15 | tee gpurun_out/r9b/pytest.txt
YM_TUNED_PATH=tools/jobs/tuned_merged_r9a.json timeout 1500 python tools/autotune.py --cfgs res101_coco,res50_coco --sizes 256,288,320,352,384,416,448,480,512,576,608,640,672,704,736,768,800 --batches 1 --skip-known --out gpurun_out/r9b/tuned_sizes2.json > gpurun_out/r9b/tune.log 2>&1
tail -3 gpurun_out/r9b/tune.log
