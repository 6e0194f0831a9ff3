set -x
mkdir -p gpurun_out/r9c
export GPU_MAX_HW_QUEUES=8
export YM_TUNED_PATH=tools/jobs/tuned_cand.json
ALL=256,288,320,352,384,416,448,480,512,544,576,608,640,672,704,736,768,800
python tools/size_bench.py --tag cand --sizes $ALL > gpurun_out/r9c/cand.jsonl 2> gpurun_out/r9c/err1.txt
cp tools/jobs/tuned_cand.json gpurun_out/r9c/tuned_work.json
export YM_TUNED_PATH=gpurun_out/r9c/tuned_work.json
for S in 256 288 320 352 384 416 448 480 512 576 608 640 672 704 736 768 800; do
  timeout 300 python tools/tune_forward.py --size $S --alt yolact_minimal_amd/tuned_gfx950.json --max-m 80000 --out gpurun_out/r9c/fw_$S.json > gpurun_out/r9c/fw_$S.log 2>&1
  tail -1 gpurun_out/r9c/fw_$S.log
  python - <<PY
import json, os
f = 'gpurun_out/r9c/fw_$S.json'
if os.path.exists(f):
    b = json.load(open('gpurun_out/r9c/tuned_work.json'))
    b.update(json.load(open(f)))
    json.dump(b, open('gpurun_out/r9c/tuned_work.json', 'w'), indent=0, sort_keys=True)
PY
done
python tools/size_bench.py --tag final --sizes $ALL > gpurun_out/r9c/final.jsonl 2> gpurun_out/r9c/err2.txt
python tools/size_bench.py --tag final_res50 --cfg res50_coco --sizes 320,416,544,640,736 > gpurun_out/r9c/final_res50.jsonl 2> gpurun_out/r9c/err3.txt
cat gpurun_out/r9c/final.jsonl | cut -c1-150
timeout 900 python -m pytest tests/test_gpu_forward.py -x -q 2>&1 | tail -5 | tee gpurun_out/r9c/pytest.txt
