set -x
mkdir -p gpurun_out/r9a
export GPU_MAX_HW_QUEUES=8
python tools/size_bench.py --tag shipped --sizes 320,416,480,544,640,736 > gpurun_out/r9a/before.jsonl 2> gpurun_out/r9a/before.err
timeout 1500 python tools/autotune.py --cfgs res101_coco --sizes 320,416,480,640,736 --batches 1 --skip-known --out gpurun_out/r9a/tuned_sizes.json > gpurun_out/r9a/tune.log 2>&1
python - <<'PY'
import json
b = json.load(open('yolact_minimal_amd/tuned_gfx950.json'))
b.update(json.load(open('gpurun_out/r9a/tuned_sizes.json')))
json.dump(b, open('gpurun_out/r9a/tuned_merged.json', 'w'), indent=0, sort_keys=True)
PY
YM_TUNED_PATH=gpurun_out/r9a/tuned_merged.json python tools/size_bench.py --tag merged --sizes 320,416,480,544,640,736 > gpurun_out/r9a/after.jsonl 2> gpurun_out/r9a/after.err
cat gpurun_out/r9a/before.jsonl gpurun_out/r9a/after.jsonl
