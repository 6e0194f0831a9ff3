set -x
mkdir -p gpurun_out/r9o
export GPU_MAX_HW_QUEUES=8
timeout 1500 python tools/autotune.py --cfgs res101_coco,res50_coco,swin_tiny_coco --sizes 320,416,480,640,736 --batches 8 --skip-known --out gpurun_out/r9o/tuned_bs8.json > gpurun_out/r9o/tune_bs8.log 2>&1
tail -2 gpurun_out/r9o/tune_bs8.log
timeout 1200 python tools/autotune.py --cfgs swin_tiny_coco --sizes 256,320,384,448,512,576,640,704,768 --batches 1 --skip-known --out gpurun_out/r9o/tuned_swin1.json > gpurun_out/r9o/tune_swin1.log 2>&1
tail -2 gpurun_out/r9o/tune_swin1.log
python - <<'PY'
import json
b = json.load(open('yolact_minimal_amd/tuned_gfx950.json'))
n0 = len(b)
for f in ('gpurun_out/r9o/tuned_bs8.json', 'gpurun_out/r9o/tuned_swin1.json'):
    try:
        for k, v in json.load(open(f)).items():
            b.setdefault(k, v)
    except Exception as e:
        print('skip', f, e)
json.dump(b, open('gpurun_out/r9o/tuned_merged.json', 'w'), indent=0, sort_keys=True)
print(n0, len(b))
PY
for C in res101_coco swin_tiny_coco; do
python tools/size_bench.py --cfg $C --batch 8 --sizes 320,416,640,736 --steps 20 --tag before 2>/dev/null | cut -c1-160 >> gpurun_out/r9o/bs8.jsonl
YM_TUNED_PATH=gpurun_out/r9o/tuned_merged.json python tools/size_bench.py --cfg $C --batch 8 --sizes 320,416,640,736 --steps 20 --tag rows 2>/dev/null | cut -c1-160 >> gpurun_out/r9o/bs8.jsonl
YM_TUNED_NEAREST=0 python tools/size_bench.py --cfg $C --batch 8 --sizes 320,416,640,736 --steps 20 --tag heuristic 2>/dev/null | cut -c1-160 >> gpurun_out/r9o/bs8.jsonl
done
python tools/size_bench.py --cfg swin_tiny_coco --batch 1 --sizes 320,448,640 --tag before 2>/dev/null | cut -c1-160 >> gpurun_out/r9o/bs8.jsonl
YM_TUNED_PATH=gpurun_out/r9o/tuned_merged.json python tools/size_bench.py --cfg swin_tiny_coco --batch 1 --sizes 320,448,640 --tag rows 2>/dev/null | cut -c1-160 >> gpurun_out/r9o/bs8.jsonl
cat gpurun_out/r9o/bs8.jsonl
