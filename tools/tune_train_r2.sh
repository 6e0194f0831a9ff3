python - <<'PY'
import json
t = json.load(open('yolact_minimal_amd/tuned_gfx950.json'))
drop = [k for k in t if k.startswith('W_') and '_N64_' in k]
for k in drop: del t[k]
json.dump(t, open('gpurun_out/tuned_base.json', 'w'), indent=0, sort_keys=True)
print('dropped', drop)
PY
YM_TUNED_PATH=gpurun_out/tuned_base.json python tools/autotune_train.py --cfgs res101_coco,res50_coco --batch 8 --out gpurun_out/tuned_T8.json > gpurun_out/tune_T8.log 2>&1; tail -2 gpurun_out/tune_T8.log
YM_TUNED_PATH=gpurun_out/tuned_base.json python tools/autotune_train.py --cfgs res101_coco --batch 16 --out gpurun_out/tuned_T16.json > gpurun_out/tune_T16.log 2>&1; tail -2 gpurun_out/tune_T16.log
