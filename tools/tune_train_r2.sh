python - <<'PY'
import json
t = json.load(open('yolact_minimal_amd/tuned_gfx950.json'))
drop = [k for k in t if k.startswith('W_')]
for k in drop: del t[k]
json.dump(t, open('gpurun_out/tuned_base.json', 'w'), indent=0, sort_keys=True)
print('dropped', len(drop), 'wgrad entries')
PY
YM_TUNED_PATH=gpurun_out/tuned_base.json python tools/autotune_train.py --cfgs res101_coco,res50_coco,swin_tiny_coco --batch 8 --out gpurun_out/tuned_W8.json > gpurun_out/tune_W8.log 2>&1; tail -1 gpurun_out/tune_W8.log
YM_TUNED_PATH=gpurun_out/tuned_base.json python tools/autotune_train.py --cfgs res101_coco --batch 16 --out gpurun_out/tuned_W16.json > gpurun_out/tune_W16.log 2>&1; tail -1 gpurun_out/tune_W16.log
python - <<'PY'
import json
t = json.load(open('gpurun_out/tuned_base.json'))
t.update(json.load(open('gpurun_out/tuned_W8.json'))); t.update(json.load(open('gpurun_out/tuned_W16.json')))
json.dump(t, open('gpurun_out/tuned_wgrad_nb.json', 'w'), indent=0, sort_keys=True)
import collections
print(collections.Counter(v[1] for k, v in t.items() if k.startswith('W_') and len(v) > 1))
PY
python -m pytest tests/test_gpu_train.py -m gpu -q -k "grads or 256 or train_step_matches" 2>&1 | tail -2
YM_TUNED_PATH=gpurun_out/tuned_wgrad_nb.json python tools/train_profile.py --steps 6 2>&1 | tail -1
python tools/train_profile.py --steps 6 2>&1 | tail -1
