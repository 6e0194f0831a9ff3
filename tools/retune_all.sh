#!/bin/bash
# Full re-tune on the GPU box: inference shapes (bs 1, 8; three backbones), then the training-only shapes (data gradients, the
# forward launches with fused statistics) at bs 8 (three backbones) and bs 16 (res101).  Weight-gradient (W_) and split-bf16
# (_mma3 / _mma6) entries are kept as they are (tools/retune_wgrad.py, tools/autotune.py --mma 3).
# Usage (from the repo root): bash tools/retune_all.sh  -> gpurun_out/tuned_final.json
set -e
mkdir -p gpurun_out
python tools/autotune.py --out gpurun_out/tuned_A.json > gpurun_out/tune_A.log 2>&1
python - <<'PY'
import json
old = json.load(open('yolact_minimal_amd/tuned_gfx950.json'))
a = json.load(open('gpurun_out/tuned_A.json'))
b = {k: v for k, v in old.items() if k.startswith('W_') or '_mma' in k}
b.update(a)
json.dump(b, open('gpurun_out/tuned_B.json', 'w'), indent=0, sort_keys=True)
PY
YM_TUNED_PATH=gpurun_out/tuned_B.json python tools/autotune_train.py --out gpurun_out/tuned_T.json > gpurun_out/tune_T.log 2>&1
python - <<'PY'
import json
b = json.load(open('gpurun_out/tuned_B.json'))
b.update(json.load(open('gpurun_out/tuned_T.json')))
json.dump(b, open('gpurun_out/tuned_B2.json', 'w'), indent=0, sort_keys=True)
PY
YM_TUNED_PATH=gpurun_out/tuned_B2.json python tools/autotune_train.py --cfgs res101_coco --batch 16 --out gpurun_out/tuned_T16.json > gpurun_out/tune_T16.log 2>&1
python - <<'PY'
import json
b = json.load(open('gpurun_out/tuned_B2.json'))
b.update(json.load(open('gpurun_out/tuned_T16.json')))
old = json.load(open('yolact_minimal_amd/tuned_gfx950.json'))
lost = [k for k in old if k not in b]
json.dump(b, open('gpurun_out/tuned_final.json', 'w'), indent=0, sort_keys=True)
print(len(b), 'entries;', len(lost), 'keys of the old table were not re-tuned (dropped):', lost[:8])
PY
