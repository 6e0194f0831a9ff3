mkdir -p gpurun_out/r9k
export GPU_MAX_HW_QUEUES=8
python bench.py --leg eval_loop 2>/dev/null | tail -1 > gpurun_out/r9k/eval_loop.json
python - <<'PY'
import json
d = json.load(open('gpurun_out/r9k/eval_loop.json'))
for k, v in d.items():
    if isinstance(v, dict) and 'img_s' in v:
        print(k, v['img_s'], v.get('stage_ms'))
PY
python tools/micro/mask_iou_probe.py
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/prof_miou -o miou -- python $GRAFT_REPO_ROOT/tools/micro/mask_iou_probe.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
find /tmp/prof_miou -name "*kernel_stats*" | head -2
python - <<'PY'
import csv, glob
for f in glob.glob('/tmp/prof_miou/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'mask' in r['Name']:
            print(r['Name'][:60], r['Calls'], r['AverageNs'], r['MinNs'], r['MaxNs'])
PY
