#!/usr/bin/env python3
"""Experiment / tuner: per-layer kernel choices for serving with requests in flight (bench.py --inflight N).
Measures the N-in-flight forward throughput with the table as it is, re-tunes every conv shape of the plan under
`InferEngine.autotune(concurrent=N)` (N copies of a launch side by side), measures again, and writes the choices as
`<signature>_cN` entries to gpurun_out/tuned_cN.json.      tune_inflight.py [cfg] [copies] [batch]"""
import json
import os
import sys

os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
import torch  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from yolact_minimal_amd.engine import tuned_table  # noqa: E402

dev = torch.device('cuda:0')
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
name = sys.argv[1] if len(sys.argv) > 1 else 'res101_coco'
COPIES = int(sys.argv[2]) if len(sys.argv) > 2 else 2
BATCH = int(sys.argv[3]) if len(sys.argv) > 3 else 1
STEPS = 200 if BATCH == 1 else 40
net, cfg = bench.build_net(name, 544, dev)


def throughput(inflight):
    net._engines.clear()
    w = bench.Workload(net, cfg, BATCH, 544, dev, with_post=False, inflight=inflight)
    t = min(bench.timed(w, STEPS, STEPS // 10, lambda: None), bench.timed(w, STEPS, 5, lambda: None)) / STEPS
    return BATCH / t


before = {s: throughput(s) for s in (1, COPIES)}
print('as tuned (per-launch latency):', {k: round(v, 1) for k, v in before.items()}, flush=True)
net._engines.clear()
eng = net._engine(torch.randn(BATCH, 3, 544, 544, device=dev))
res = eng.autotune(10, verbose=True, concurrent=COPIES)
saved = {k: tuned_table().get(k) for k in res}
tuned_table().update({k: v[:7] for k, v in res.items()})
after = {s: throughput(s) for s in (1, COPIES)}
print('tuned with two copies side by side:', {k: round(v, 1) for k, v in after.items()}, flush=True)
os.makedirs('gpurun_out', exist_ok=True)
json.dump({k + f'_c{COPIES}': v[:7] for k, v in res.items() if v[:7] != (saved.get(k) or [])[:7]}, open(f'gpurun_out/tuned_c{COPIES}_bs{BATCH}.json', 'w'), indent=0, sort_keys=True)
json.dump(dict(before=before, after=after, detail=res), open(f'gpurun_out/tuned_c{COPIES}_bs{BATCH}_detail.json', 'w'), indent=0, sort_keys=True)
