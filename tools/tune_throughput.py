#!/usr/bin/env python3
"""Experiment: coordinate descent on the END-TO-END forward throughput with requests in flight.  The table holds, per conv shape,
the choice with the lowest latency for a launch that has the chip to itself; with N requests in flight other choices (no K split,
no tail split, a larger tile) might serve the pipeline better.  For the shapes with the largest FLOP share, try a few alternatives
in place (all engines of the pipeline), keep one only if the pipeline's img/s improves twice in a row.
    tune_throughput.py [cfg] [inflight] [batch] [top]      -> gpurun_out/tuned_if<inflight>_bs<batch>.json"""
import json
import os
import sys

os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
import torch  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

dev = torch.device('cuda:0')
name = sys.argv[1] if len(sys.argv) > 1 else 'res101_coco'
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4
BATCH = int(sys.argv[3]) if len(sys.argv) > 3 else 1
TOP = int(sys.argv[4]) if len(sys.argv) > 4 else 28
STEPS = 240 if BATCH == 1 else 40
net, cfg = bench.build_net(name, 544, dev)
w = bench.Workload(net, cfg, BATCH, 544, dev, with_post=False, inflight=N)
engines = w.pipe.engines


def current(sig):
    for c in engines[0].convs:
        if c.sig == sig:
            return [c.tile[0], c.tile[1], c.ksplit, c.kwaves, c.stages, c.tail[0], c.tail[1], int(c.desc.grid_wgs)]


def apply(sig, v):
    for e in engines:
        for c in e.convs:
            if c.sig == sig:
                c.tile, c.ksplit, c.kwaves, c.stages, c.tail = (v[0], v[1]), v[2], v[3], v[4], (v[5], v[6])
                c.desc.grid_wgs = v[7]
        e.retune()
    w.pipe.warm_up(w.img)


def measure():
    t = min(bench.timed(w, STEPS, 4, lambda: None), bench.timed(w, STEPS, 2, lambda: None)) / STEPS
    return BATCH / t


share = {}
for c in engines[0].convs:
    if not c.stem and c.desc.nlevels == 0:
        share[c.sig] = share.get(c.sig, 0.0) + c.flops
order = sorted(share, key=share.get, reverse=True)[:TOP]
best = measure()
print(f'start: {best:.1f} img/s with {N} in flight, batch {BATCH}', flush=True)
kept = {}
for sig in order:
    cur = current(sig)
    alts = []
    if cur[2] > 1 or cur[5] > 0:
        alts.append(cur[:2] + [1, 0, cur[4], 0, 0, 0])                        # the same kernel without any K split
    if cur[2] > 2:
        alts.append(cur[:2] + [cur[2] // 2, 0, cur[4], 0, 0, 0])
    alts += [[64, 64, 1, 0, 43, 0, 0, 0], [128, 64, 1, 0, 22, 0, 0, 0], [64, 128, 1, 0, 22, 0, 0, 0], [128, 128, 1, 0, 22, 0, 0, 0]]
    seen = [cur]
    for a in alts:
        if a in seen:
            continue
        seen.append(a)
        try:
            apply(sig, a)
            r = measure()
            if r > best * 1.004:
                r = min(r, measure())
        except RuntimeError as e:
            print('   ', sig, a, 'failed:', str(e)[:80], flush=True)
            r = 0.0
        tag = ''
        if r > best * 1.003:
            best, cur, kept[sig], tag = r, a, a, '  <-- kept'
        print(f'{sig:40s} {share[sig] / sum(share.values()):5.1%} {a} {r:7.1f}{tag}', flush=True)
    apply(sig, cur)
final = measure()
print(f'final: {final:.1f} img/s; kept {len(kept)}', flush=True)
os.makedirs('gpurun_out', exist_ok=True)
json.dump({'start_end': [best, final], 'kept': kept}, open(f'gpurun_out/tuned_if{N}_bs{BATCH}.json', 'w'), indent=0, sort_keys=True)
