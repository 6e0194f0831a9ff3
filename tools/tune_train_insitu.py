#!/usr/bin/env python3
"""In-situ tuning of the training plan: every candidate row of the tuned table is judged by the time of the WHOLE training step
(two streams: data-gradient chain + weight gradients), not by its launch alone on an idle chip (tools/autotune.py, YM_TUNE_TRAIN=1).

Why: in the step a launch shares the chip with the other stream's kernels.  Workgroup-quantisation tails that cost a lone launch are
filled by the neighbour, while residency (LDS / VGPR footprint) and per-flop efficiency of the tile count for more than they do
alone — the isolated winners are not the in-situ winners (round 6: a prefetch that took 15 % off a kernel's own time slowed the step).

Coordinate descent: shapes in decreasing order of their share of the step; per shape the candidates below; a candidate is kept when
it beats the incumbent by more than `--gain` ms in two independent measurements.  Rows are written to --out (JSON, same format as
yolact_minimal_amd/tuned_gfx950.json); tools/table_gate.py / tests/test_gpu_train_fullsize.py decide whether they enter the table.

    python tools/tune_train_insitu.py --cfg res101_coco --batch 8 --out gpurun_out/insitu_b8.json [--budget 600]
"""
import argparse
import json
import os
import statistics
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yolact_minimal_amd.utils.synthetic import synth_targets  # noqa: E402
from yolact_minimal_amd.config import build_cfg  # noqa: E402
from yolact_minimal_amd.modules.yolact import Yolact  # noqa: E402
from yolact_minimal_amd.trainer import Trainer  # noqa: E402
from yolact_minimal_amd import train_engine as T  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--cfg', default='res101_coco')
ap.add_argument('--batch', type=int, default=8)
ap.add_argument('--steps', type=int, default=8)
ap.add_argument('--gain', type=float, default=0.08, help='ms per step a candidate must win by (twice)')
ap.add_argument('--budget', type=float, default=600.0, help='seconds')
ap.add_argument('--out', default='gpurun_out/insitu.json')
ap.add_argument('--kinds', default='TWF', help='T: data gradients, W: weight gradients, F: forward convs')
ap.add_argument('--only', default='', help='comma-separated substrings: only table rows whose key contains one of them')
ap.add_argument('--splitk', action='store_true', help='candidates: the large tiles WITH a K split (128x128 / 128x64 x ksplit 2-4) only')
args = ap.parse_args()

dev = torch.device('cuda:0')
cfg = build_cfg(args.cfg, 'train', 544, train_bs=args.batch, bs_per_gpu=args.batch)
torch.manual_seed(0)
tr = Trainer(Yolact(cfg), cfg, dev)
img = torch.randn(args.batch, 3, 544, 544, device=dev)
boxes, masks = synth_targets(args.batch, 544, seed=0)
boxes, masks = [b.to(dev) for b in boxes], [m.to(dev) for m in masks]
table = T._table()

# ---- which table rows does one step read, how often, and for how many flops? -----------------------------------------------------
seen = {}
orig_conv, orig_wgrad = T._configure_conv, T._configure_wgrad


def spy_conv(d, key, stats=False):
    k = key + '_st' if (stats and table.get(key + '_st') is not None) else key
    M = d.B * d.Ho * d.Wo
    seen[k] = dict(kind='T' if key.startswith('T_') else 'F', flops=2.0 * M * d.Cout * d.k_pad, stats=stats, M=M, N=d.Cout, nkt=d.k_pad // 32,
                   transposed=int(d.transposed), k=d.KH)
    return orig_conv(d, key, stats)


def spy_wgrad(d, key):
    seen[key] = dict(kind='W', flops=2.0 * d.B * d.Ho * d.Wo * d.Cout * d.KH * d.KW * d.Cin, cout=d.Cout_real)
    return orig_wgrad(d, key)


T._configure_conv, T._configure_wgrad = spy_conv, spy_wgrad
T.tuned_table_changed()
tr.step(img, boxes, masks)
T._configure_conv, T._configure_wgrad = orig_conv, orig_wgrad
T.launch_counts = {}
tr.step(img, boxes, masks)
torch.cuda.synchronize()
T.launch_counts = None


def measure(reps=1):
    out = []
    for _ in range(reps):
        tr.step(img, boxes, masks)
        tr.step(img, boxes, masks)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            tr.step(img, boxes, masks)
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) / args.steps * 1e3)
    return min(out)


def candidates(key, info):
    cur = table.get(key)
    out = []
    if info['kind'] == 'W':
        ms0 = (cur[0] if cur else 0) or 1
        rings = (22, 23, 24) if info['cout'] > 64 else (22,)
        for nb in rings:
            for f in (0.5, 0.75, 1.0, 1.5, 2.0):
                ms = max(1, int(round(ms0 * f)))
                if [ms, nb] != cur and [ms, nb] not in out:
                    out.append([ms, nb])
        return out
    wgs64 = -(-info['M'] // 64) * -(-info['N'] // 64)
    if args.splitk:
        # a 64x64 tile moves 16 KB per 262 kFLOP of its K step through L2 -> LDS (16 FLOP/B: 9.8 TB/s at the MFMA peak), a 128x128 tile
        # half of that: large tiles whose workgroup count is restored by a K split
        for tm, tn in ((128, 128), (128, 64), (64, 128)):
            if tn > 64 and info['N'] <= 64:
                continue
            for ks in (2, 3, 4, 6):
                if ks * 2 > info['nkt']:
                    continue
                for stg in (22, 2):
                    row = [tm, tn, ks, 0, stg, 0, 0]
                    if row != cur:
                        out.append(row)
        return out
    for tm, tn in ((64, 64), (128, 64), (64, 128), (128, 128)):
        if tn > 64 and info['N'] <= 64:
            continue
        for stg in (22, 2) + ((23,) if (tm, tn) == (64, 64) and info['nkt'] >= 3 else ()) + \
                ((43,) if (tm, tn) == (64, 64) and not info['stats'] and info['kind'] == 'F' else ()):
            row = [tm, tn, 1, 0, stg, 0, 0]
            if row != cur and row not in out:
                out.append(row)
    if cur and cur[2] == 1 and wgs64 < 1024 and info['nkt'] >= 8:           # a K split for the small launches
        out.append([cur[0], cur[1], 2, 0, cur[4], 0, 0])
    return out


base = measure(3)
print(f'baseline {base:.3f} ms/step, {len(seen)} table rows in use', flush=True)
order = sorted((k for k in seen if seen[k]['kind'] in args.kinds and (not args.only or any(t in k for t in args.only.split(',')))),
               key=lambda k: -seen[k]['flops'])
t_start = time.time()
accepted = {}
best = base
for key in order:
    if time.time() - t_start > args.budget:
        print('budget exhausted', flush=True)
        break
    info = seen[key]
    cur = table.get(key)
    tried = []
    for cand in candidates(key, info):
        table[key] = cand
        T.tuned_table_changed()
        try:
            t = measure(1)
        except RuntimeError as e:                         # a configuration the kernel refuses
            tried.append((cand, None))
            continue
        tried.append((cand, round(t, 3)))
        if t < best - args.gain:
            t2 = measure(2)                               # confirm (and re-measure the incumbent's neighbourhood drift)
            if t2 < best - args.gain:
                print(f'  {key}: {cur} -> {cand}: {best:.3f} -> {t2:.3f} ms', flush=True)
                best, cur = t2, cand
                accepted[key] = cand
    if cur is None:
        table.pop(key, None)
    else:
        table[key] = cur
    T.tuned_table_changed()
    print(f'{key} [{info["kind"]}] {info["flops"] / 1e9:.1f} GF: kept {cur}; tried {tried}', flush=True)
final = measure(3)
print(f'final {final:.3f} ms/step (baseline {base:.3f}); {len(accepted)} rows changed', flush=True)
os.makedirs(os.path.dirname(args.out) or '.', exist_ok=True)
with open(args.out, 'w') as f:
    json.dump(dict(rows=accepted, baseline_ms=base, final_ms=final, cfg=args.cfg, batch=args.batch), f, indent=0, sort_keys=True)
