#!/usr/bin/env python3
"""A/B of train_engine switches inside ONE process (same box, same clocks, same allocator state): every named setting is applied in
turn, `--reps` times round robin, `--steps` timed steps each after 2 untimed ones.

    python tools/train_ab.py --set base:_GRAD_JOIN=0,_REDUCE_BATCH=0 --set join:_GRAD_JOIN=1,_REDUCE_BATCH=0 --set both:_GRAD_JOIN=1,_REDUCE_BATCH=24
"""
import argparse
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
ap.add_argument('--reps', type=int, default=3)
ap.add_argument('--set', action='append', default=[], help='name:ATTR=value[,ATTR=value...] (attributes of train_engine; ints / 0 / 1)')
args = ap.parse_args()
dev = torch.device('cuda:0')
cfg = build_cfg(args.cfg, 'train', 544, train_bs=args.batch, bs_per_gpu=args.batch)
torch.manual_seed(0)
tr = Trainer(Yolact(cfg), cfg, dev)
img = torch.randn(args.batch, 3, 544, 544, device=dev)
boxes, masks = synth_targets(args.batch, 544, seed=0)
boxes, masks = [b.to(dev) for b in boxes], [m.to(dev) for m in masks]
settings = []
for s in args.set:
    name, rest = s.split(':', 1)
    kv = {}
    for item in rest.split(','):
        k, v = item.split('=')
        old = getattr(T, k)
        kv[k] = bool(int(v)) if isinstance(old, bool) else int(v)
    settings.append((name, kv))
res = {name: [] for name, _ in settings}
for rep in range(args.reps):
    for name, kv in settings:
        for k, v in kv.items():
            setattr(T, k, v)
        T._desc_cache.clear()
        for _ in range(2):
            tr.step(img, boxes, masks)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            tr.step(img, boxes, masks)
        torch.cuda.synchronize()
        res[name].append((time.perf_counter() - t0) / args.steps * 1e3)
        print(f'rep {rep} {name:12s} {res[name][-1]:.2f} ms/step', flush=True)
for name, v in res.items():
    print(f'{name:12s} median {statistics.median(v):.2f}  min {min(v):.2f}  max {max(v):.2f} ms/step  {v}')
print('reduce launches / layers:', T.wgrad_reduce_launches, ' join passes:', T.grad_join_passes[0], ' fused BN-backward launches:', T.bn_bwd_fused_launches[0])
