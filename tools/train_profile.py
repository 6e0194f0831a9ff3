#!/usr/bin/env python3
"""Run K training steps only (for `rocprofv3 --kernel-trace --stats -- python tools/train_profile.py`)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yolact_minimal_amd.utils.synthetic import synth_targets  # noqa: E402
from yolact_minimal_amd.config import build_cfg  # noqa: E402
from yolact_minimal_amd.modules.yolact import Yolact  # noqa: E402
from yolact_minimal_amd.trainer import Trainer  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--cfg', default='res101_coco')
ap.add_argument('--batch', type=int, default=8)
ap.add_argument('--steps', type=int, default=8)
ap.add_argument('--ab', default='', help="A/B inside one process: name of a boolean switch of train_engine (e.g. _FUSE_BN_BWD)")
ap.add_argument('--prio', action='store_true', help='run the step on a high-priority stream (the weight-gradient side stream stays normal)')
args = ap.parse_args()
dev = torch.device('cuda:0')
if args.prio:
    torch.cuda.set_stream(torch.cuda.Stream(device=dev, priority=-1))
cfg = build_cfg(args.cfg, 'train', 544, train_bs=args.batch, bs_per_gpu=args.batch)
torch.manual_seed(0)
tr = Trainer(Yolact(cfg), cfg, dev)
img = torch.randn(args.batch, 3, 544, 544, device=dev)
boxes, masks = synth_targets(args.batch, 544, seed=0)
boxes, masks = [b.to(dev) for b in boxes], [m.to(dev) for m in masks]
for _ in range(2):
    tr.step(img, boxes, masks)
torch.cuda.synchronize()
import time  # noqa: E402
t0 = time.perf_counter()
for _ in range(args.steps):
    tr.step(img, boxes, masks)
torch.cuda.synchronize()
print(f'{(time.perf_counter() - t0) / args.steps * 1e3:.2f} ms/step over {args.steps} steps (+2 warm-up)')
if args.ab:
    from yolact_minimal_amd import train_engine as T  # noqa: E402
    for rep in range(3):
        for val in (False, True):
            setattr(T, args.ab, val)
            T._desc_cache.clear()
            for _ in range(2):
                tr.step(img, boxes, masks)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                tr.step(img, boxes, masks)
            torch.cuda.synchronize()
            print(f'  {args.ab}={val}: {(time.perf_counter() - t0) / args.steps * 1e3:.2f} ms/step')
    print('fused BN-backward launches so far:', T.bn_bwd_fused_launches[0])
