#!/usr/bin/env python3
"""Where do the device-to-device copies / clones of a training step come from?  torch.profiler with Python stacks, grouped."""
import os
import sys
from collections import Counter

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.yolact_ref import synth_targets  # noqa: E402  (input generator)
from yolact_minimal_amd.config import build_cfg  # noqa: E402
from yolact_minimal_amd.modules.yolact import Yolact  # noqa: E402
from yolact_minimal_amd.trainer import Trainer  # noqa: E402

dev = torch.device('cuda:0')
cfg = build_cfg(sys.argv[1] if len(sys.argv) > 1 else 'res101_coco', 'train', 544, train_bs=8, bs_per_gpu=8)
torch.manual_seed(0)
tr = Trainer(Yolact(cfg), cfg, dev)
img = torch.randn(8, 3, 544, 544, device=dev)
boxes, masks = synth_targets(8, 544, seed=0)
boxes, masks = [b.to(dev) for b in boxes], [m.to(dev) for m in masks]
for _ in range(2):
    tr.step(img, boxes, masks)
torch.cuda.synchronize()
from torch.profiler import ProfilerActivity, profile  # noqa: E402
with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
    tr.step(img, boxes, masks)
    torch.cuda.synchronize()
names = Counter()
stacks = Counter()
for e in prof.events():
    names[e.name] += 1
    if e.name in ('aten::copy_', 'aten::clone', 'aten::contiguous', 'aten::add', 'aten::add_', 'aten::zeros_like', 'aten::mul'):
        st = [s for s in (e.stack or []) if 'yolact_minimal_amd' in s or 'autograd' in s][:3]
        stacks[(e.name, ' <- '.join(s.split('/')[-1] for s in st))] += 1
print('--- op counts (top 30)')
for n, c in names.most_common(30):
    print(f'{c:6d} {n}')
print('--- copy-like ops by stack')
for (n, st), c in stacks.most_common(40):
    print(f'{c:5d} {n:18s} {st}')
