#!/usr/bin/env python3
"""Who leads, the host or the device?  Per phase of a training step (forward + loss, backward, optimizer) the host time spent
enqueueing it and the device time spent executing it (events recorded at the same program points)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yolact_minimal_amd.utils.synthetic import synth_targets  # noqa: E402
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
for _ in range(3):
    tr.step(img, boxes, masks)
torch.cuda.synchronize()
N = 6
host, evs = [], []
for _ in range(N):
    e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    t = [time.perf_counter()]
    e[0].record()
    tr.opt.lr = 1e-3
    tr.opt.zero_grad()
    losses = tr.model(img, boxes, masks)
    t.append(time.perf_counter()); e[1].record()
    total = losses[0] + losses[1] + losses[2] + losses[3]
    total.backward()
    t.append(time.perf_counter()); e[2].record()
    tr.opt.step()
    tr.net.mark_weights_changed()
    t.append(time.perf_counter()); e[3].record()
    host.append(t)
    evs.append(e)
torch.cuda.synchronize()
names = ('forward+loss', 'backward', 'optimizer')
for k, nm in enumerate(names):
    h = sum(t[k + 1] - t[k] for t in host[1:]) / (N - 1) * 1e3
    d = sum(e[k].elapsed_time(e[k + 1]) for e in evs[1:]) / (N - 1)
    print(f'{nm:14s} host {h:7.2f} ms   device {d:7.2f} ms')
wall = (host[-1][-1] - host[1][0]) / (N - 1) * 1e3
print(f'host loop {wall:.2f} ms/step (device-bound if this matches the device sum)')
