#!/usr/bin/env python3
"""Allocator footprint of the training step over many steps (the side-stream weight gradients use record_stream: the footprint
must settle instead of growing)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yolact_minimal_amd.utils.synthetic import synth_targets  # noqa: E402
from yolact_minimal_amd.config import build_cfg  # noqa: E402
from yolact_minimal_amd.modules.yolact import Yolact  # noqa: E402
from yolact_minimal_amd.trainer import Trainer  # noqa: E402

dev = torch.device('cuda:0')
name = sys.argv[1] if len(sys.argv) > 1 else 'res101_coco'
cfg = build_cfg(name, 'train', 544, train_bs=8, bs_per_gpu=8)
torch.manual_seed(0)
tr = Trainer(Yolact(cfg), cfg, dev)
img = torch.randn(8, 3, 544, 544, device=dev)
boxes, masks = synth_targets(8, 544, seed=0)
boxes, masks = [b.to(dev) for b in boxes], [m.to(dev) for m in masks]
for i in range(60):
    tr.step(img, boxes, masks)
    if i % 10 == 9:
        torch.cuda.synchronize()
        print(f'step {i + 1}: allocated {torch.cuda.memory_allocated() / 2 ** 30:.2f} GiB, peak {torch.cuda.max_memory_allocated() / 2 ** 30:.2f} GiB, '
              f'reserved {torch.cuda.memory_reserved() / 2 ** 30:.2f} GiB')
