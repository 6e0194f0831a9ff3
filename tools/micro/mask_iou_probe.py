#!/usr/bin/env python3
"""ym_mask_iou on the bench's case (100 predicted x 15 gt masks at 480x640) through the C ABI, scratch allocated once: device time per
call by HIP events; under `rocprofv3 --kernel-trace --stats` the two kernels' own durations.
  python tools/micro/mask_iou_probe.py [--n 100 --g 15 --iters 50]"""
import argparse
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from yolact_minimal_amd import hip as H  # noqa: E402
from yolact_minimal_amd.utils.synthetic import synth_eval_case  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--n', type=int, default=100)
ap.add_argument('--g', type=int, default=15)
ap.add_argument('--iters', type=int, default=50)
args = ap.parse_args()
dev = torch.device('cuda:0')
_, _, _, masks, _, gt_masks, h, w = synth_eval_case(1, args.n, args.g, 480, 640, 10)
a, b = masks.to(dev).reshape(args.n, -1).contiguous(), gt_masks.to(dev).reshape(args.g, -1).contiguous()
P = a.shape[1]
nb = H.lib().ym_mask_iou_workspace_bytes(args.n, args.g, P)
ws, iou = torch.empty(nb, device=dev, dtype=torch.uint8), torch.empty(args.n, args.g, device=dev)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
best = 1e30
for _ in range(4):
    e0.record()
    for _ in range(args.iters):
        H.check(H.lib().ym_mask_iou(H.ptr(a), args.n, H.ptr(b), args.g, P, H.ptr(iou), ctypes.c_void_p(ws.data_ptr()), nb, H.stream_ptr()), 'mask_iou')
    e1.record()
    torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) / args.iters * 1e3)
nbytes = (args.n + args.g) * P * 4
ref = (a @ b.t())
area = a.sum(1, keepdim=True) + b.sum(1)[None] - ref
print(f'mask_iou {args.n} x {args.g} x {P}: {best:.1f} us per call, {nbytes / best / 1e3:.0f} GB/s = {nbytes / best / 1e3 / 8000:.3f} of the HBM peak; '
      f'equal to the fp32 matmul formula: {bool(torch.equal(iou, ref / area))}')
