#!/usr/bin/env python3
"""Sweep kernel configurations for every conv / dgrad / wgrad shape of a training step (YM_TUNE_TRAIN=1) and write
the new table entries.  Run on the GPU box: python tools/autotune_train.py --out gpurun_out/tuned_train.json"""
import argparse
import json
import os
import sys

os.environ['YM_TUNE_TRAIN'] = '1'
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from yolact_minimal_amd.utils.synthetic import synth_targets  # noqa: E402
from yolact_minimal_amd import train_engine  # noqa: E402
from yolact_minimal_amd.config import build_cfg  # noqa: E402
from yolact_minimal_amd.modules.yolact import Yolact  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--out', default='gpurun_out/tuned_train.json')
    ap.add_argument('--cfgs', default='res101_coco,res50_coco,swin_tiny_coco')
    ap.add_argument('--batch', type=int, default=8)
    ap.add_argument('--sizes', default='544', help='--img_size values whose layer shapes are swept (rows exist for 544)')
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    for name, size in ((n, int(s)) for n in args.cfgs.split(',') for s in args.sizes.split(',')):
        cfg = build_cfg(name, 'train', size, train_bs=args.batch, bs_per_gpu=args.batch)
        torch.manual_seed(0)
        net = Yolact(cfg).train().to(dev)
        img = torch.randn(args.batch, 3, size, size, device=dev)
        boxes, masks = synth_targets(args.batch, size, seed=0)
        losses = net(img, [b.to(dev) for b in boxes], [m.to(dev) for m in masks])
        sum(losses).backward()
        torch.cuda.synchronize()
        print(name, size, 'entries so far', len(train_engine._new_entries), flush=True)
        del net, losses
        torch.cuda.empty_cache()
    os.makedirs(os.path.dirname(args.out) or '.', exist_ok=True)
    train_engine.dump_new_entries(args.out)
    print('wrote', args.out, len(train_engine._new_entries))


if __name__ == '__main__':
    main()
