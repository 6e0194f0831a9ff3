#!/usr/bin/env python3
"""Training step time (Trainer.step: forward + loss + backward + SGD, one GPU) at another `--img_size` / batch, under whatever plan
source the environment selects (`YM_TUNED_NEAREST=0`: shapes without a row run on the planner heuristic; default: on the row
of the nearest tuned shape, plan_transfer.py; `YM_TUNED_PATH`: a candidate table).  One JSON line.
  python tools/train_size_bench.py --cfg res101_coco --size 416 --batch 8"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--cfg', default='res101_coco')
    ap.add_argument('--size', type=int, default=416)
    ap.add_argument('--batch', type=int, default=8)
    ap.add_argument('--steps', type=int, default=8)
    ap.add_argument('--tag', default='')
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    torch.cuda.set_device(dev)
    r = bench.train_bench(args.cfg, args.size, args.batch, args.steps, 3, 1, 0, dev, lambda: None)
    print(json.dumps(dict(tag=args.tag, cfg=args.cfg, size=args.size, batch=args.batch, ms_per_step=r['ms_per_step'], img_s=r['img_s'],
                          losses=r.get('last_losses'), finite=r.get('finite'))), flush=True)


if __name__ == '__main__':
    main()
