#!/usr/bin/env python3
"""Forward time (graph replay, one request at a time) of a config at several `--img_size` values, with whatever plan source the
environment selects: the shipped table (default), `YM_TUNED_PATH=...` (a candidate table), `YM_NO_TUNED=1` (planner heuristics
only), `YM_TUNED_NEAREST=0` (no nearest-row transfer for shapes without a row).  One JSON line per size.
  python tools/size_bench.py --cfg res101_coco --sizes 320,544,736"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_net, Workload, timed, F32_MFMA_PEAK_TFLOPS  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--cfg', default='res101_coco')
    ap.add_argument('--sizes', default='320,544,736')
    ap.add_argument('--batch', type=int, default=1)
    ap.add_argument('--steps', type=int, default=60)
    ap.add_argument('--tag', default='')
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    for size in (int(s) for s in args.sizes.split(',')):
        net, cfg = build_net(args.cfg, size, dev)
        w = Workload(net, cfg, args.batch, size, dev, with_post=False)
        t = min(timed(w, args.steps, 5, lambda: None), timed(w, args.steps, 0, lambda: None), timed(w, args.steps, 0, lambda: None)) / args.steps
        fl = w.engine.total_flops
        src = {}
        for c in w.engine.convs:
            src[getattr(c, 'plan_source', '?')] = src.get(getattr(c, 'plan_source', '?'), 0) + 1
        print(json.dumps(dict(tag=args.tag, cfg=args.cfg, size=size, batch=args.batch, forward_ms=round(t * 1e3, 4),
                              img_s=round(args.batch / t, 1), gflop=round(fl / 1e9, 1),
                              frac_f32_mfma_peak=round(fl / t / 1e12 / F32_MFMA_PEAK_TFLOPS, 4), plan_sources=src)), flush=True)
        net._engines.clear()
        del w, net
        torch.cuda.empty_cache()


if __name__ == '__main__':
    main()
