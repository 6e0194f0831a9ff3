#!/usr/bin/env python3
"""Measure the best (tile, ksplit) per conv shape on this MI355X and write yolact_minimal_amd/tuned_gfx950.json
(merged with an existing table).  Run on the GPU box:  python tools/autotune.py --out gpurun_out/tuned_gfx950.json
`--sizes 320,736` tunes the layer shapes of other `--img_size` values (detect.py / eval.py / train.py accept any multiple of 32);
`--skip-known` leaves shapes that already have a row in the shipped table alone."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ['YM_NO_TUNED'] = '1'
from bench import build_net  # noqa: E402
from yolact_minimal_amd import engine as E  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--out', default='gpurun_out/tuned_gfx950.json')
    ap.add_argument('--cfgs', default='res101_coco,res50_coco,swin_tiny_coco')
    ap.add_argument('--batches', default='1,8')
    ap.add_argument('--iters', type=int, default=10)
    ap.add_argument('--sizes', default='544')
    ap.add_argument('--skip-known', action='store_true')
    ap.add_argument('--mma', type=int, default=0, help='0: f32 MFMA; 3 / 6: split-bf16 (entries get the suffix _mma3 / _mma6)')
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    table, detail = {}, {}
    known = set()
    if args.skip_known and os.path.exists(E.TUNED_PATH):
        known = set(json.load(open(E.TUNED_PATH)))
    for cfg_name, size in ((c, int(s)) for c in args.cfgs.split(',') for s in args.sizes.split(',')):
        net, cfg = build_net(cfg_name, size, dev)
        for b in (int(x) for x in args.batches.split(',')):
            img = torch.randn(b, 3, size, size, device=dev)
            eng = net._engine(img)
            print(f'# {cfg_name} {size}px bs={b}', flush=True)
            res = eng.autotune(args.iters, verbose=True, mma=args.mma, skip=known | set(table))
            for k, v in res.items():
                if k not in table:
                    table[k] = v if len(v) > 7 and v[7] else v[:7]      # (eighth field: grid_wgs / waves per workgroup, only when set)
                    detail[k] = list(v[:7]) + list(eng.autotune_detail[k])
            net._engines.clear()
            torch.cuda.empty_cache()
    os.makedirs(os.path.dirname(args.out) or '.', exist_ok=True)
    json.dump(table, open(args.out, 'w'), indent=0, sort_keys=True)
    json.dump(detail, open(args.out.replace('.json', '_detail.json'), 'w'), indent=0, sort_keys=True)
    tot_b = sum(v[8] for v in detail.values())
    tot_a = sum(v[7] for v in detail.values())
    print(f'{len(table)} shapes; sum of per-shape best {tot_a:.0f} us vs default {tot_b:.0f} us')


if __name__ == '__main__':
    main()
