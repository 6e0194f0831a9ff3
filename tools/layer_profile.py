#!/usr/bin/env python3
"""Per-layer conv table (shape, tile, ms, TFLOP/s) measured with HIP events, eager launches."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_net, conv_roofline  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--cfg', default='res101_coco')
    ap.add_argument('--batch', type=int, default=1)
    ap.add_argument('--iters', type=int, default=10)
    ap.add_argument('--json', default=None)
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    net, cfg = build_net(args.cfg, 544, dev)
    img = torch.randn(args.batch, 3, 544, 544, device=dev)
    eng = net._engine(img)
    flops, secs, n, layers = conv_roofline(eng, img, args.iters)
    convs = [a for k, a in eng.ops if k == 'conv']
    rows = []
    for c, l in zip(convs, layers):
        d = c.desc
        M = d.B * d.Ho * d.Wo
        rows.append(dict(name=c.name, M=M, N=d.Cout, K=d.k_pad, k=d.KH, s=d.stride, ms=l['ms'], tf=l['gflop'] / l['ms']))
    # aggregate identical shapes
    agg = {}
    for r in rows:
        key = (r['M'], r['N'], r['K'], r['k'], r['s'])
        a = agg.setdefault(key, dict(count=0, ms=0.0, gflop=0.0))
        a['count'] += 1
        a['ms'] += r['ms']
        a['gflop'] += r['tf'] * r['ms']
    print(f'{args.cfg} bs={args.batch}: {n} convs, {flops/1e9:.1f} GFLOP, conv time {secs*1e3:.3f} ms -> {flops/secs/1e12:.1f} TF')
    print(f'{"M":>7} {"N":>5} {"K":>5} k s {"cnt":>3} {"ms_tot":>8} {"%":>5} {"TF":>6}')
    for key, a in sorted(agg.items(), key=lambda kv: -kv[1]['ms']):
        print(f'{key[0]:7d} {key[1]:5d} {key[2]:5d} {key[3]} {key[4]} {a["count"]:3d} {a["ms"]:8.4f} {100*a["ms"]/(secs*1e3):5.1f} {a["gflop"]/a["ms"]:6.1f}')
    if args.json:
        json.dump(rows, open(args.json, 'w'), indent=1)


if __name__ == '__main__':
    main()
