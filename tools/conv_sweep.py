#!/usr/bin/env python3
"""Time every (kernel, tile, split) candidate for a few conv shapes; prints a table per shape."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yolact_minimal_amd import hip  # noqa: E402


def make_desc(b, h, w, cin, cout, k, stride, residual, dev):
    pad = k // 2
    ho, wo = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
    x = torch.randn(b, h, w, cin, device=dev)
    wt = torch.randn(cout, k * k * cin, device=dev) * 0.02
    sc, sh = torch.rand(cout, device=dev) + 0.5, torch.randn(cout, device=dev)
    out = torch.empty(b, ho, wo, cout, device=dev)
    res = torch.randn(b, ho, wo, cout, device=dev) if residual else None
    d = hip.ConvDesc()
    d.inp, d.weight, d.scale, d.shift = x.data_ptr(), wt.data_ptr(), sc.data_ptr(), sh.data_ptr()
    d.residual = res.data_ptr() if residual else None
    d.B, d.H, d.W, d.Cin, d.Cout, d.KH, d.KW = b, h, w, cin, cout, k, k
    d.stride, d.pad, d.Ho, d.Wo, d.k_pad, d.nseg = stride, pad, ho, wo, k * k * cin, 1
    d.seg[0].n_begin, d.seg[0].n_end, d.seg[0].out = 0, cout, out.data_ptr()
    d.seg[0].batch_stride, d.seg[0].pitch, d.seg[0].act = ho * wo * cout, cout, 1
    return d, (x, wt, sc, sh, out, res)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--shapes', default='1,34,34,256,256,3,1,0;1,34,34,1024,256,1,1,0;1,34,34,256,1024,1,1,1;1,5,5,256,256,3,1,0')
    ap.add_argument('--iters', type=int, default=20)
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    ws = torch.empty(1 << 28, dtype=torch.uint8, device=dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for spec in args.shapes.split(';'):
        b, h, w, cin, cout, k, stride, res = (int(v) for v in spec.split(','))
        d, keep = make_desc(b, h, w, cin, cout, k, stride, res, dev)
        M, nkt = b * d.Ho * d.Wo, d.k_pad // 32
        flops = 2.0 * M * cout * d.k_pad
        rows = []
        cands = [((0, 0), 0, 0, 0)]
        for t in ((128, 128), (128, 64), (64, 128), (64, 64)):
            for ks in (1, 2, 3, 4, 6, 8, 12, 16, 24):
                if ks <= nkt:
                    cands.append((t, ks, 0, 2))
                    if t != (128, 128):
                        cands.append((t, ks, 0, 3))
        for t in ((32, 32), (64, 32), (32, 64), (64, 64)):
            for kw in (1, 2, 4, 8):
                if not (kw == 8 and t == (64, 64)):
                    cands.append((t, 1, kw, 0))
        for tile, ks, kw, stg in cands:
            d.tile_m, d.tile_n, d.ksplit, d.kwaves, d.stages = tile[0], tile[1], ks, kw, stg
            if hip.conv_workspace_bytes(d) > ws.numel():
                continue
            try:
                for _ in range(3):
                    hip.conv2d_fwd(d, ws)
            except RuntimeError:
                continue
            best = 1e9
            for _ in range(3):
                e0.record()
                for _ in range(args.iters):
                    hip.conv2d_fwd(d, ws)
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / args.iters * 1e3)
            rows.append((best, tile, ks, kw, stg))
        rows.sort()
        print(f'== M={M} N={cout} K={d.k_pad} k={k} s={stride} res={res}: {flops / 1e9:.2f} GFLOP')
        for t, tile, ks, kw, stg in rows[:12]:
            print(f'   {"wave" if kw else "wg  "} tile={tile} ksplit={ks} kwaves={kw} stages={stg}: {t:7.1f} us  {flops / t / 1e6:6.1f} TF')
        worst = [r for r in rows if r[3] > 0][:6]
        print('   best wave-kernel variants:', [(f'{r[0]:.1f}', r[1], r[3]) for r in worst])


if __name__ == '__main__':
    main()
