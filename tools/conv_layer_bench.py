"""Time ONE conv shape under a list of kernel configurations (tile, ksplit, stages, mma), optionally for rocprofv3 --pmc.
    python tools/conv_layer_bench.py --shape 8,136,136,256,256,3,1 --cfgs 128x128:1:0:3,128x128:1:3:3,128x128:1:22:0 [--iters 20]
shape = B,H,W,Cin,Cout,K,stride (pad = K//2);  cfg = tile_m x tile_n : ksplit : stages : mma"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from yolact_minimal_amd import hip
    ap = argparse.ArgumentParser()
    ap.add_argument('--shape', default='8,136,136,256,256,3,1')
    ap.add_argument('--cfgs', default='128x128:1:0:0,128x128:1:0:3,128x128:1:3:3')
    ap.add_argument('--iters', type=int, default=20)
    ap.add_argument('--relu', type=int, default=1)
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    b, h, w, cin, cout, k, stride = (int(v) for v in args.shape.split(','))
    pad = k // 2
    g = torch.Generator().manual_seed(0)
    x = torch.randn(b, h, w, cin, generator=g).to(dev)
    wt = (torch.randn(cout, cin, k, k, generator=g) * 0.03).to(dev)
    k_pad = k * k * cin
    wp = hip.pack_conv_weight(wt, cin, k_pad)
    bias = torch.randn(cout, generator=g).to(dev)
    ho, wo = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
    out = torch.empty(b, ho, wo, cout, device=dev)
    counters = torch.zeros(hip.TILE_COUNTERS, device=dev, dtype=torch.int32)
    ws = torch.empty(1 << 28, dtype=torch.uint8, device=dev)
    flops = 2.0 * b * ho * wo * cout * k * k * cin
    ref = None
    for cfg in args.cfgs.split(','):
        tile, ks, stg, mma = cfg.split(':')
        tm, tn = (int(v) for v in tile.split('x'))
        d = hip.ConvDesc()
        d.inp, d.weight, d.shift = x.data_ptr(), wp.data_ptr(), bias.data_ptr()
        d.B, d.H, d.W, d.Cin, d.Cout, d.KH, d.KW = b, h, w, cin, cout, k, k
        d.stride, d.pad, d.Ho, d.Wo, d.k_pad, d.nseg = stride, pad, ho, wo, k_pad, 1
        d.seg[0].n_begin, d.seg[0].n_end, d.seg[0].out = 0, cout, out.data_ptr()
        d.seg[0].batch_stride, d.seg[0].pitch, d.seg[0].act = ho * wo * cout, cout, args.relu
        d.tile_counters = counters.data_ptr()
        d.tile_m, d.tile_n, d.ksplit, d.stages, d.mma = tm, tn, int(ks), int(stg), int(mma)
        for _ in range(3):
            hip.conv2d_fwd(d, ws)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.iters):
            hip.conv2d_fwd(d, ws)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / args.iters * 1e3
        err = None
        if ref is None:
            ref = out.clone()
        else:
            err = float((out - ref).abs().max() / ref.abs().max())
        print(json.dumps(dict(shape=args.shape, cfg=cfg, us=round(us, 1), tflops_f32_equiv=round(flops / us / 1e6, 1), rel_err_vs_first=err)), flush=True)


if __name__ == '__main__':
    main()
