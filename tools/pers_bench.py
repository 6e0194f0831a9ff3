#!/usr/bin/env python3
"""Persistent conv kernel (stages 4x, csrc/conv_persist.hip) against the tuned per-item kernels, per layer shape.

    python tools/pers_bench.py bs1|bs8|train|swin|swin_train [--write]     (train = the data-gradient launches of the bs=8 shapes)

For every 64x64-tile shape of the plan: time the tuned entry, then the persistent kernel over ring depth x workgroups per CU
(with the tail split that evens out the last round of a static item assignment) and, at bs=1, over the K split.  `--write`
stores the winners in tuned_gfx950.json as [64, 64, ksplit, 0, 4x, tail_tiles, tail_ksplit, grid_wgs].
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from yolact_minimal_amd import hip  # noqa: E402
from tools.conv_sweep import make_desc  # noqa: E402

dev = torch.device('cuda:0')
TUNED = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'yolact_minimal_amd', 'tuned_gfx950.json')
tuned = json.load(open(TUNED))
ws = torch.empty(1 << 28, dtype=torch.uint8, device=dev)
counters = torch.zeros(hip.TILE_COUNTERS, dtype=torch.int32, device=dev)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
CUS = 256


def bottleneck(bs, hw, wide, mid, first_stride=None):
    """(b, h, w, cin, cout, k, stride, residual) of one ResNet stage's repeated block"""
    return [(bs, hw, hw, wide, mid, 1, 1, 0), (bs, hw, hw, mid, mid, 3, 1, 0), (bs, hw, hw, mid, wide, 1, 1, 1)]


def shapes(which):
    bs = 1 if which == 'bs1' else 8
    out = []
    if which in ('swin', 'swin_train'):
        # Swin-T bs=8 linears the persistent kernel covers (ReLU / identity epilogues: qkv, proj, fc2, merging); swin_train: the
        # data gradients of ALL its linears (no BatchNorm, so none of them carries fused statistics)
        for hw, c in ((136, 96), (68, 192), (34, 384), (17, 768)):
            out += [(8, hw, hw, c, 3 * c, 1, 1, 0), (8, hw, hw, c, c, 1, 1, 1), (8, hw, hw, 4 * c, c, 1, 1, 1)]
            if which == 'swin_train':
                out.append((8, hw, hw, c, 4 * c, 1, 1, 0))
            if hw > 17:
                out.append((8, hw // 2, hw // 2, 4 * c, 2 * c, 1, 1, 0))
        return out
    for hw, wide, mid in ((136, 256, 64), (68, 512, 128), (34, 1024, 256), (17, 2048, 512)):
        out += bottleneck(bs, hw, wide, mid)
    # FPN / protonet / head 3x3 256 -> 256 at 68 / 34 / 17 and the lateral 1x1s
    out += [(bs, 68, 68, 256, 256, 3, 1, 0), (bs, 34, 34, 256, 256, 3, 1, 0), (bs, 17, 17, 256, 256, 3, 1, 0),
            (bs, 17, 17, 2048, 256, 1, 1, 0), (bs, 34, 34, 1024, 256, 1, 1, 1), (bs, 68, 68, 512, 256, 1, 1, 1)]
    return out


def make_dgrad_desc(b, h, w, cin, cout, k, stride):
    """data gradient of conv(x [b,h,w,cin] -> y [b,ho,wo,cout]): dy in, dx out (ym_conv_desc.transposed)"""
    pad = k // 2
    ho, wo = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
    cout_pad = (cout + 31) // 32 * 32
    dy = torch.randn(b, ho, wo, cout_pad, device=dev)
    wt = torch.randn(cin, k * k * cout_pad, device=dev) * 0.02
    dx = torch.empty(b, h, w, cin, device=dev)
    d = hip.ConvDesc()
    d.inp, d.weight = dy.data_ptr(), wt.data_ptr()
    d.B, d.H, d.W, d.Cin, d.Cout, d.KH, d.KW = b, ho, wo, cout_pad, cin, k, k
    d.stride, d.pad, d.Ho, d.Wo, d.k_pad, d.nseg = stride, pad, h, w, k * k * cout_pad, 1
    d.seg[0].n_begin, d.seg[0].n_end, d.seg[0].out = 0, cin, dx.data_ptr()
    d.seg[0].batch_stride, d.seg[0].pitch, d.seg[0].act = h * w * cin, cin, 0
    d.transposed = 1
    return d, (dy, wt, dx), f'T_M{b * h * w}_N{cin}_C{cout_pad}_k{k}_s{stride}'


def time_desc(d, iters=30, reps=3):
    try:
        for _ in range(3):
            hip.conv2d_fwd(d, ws)
    except RuntimeError as ex:
        return None
    best = 1e9
    for _ in range(reps):
        e0.record()
        for _ in range(iters):
            hip.conv2d_fwd(d, ws)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters * 1e3)
    return best


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else 'bs1'
    write = '--write' in sys.argv
    only = [a for a in sys.argv[2:] if not a.startswith('--')]
    total_old = total_new = 0.0
    for spec in shapes(which):
        if which in ('train', 'swin_train'):
            d, keep, sig = make_dgrad_desc(*spec[:7])
            M = spec[0] * d.Ho * d.Wo
            spec = (spec[0], 0, 0, d.Cin, d.Cout) + tuple(spec[5:])
        else:
            d, keep = make_desc(*spec, dev)
            M = spec[0] * d.Ho * d.Wo
            sig = f'M{M}_N{spec[4]}_C{spec[3]}_k{spec[5]}_s{spec[6]}_seg1_r{spec[7]}'
        if only and sig not in only:
            continue
        nkt = d.k_pad // 32
        hit = tuned.get(sig) or [0, 0, 0, 0, 0, 0, 0]
        d.tile_counters = counters.data_ptr()
        d.tile_m, d.tile_n, d.ksplit, d.kwaves, d.stages = hit[0], hit[1], hit[2], hit[3], hit[4]
        d.tail_tiles, d.tail_ksplit = (hit[5], hit[6]) if len(hit) > 6 else (0, 0)
        d.grid_wgs = hit[7] if len(hit) > 7 else 0
        t_old = time_desc(d)
        flops = 2.0 * M * spec[4] * d.k_pad
        print(f'== {sig}: tuned {hit} {t_old:.1f} us {flops / t_old / 1e6:.1f} TF', flush=True)
        tiles = -(-M // 64) * -(-spec[4] // 64)
        rows = []
        d.tile_m, d.tile_n, d.kwaves = 64, 64, 0
        ksplits = [1]
        if tiles < 2 * CUS:
            ksplits = sorted({k for k in (1, 2, 3, 4, 6, 8, 12) if k <= nkt and tiles * k <= 6 * CUS})
        for ks in ksplits:
            items0 = tiles * ks
            for ns in (2, 3, 4, 6, 8):
                lds = ns * 16384 + 16384 + 16          # ring + the deferred epilogue tile
                for per_cu in (1, 2, 3, 4):
                    if per_cu * lds > 160 * 1024:
                        continue
                    g = CUS * per_cu
                    if g >= items0:
                        if per_cu > 1 and CUS * (per_cu - 1) >= items0:
                            continue                 # same launch as the smaller per_cu
                        cands = [(0, 0, items0)]
                    else:
                        # tail: the tiles of the last, partial round are split so that they fill one round of workgroups
                        cands = [(0, 0, g)]
                        if ks == 1:
                            tt = tiles % g
                            if tt:
                                for sp in sorted({min(nkt, max(1, g // tt)), min(nkt, max(1, (g // 2) // tt)), 2}):
                                    if sp > 1:
                                        cands.append((tt, sp, g))
                    for tt, sp, gw in cands:
                        d.ksplit, d.stages, d.tail_tiles, d.tail_ksplit, d.grid_wgs = ks, 40 + ns, tt, sp, gw
                        if hip.conv_workspace_bytes(d) > ws.numel():
                            continue
                        t = time_desc(d, iters=20, reps=2)
                        if t is not None:
                            rows.append((t, ks, ns, per_cu, tt, sp, gw))
        rows.sort()
        for t, ks, ns, per_cu, tt, sp, gw in rows[:6]:
            print(f'   pers ks={ks} ring={ns} wg/cu={per_cu} tail=({tt},{sp}) grid={gw}: {t:7.1f} us {flops / t / 1e6:6.1f} TF')
        total_old += t_old
        if rows and rows[0][0] < t_old * 0.985:
            t, ks, ns, per_cu, tt, sp, gw = rows[0]
            total_new += t
            if write:
                tuned[sig] = [64, 64, ks, 0, 40 + ns, tt, sp, gw]
        else:
            total_new += t_old
    print(f'sum over shapes: tuned {total_old:.1f} us -> with persistent winners {total_new:.1f} us')
    if write:
        # (on a gpurun box only gpurun_out/ travels back: the table is written there as well)
        outs = [TUNED] + ([os.path.join(os.path.dirname(os.path.dirname(TUNED)), 'gpurun_out', 'tuned_gfx950.json')]
                          if os.path.isdir(os.path.join(os.path.dirname(os.path.dirname(TUNED)), 'gpurun_out')) else [])
        for path in outs:
            with open(path, 'w') as f:
                json.dump(tuned, f, indent=0, sort_keys=True)
        print('tuned table updated:', outs)


if __name__ == '__main__':
    main()
