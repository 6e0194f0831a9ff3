#!/usr/bin/env python3
"""Per-shape table of the conv launches of ONE training step (forward / data gradient / weight gradient): launches per step, the
launch timed alone on the chip (HIP events, back to back), GFLOP, TFLOP/s against the 157.3 f32 MFMA peak, the tuned row.
Ranks where the MFMA time of the step goes and which shapes sit furthest below the peak.

    python tools/train_layer_table.py [--cfg res101_coco] [--batch 8] [--json out.json]
"""
import argparse
import ctypes
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yolact_minimal_amd.utils.synthetic import synth_targets  # noqa: E402
from yolact_minimal_amd.config import build_cfg  # noqa: E402
from yolact_minimal_amd.modules.yolact import Yolact  # noqa: E402
from yolact_minimal_amd.trainer import Trainer  # noqa: E402
from yolact_minimal_amd import train_engine as T, hip  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--cfg', default='res101_coco')
ap.add_argument('--batch', type=int, default=8)
ap.add_argument('--iters', type=int, default=10)
ap.add_argument('--json', default='')
ap.add_argument('--tune-ws', action='store_true', help='forward launches with fused statistics that the weight-stationary 1x1 kernel covers: time its tiles x rings, collect <key>_st rows')
ap.add_argument('--write', action='store_true', help='with --tune-ws: write the winning <key>_st rows (training-only rows) into the tuned table')
args = ap.parse_args()
os.environ['YM_WGRAD_STREAM'] = '0'
T._WGRAD_STREAM = False
dev = torch.device('cuda:0')
cfg = build_cfg(args.cfg, 'train', 544, train_bs=args.batch, bs_per_gpu=args.batch)
torch.manual_seed(0)
tr = Trainer(Yolact(cfg), cfg, dev)
img = torch.randn(args.batch, 3, 544, 544, device=dev)
boxes, masks = synth_targets(args.batch, 544, seed=0)
boxes, masks = [b.to(dev) for b in boxes], [m.to(dev) for m in masks]
tr.step(img, boxes, masks)
T.launch_counts = {}
tr.step(img, boxes, masks)
torch.cuda.synchronize()
counts, T.launch_counts = T.launch_counts, None
big = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
# operands large enough for any layer (contents do not matter for the timing; stale pointers of the step are not reused)
opa = torch.randn(1 << 28, device=dev)       # 1 GiB of floats
opb = torch.randn(1 << 26, device=dev)
out = torch.empty(1 << 28, device=dev)
stats = torch.zeros(1 << 20, dtype=torch.float64, device=dev)
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def timeit(fn):
    fn(); fn()
    best = 1e30
    for _ in range(2):
        torch.cuda.synchronize()
        ev0.record()
        for _ in range(args.iters):
            fn()
        ev1.record()
        torch.cuda.synchronize()
        best = min(best, ev0.elapsed_time(ev1) / args.iters * 1e3)
    return best


rows = []
ws_rows = {}
for key, n in counts.items():
    ent = T._desc_cache[key]
    d = ent[0]
    if key[0] in ('f', 'd'):
        M = d.B * d.Ho * d.Wo
        N, K = d.Cout, d.KH * d.KW * d.Cin
        # a dgrad with stride 2 only multiplies 1/4 of the taps per pixel (parity classes)
        gflop = 2.0 * M * N * K / 1e9 / (d.stride * d.stride if key[0] == 'd' else 1)
        d.inp, d.weight = opa.data_ptr(), opb.data_ptr()
        for i in range(d.nseg):
            d.seg[i].out = out.data_ptr()
        if d.residual:
            d.residual = out.data_ptr()
        if d.shift:
            d.shift = opb.data_ptr()
        if d.bn_sum:
            d.bn_sum, d.bn_sumsq = stats.data_ptr(), stats.data_ptr() + 8 * d.Cout
        if d.bnb_y:
            d.bnb_y = opa.data_ptr()
            d.bnb_out = opa.data_ptr() if d.bnb_out else None
            d.bnb_mean = d.bnb_invstd = d.bnb_gamma = opb.data_ptr()
            d.bnb_beta = opb.data_ptr() if d.bnb_beta else None
        us = timeit(lambda: hip.conv2d_fwd(d, big))
        cfg_s = f'{d.tile_m}x{d.tile_n} ks{d.ksplit} st{d.stages} tail{d.tail_tiles}x{d.tail_ksplit}' + (' +bnsum' if d.bn_sum else '') + (' +bnb' if d.bnb_y else '')
        if (args.tune_ws and key[0] == 'f' and d.bn_sum and (d.KH, d.KW, d.stride, d.pad, d.nseg) == (1, 1, 1, 0, 1) and d.Cin % 32 == 0 and
                d.Cin <= 256 and d.seg[0].act in (0, 1)):
            keep = (d.tile_m, d.tile_n, d.ksplit, d.kwaves, d.stages, d.tail_tiles, d.tail_ksplit, d.grid_wgs)
            best = (us, None)
            for tm, tn in ((64, 256), (128, 128), (256, 64)):
                if tn * d.Cin * 4 > 64 * 1024:
                    continue
                for st in (52, 53, 54):
                    d.tile_m, d.tile_n, d.ksplit, d.kwaves, d.stages, d.tail_tiles, d.tail_ksplit, d.grid_wgs = tm, tn, 1, 0, st, 0, 0, 0
                    t = timeit(lambda: hip.conv2d_fwd(d, big))
                    if t < best[0]:
                        best = (t, [tm, tn, 1, 0, st, 0, 0])
            d.tile_m, d.tile_n, d.ksplit, d.kwaves, d.stages, d.tail_tiles, d.tail_ksplit, d.grid_wgs = keep
            name = f'M{M}_N{N}_C{d.Cin}_k1_s1_seg1_r{int(bool(d.residual))}_st'
            print(f'  ws: {name:44s} x{n:2d} {us:7.1f} -> {best[0]:7.1f} us {best[1]}', flush=True)
            if best[1] is not None and best[0] < 0.97 * us:
                ws_rows[name] = best[1]
                us = best[0]
                cfg_s = f'{best[1][0]}x{best[1][1]} st{best[1][4]} (weight-stationary) +bnsum'
        kind = 'fwd' if key[0] == 'f' else 'dgrad'
        shape = f'M{M} N{N} K{K} k{d.KH} s{d.stride}'
    else:
        M = d.B * d.Ho * d.Wo
        N, K = d.Cout, d.KH * d.KW * d.Cin
        gflop = 2.0 * M * N * K / 1e9
        d.x, d.dy, d.dw = opa.data_ptr(), opa.data_ptr(), out.data_ptr()
        if d.row_end[0]:
            d.dw_seg[0] = d.dw_seg[1] = out.data_ptr()
        us = timeit(lambda: hip.check(hip.lib().ym_conv2d_wgrad(ctypes.byref(d), ctypes.c_void_p(big.data_ptr()), big.numel(), hip.stream_ptr()), 'wgrad'))
        cfg_s = f'msplit{d.msplit} lds{d.lds_buffers}'
        kind = 'wgrad'
        shape = f'M{M} N{N} K{K} k{d.KH} s{d.stride}'
    rows.append(dict(kind=kind, shape=shape, n=n, us=us, gflop=gflop, cfg=cfg_s))
for r in rows:
    r['tf'] = r['gflop'] / r['us'] * 1e3                                  # GFLOP / us = PFLOP/s -> TFLOP/s
    r['tot_ms'] = r['n'] * r['us'] / 1e3
    r['lost_ms'] = r['tot_ms'] - r['n'] * r['gflop'] / 157.3               # GFLOP / (TFLOP/s) = ms at the f32 MFMA peak
tot = {}
for r in rows:
    t = tot.setdefault(r['kind'], [0.0, 0.0])
    t[0] += r['tot_ms']
    t[1] += r['n'] * r['gflop']
for k, (ms, gf) in tot.items():
    print(f'{k:6s} {ms:8.3f} ms/step alone  {gf:9.1f} GFLOP  {gf / ms:6.1f} TFLOP/s = {gf / ms / 157.3:.3f} of peak')
print(f'{"kind":6s} {"shape":34s} {"n":>3s} {"us":>8s} {"tot ms":>7s} {"TF/s":>6s} {"lost ms":>7s}  tuned')
for r in sorted(rows, key=lambda r: -r['lost_ms']):
    print(f'{r["kind"]:6s} {r["shape"]:34s} {r["n"]:3d} {r["us"]:8.1f} {r["tot_ms"]:7.3f} {r["tf"]:6.1f} {r["lost_ms"]:7.3f}  {r["cfg"]}')
if args.json:
    json.dump(rows, open(args.json, 'w'), indent=1)
if args.tune_ws:
    print(f'{len(ws_rows)} weight-stationary <key>_st rows:', json.dumps(ws_rows))
    if args.write and ws_rows:
        from yolact_minimal_amd import engine as E
        table = json.load(open(E.TUNED_PATH))
        table.update(ws_rows)
        json.dump(table, open(E.TUNED_PATH, 'w'), indent=0, sort_keys=True)
        print('written to', E.TUNED_PATH)
