#!/usr/bin/env python3
"""Which 1x1 / stride-1 layers of the batch-8 plans run faster on the weight-stationary kernel (csrc/conv_ws.hip)?  Every eligible
shape (Cin <= 256, one plain NHWC output, ReLU / identity) of the res101 / res50 / swin_tiny plans is timed alone with its
current table row and with the kernel's tiles x ring depths; winners (>= 3 % faster) are merged into the tuned table through the
reference-digest gate (tools/table_gate.py).

    python tools/tune_ws.py [--batch 8] [--write] [--out rows.json]
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402
from yolact_minimal_amd import hip, engine as E  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=8)
ap.add_argument('--cfgs', default='res101_coco,swin_tiny_coco')
ap.add_argument('--write', action='store_true')
ap.add_argument('--out', default='')
ap.add_argument('--iters', type=int, default=20)
args = ap.parse_args()
dev = torch.device('cuda:0')
big = torch.empty(1 << 28, device=dev, dtype=torch.uint8)
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def launch_time(d):
    if hip.conv_workspace_bytes(d) > big.numel():
        return None
    try:
        for _ in range(3):
            hip.conv2d_fwd(d, big)
    except RuntimeError:
        return None
    best = 1e30
    for _ in range(3):
        torch.cuda.synchronize()
        ev0.record()
        for _ in range(args.iters):
            hip.conv2d_fwd(d, big)
        ev1.record()
        torch.cuda.synchronize()
        best = min(best, ev0.elapsed_time(ev1) / args.iters * 1e3)
    return best


rows, seen = {}, set()
tot_old = tot_new = 0.0
for cfg_name in args.cfgs.split(','):
    net, cfg = bench.build_net(cfg_name, 544, dev)
    img = torch.randn(args.batch, 3, 544, 544, device=dev)
    eng = net._engine(img)
    eng.run(img)
    torch.cuda.synchronize()
    for c in eng.convs:
        d = c.desc
        if c.sig in seen or c.stem or d.nlevels or d.nseg != 1 or (d.KH, d.KW, d.stride, d.pad) != (1, 1, 1, 0):
            continue
        if d.Cin % 32 or d.Cin > 256 or c.act not in (E.ACT_NONE, E.ACT_RELU):
            continue
        seen.add(c.sig)
        n_same = sum(1 for o in eng.convs if o.sig == c.sig)
        keep = (d.tile_m, d.tile_n, d.ksplit, d.kwaves, d.stages, d.tail_tiles, d.tail_ksplit, d.grid_wgs)
        t0 = launch_time(d)
        best = (t0, None)
        for tm, tn in ((64, 256), (128, 128), (256, 64)):
            if tn * d.Cin * 4 > 64 * 1024:
                continue
            for st in (52, 53, 54):
                d.tile_m, d.tile_n, d.ksplit, d.kwaves, d.stages, d.tail_tiles, d.tail_ksplit, d.grid_wgs = tm, tn, 1, 0, st, 0, 0, 0
                t = launch_time(d)
                if t is not None and t < best[0]:
                    best = (t, [tm, tn, 1, 0, st, 0, 0])
        d.tile_m, d.tile_n, d.ksplit, d.kwaves, d.stages, d.tail_tiles, d.tail_ksplit, d.grid_wgs = keep
        gain = best[1] is not None and best[0] < 0.97 * t0
        gf = c.flops / 1e9
        print(f'{cfg_name:14s} {c.sig:40s} x{n_same:2d} {t0:7.1f} us ({gf / t0 * 1e3:5.1f} TF) -> {best[0]:7.1f} us ({gf / best[0] * 1e3:5.1f} TF) '
              f'{best[1]}{"  <-- kept" if gain else ""}', flush=True)
        tot_old += n_same * t0
        tot_new += n_same * (best[0] if gain else t0)
        if gain:
            rows[c.sig] = best[1]
    del eng, net
    torch.cuda.empty_cache()
print(f'eligible launches: {tot_old / 1e3:.3f} -> {tot_new / 1e3:.3f} ms per batch-{args.batch} forward(s); {len(rows)} rows')
if args.out:
    json.dump(rows, open(args.out, 'w'), indent=0, sort_keys=True)
if args.write and rows:
    from tools.table_gate import merge_rows, GateRefused
    torch.cuda.synchronize()
    try:
        merge_rows(rows, E.TUNED_PATH, extra_tests=('tests/test_gpu_swin.py::test_swin_forward_544_bs8_digest_under_the_tuned_plan',))
        print(f'wrote {len(rows)} rows to {E.TUNED_PATH} (reference digests green under the candidate table)')
    except GateRefused as exc:
        print(f'REFUSED: {exc}')
        sys.exit(3)
