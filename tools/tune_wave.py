#!/usr/bin/env python3
"""Wave-private DMA-ring kernel (conv_wdma_f32, round 4) against the tuned choice, for every conv of an inference plan: time the
tuned entry and the candidates {32x32, 64x32, 32x64 wave tile} x {1, 2, 4 waves per tile (K split inside the workgroup)} x {ring of
2, 3, 4}, check every candidate's output against the tuned kernel's (same inputs; the K-sum order differs, so to 1e-5 of the
largest output), and write the winners ([tile_m, tile_n, 1, kwaves, stages, 0, 0]).

    python tools/tune_wave.py [--batch 1] [--cfg res101_coco] [--write] [--out file.json] [--max-m 20000]
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
ap.add_argument('--batch', type=int, default=1)
ap.add_argument('--cfg', default='res101_coco')
ap.add_argument('--write', action='store_true')
ap.add_argument('--out', default='')
ap.add_argument('--iters', type=int, default=30)
ap.add_argument('--max-m', type=int, default=20000, help='skip layers with more GEMM rows (the LDS-tiled kernels own the large layers)')
ap.add_argument('--margin', type=float, default=0.97)
args = ap.parse_args()
dev = torch.device('cuda:0')
net, cfg = bench.build_net(args.cfg, 544, dev)
img = torch.randn(args.batch, 3, 544, 544, device=dev)
eng = net._engine(img)
eng.run(img)
torch.cuda.synchronize()
big = torch.empty(1 << 28, device=dev, dtype=torch.uint8)
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def time_desc(d):
    if hip.conv_workspace_bytes(d) > big.numel():
        return None
    try:
        for _ in range(3):
            hip.conv2d_fwd(d, big)
    except RuntimeError:
        return None
    best = 1e30
    for _ in range(4):
        torch.cuda.synchronize()
        ev0.record()
        for _ in range(args.iters):
            hip.conv2d_fwd(d, big)
        ev1.record()
        torch.cuda.synchronize()
        best = min(best, ev0.elapsed_time(ev1) / args.iters * 1e3)
    return best


table = E.tuned_table()
new, tot0, tot1, seen, worst = {}, 0.0, 0.0, set(), 0.0
for c in eng.convs:
    d = c.desc
    M = d.B * d.Ho * d.Wo
    if c.sig in seen or c.stem or d.nlevels or d.Cin % 32 or M > args.max_m:
        continue
    seen.add(c.sig)
    keep = (d.tile_m, d.tile_n, d.ksplit, d.kwaves, d.stages, d.tail_tiles, d.tail_ksplit, d.grid_wgs)

    def set_cfg(v):
        d.tile_m, d.tile_n, d.ksplit, d.kwaves, d.stages, d.tail_tiles, d.tail_ksplit, d.grid_wgs = v

    # reference output of the tuned kernel on the plan's own buffers (single plain output only; segmented heads: checked by the tests)
    ref = None
    if d.nseg == 1:
        hip.conv2d_fwd(d, big)
        torch.cuda.synchronize()
        n_out = M * d.Cout
        ref = torch.empty(n_out, device=dev, dtype=torch.float32)
        # copy out of the raw output pointer through a view tensor built on the engine's buffer that owns it
        owner = next((b for b in eng._bufs if b.data_ptr() <= d.seg[0].out < b.data_ptr() + b.numel() * 4), None)
        if owner is not None:
            off = (d.seg[0].out - owner.data_ptr()) // 4
            view = owner.reshape(-1)[off:off + n_out]
            ref.copy_(view)
        else:
            ref = None
    base = time_desc(d)
    best = (base, keep)
    for tm, tn in ((32, 32), (64, 32), (32, 64)):
        for kwv in (1, 2, 4):
            if kwv > d.k_pad // 32:
                continue
            for stg in (22, 23, 24):
                if stg == 24 and (tm, tn) != (32, 32):
                    continue
                cand = (tm, tn, 1, kwv, stg, 0, 0, 0)
                set_cfg(cand)
                t = time_desc(d)
                if t is None:
                    continue
                if ref is not None:
                    hip.conv2d_fwd(d, big)
                    torch.cuda.synchronize()
                    err = float((view - ref).abs().max()) / max(1e-30, float(ref.abs().max()))
                    worst = max(worst, err)
                    if not err < 1e-5:
                        print(f'  MISMATCH {c.sig} {cand}: max rel err {err:.3e}', flush=True)
                        continue
                if os.environ.get('YM_TUNE_VERBOSE') and c.sig in os.environ['YM_TUNE_VERBOSE']:
                    print(f'      {cand[:5]} {t:7.2f} us', flush=True)
                if t < best[0] * args.margin:
                    best = (t, cand)
    set_cfg(keep)
    tot0 += base
    tot1 += best[0]
    print(f'{c.sig:42s} {list(keep[:7])} {base:7.2f} us -> {list(best[1][:7])} {best[0]:7.2f} us', flush=True)
    if best[1] != keep:
        new[c.sig] = list(best[1][:7])
print(f'{len(seen)} shapes (M <= {args.max_m}): {tot0:.1f} -> {tot1:.1f} us summed over distinct shapes; {len(new)} entries change; worst candidate error {worst:.2e}')
if args.out:
    json.dump(new, open(args.out, 'w'), indent=0, sort_keys=True)
if args.write:
    table.update(new)
    json.dump(table, open(E.TUNED_PATH, 'w'), indent=0, sort_keys=True)
