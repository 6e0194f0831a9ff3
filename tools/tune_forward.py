#!/usr/bin/env python3
"""Coordinate descent on the forward time of ONE request (hipGraph replay of the whole plan), shape by shape, over the wave-private
DMA-ring kernel's configurations.  The per-launch tuner (tools/tune_wave.py, a launch repeated back to back) undervalues kernels
without a cross-workgroup K-slice exchange: in the dependent chain of a real forward the exchange's last arriver delays the NEXT
launch (tools/chain_trace_rt.py: a layer3 block 20.7 -> 17.6 us per launch with the wave kernels, where back-to-back timing of the
3x3 shows a tie).  So: candidates are pre-filtered by their own launch time (within --slack of the best) and then judged by the
plan's forward time; a change is kept if it wins twice.

    python tools/tune_forward.py [--batch 1] [--cfg res101_coco] [--write] [--out file.json]
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
ap.add_argument('--max-m', type=int, default=20000)
ap.add_argument('--slack', type=float, default=1.2)
ap.add_argument('--replays', type=int, default=40)
ap.add_argument('--size', type=int, default=544, help='--img_size of the plan that is tuned')
ap.add_argument('--alt', default='', help='comma-separated table files: their row for a shape (exact, or transferred from the nearest tuned '
                'shape: plan_transfer.py) is a candidate too, whatever its own launch time')
ap.add_argument('--inflight', type=int, default=1, help='> 1: optimise the img/s of a RequestPipeline with that many requests in flight; winners are written as <sig>_tp')
args = ap.parse_args()
dev = torch.device('cuda:0')
net, cfg = bench.build_net(args.cfg, args.size, dev)
img = torch.randn(args.batch, 3, args.size, args.size, device=dev)
alts = [json.load(open(f)) for f in args.alt.split(',') if f]
pipe = None
if args.inflight > 1:
    os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
    from yolact_minimal_amd.pipeline import RequestPipeline
    pipe = RequestPipeline(net, cfg, args.size, args.size, dev, depth=args.inflight, with_post=False, batch=args.batch, return_outputs=False)
    pipe.warm_up(img)
    eng = pipe.engines[0]
else:
    eng = net._engine(img)
    eng.run(img)
torch.cuda.synchronize()
big = torch.empty(1 << 28, device=dev, dtype=torch.uint8)
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def launch_time(d, iters=20):
    if hip.conv_workspace_bytes(d) > big.numel():
        return None
    try:
        for _ in range(2):
            hip.conv2d_fwd(d, big)
    except RuntimeError:
        return None
    best = 1e30
    for _ in range(3):
        torch.cuda.synchronize()
        ev0.record()
        for _ in range(iters):
            hip.conv2d_fwd(d, big)
        ev1.record()
        torch.cuda.synchronize()
        best = min(best, ev0.elapsed_time(ev1) / iters * 1e3)
    return best


def forward_ms():
    if pipe is not None:              # ms per request with `inflight` requests overlapped
        import time
        pipe.warm_up(img)
        best = 1e30
        n = args.replays * args.inflight
        for _ in range(2):
            for _ in range(2 * args.inflight):
                pipe.submit(img)
            pipe.drain()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                pipe.submit(img)
            pipe.drain()
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / n * 1e3)
        return best
    eng.run(img)                      # (re)captures the graph after a retune
    torch.cuda.synchronize()
    best = 1e30
    for _ in range(2):
        ev0.record()
        for _ in range(args.replays):
            eng.run(img)
        ev1.record()
        torch.cuda.synchronize()
        best = min(best, ev0.elapsed_time(ev1) / args.replays)
    return best


def get(c):
    return [c.tile[0], c.tile[1], c.ksplit, c.kwaves, c.stages, c.tail[0], c.tail[1], c.grid_wgs]


def put(sig, v):
    for e in (pipe.engines if pipe is not None else [eng]):
        for c in e.convs:
            if c.sig == sig:
                c.tile, c.ksplit, c.kwaves, c.stages, c.tail, c.grid_wgs = (v[0], v[1]), v[2], v[3], v[4], (v[5], v[6]), v[7]
        e.retune()


groups = {}
for c in eng.convs:
    d = c.desc
    if c.stem or d.nlevels or d.Cin % 32 or d.B * d.Ho * d.Wo > args.max_m:
        continue
    groups.setdefault(c.sig, []).append(c)
base_ms = forward_ms()
print(f'start: forward {base_ms:.4f} ms ({args.cfg}, batch {args.batch}); {len(groups)} shapes', flush=True)
cur_ms, kept = base_ms, {}
# shapes by their share of the launches' time (count x launch time)
order = []
for sig, cs in groups.items():
    t = launch_time(cs[0].desc)
    order.append((-(t or 0) * len(cs), sig, t))
for _, sig, t0 in sorted(order):
    cs = groups[sig]
    c, d = cs[0], cs[0].desc
    cur = get(c)
    M = d.B * d.Ho * d.Wo
    cands = []
    for tm, tn in ((32, 32), (64, 32), (32, 64)):
        for kwv in (1, 2, 4):
            if kwv > d.k_pad // 32:
                continue
            vs = [[tm, tn, 1, kwv, 22, 0, 0, 0]]
            if kwv < 4:                                   # fewer waves per workgroup: single tiles are balanced over the CUs
                vs += [[tm, tn, 1, kwv, 22, 0, 0, wpb] for wpb in (1, 2) if wpb >= kwv]
            tiles = -(-M // 32) * -(-d.Cout // 32)
            if (tm, tn, kwv) == (32, 32, 4) and tiles > 256 and d.nseg == 1 and d.tile_counters:
                # tail split: the tiles past the last full round of 256 CUs are computed as K slices in the CUs' second slots
                for ts in (4, 6, 8):
                    if ts * 2 <= d.k_pad // 32:
                        vs.append([tm, tn, 1, kwv, 22, tiles % 256 or 256, ts, 0])
            for v in vs:
                keep = (d.tile_m, d.tile_n, d.ksplit, d.kwaves, d.stages, d.tail_tiles, d.tail_ksplit, d.grid_wgs)
                d.tile_m, d.tile_n, d.ksplit, d.kwaves, d.stages, d.tail_tiles, d.tail_ksplit, d.grid_wgs = v
                t = launch_time(d)
                d.tile_m, d.tile_n, d.ksplit, d.kwaves, d.stages, d.tail_tiles, d.tail_ksplit, d.grid_wgs = keep
                if t is not None:
                    cands.append((t, v))
    if pipe is not None:              # the latency entry is a candidate too (the slot may currently run a `_tp` row)
        alt = E.tuned_table().get(sig)
        if alt:
            cands.append((0.0, list(alt[:7]) + [alt[7] if len(alt) > 7 else 0]))
    for tab in alts:                  # another table's choice for this shape (its own row, or the nearest tuned shape's, re-derived)
        from yolact_minimal_amd import plan_transfer
        row, _ = plan_transfer.lookup(tab, sig, M, d.Cout, d.k_pad // 32, d.nseg)
        if row is not None:
            row = list(row[:7]) + [0] * (7 - len(row[:7])) + [row[7] if len(row) > 7 else 0]
            if row != cur and all(row != v for _, v in cands):
                cands.append((0.0, row))
    cands.sort()
    tried = 0
    t_ref = min([t0] + [t for t, _ in cands if t > 0])
    for t, v in cands:
        if v == cur or t > args.slack * t_ref or tried >= (7 if pipe is not None else 5 + len(alts)):
            continue
        tried += 1
        put(sig, v)
        ms = forward_ms()
        tag = ''
        if ms < cur_ms * 0.998:
            ms2 = forward_ms()
            if max(ms, ms2) < cur_ms * 0.998:
                cur_ms, cur, kept[sig], tag = max(ms, ms2), v, (v if v[7] else v[:7]), '  <-- kept'
        print(f'{sig:42s} x{len(cs):2d} launch {t0:6.2f} -> {t:6.2f} us {v} forward {ms:.4f} ms{tag}', flush=True)
        put(sig, cur)
final = forward_ms()
print(f'final: forward {base_ms:.4f} -> {final:.4f} ms; {len(kept)} entries change', flush=True)
if pipe is not None:
    kept = {k + '_tp': v for k, v in kept.items()}
if args.out:
    json.dump(kept, open(args.out, 'w'), indent=0, sort_keys=True)
if args.write and kept:
    # rows change the summation order of their layers: they reach the committed table only through the reference-digest gate
    from tools.table_gate import merge_rows, GateRefused
    torch.cuda.synchronize()
    try:
        merge_rows(kept, E.TUNED_PATH)
        print(f'wrote {len(kept)} rows to {E.TUNED_PATH} (544 px reference digests green under the candidate table)')
    except GateRefused as exc:
        print(f'REFUSED: {exc}')
        sys.exit(3)
