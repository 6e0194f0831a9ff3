#!/usr/bin/env python3
"""Timeline of a dependent chain of bs=1 bottleneck convs on ONE clock.  tools/chain_trace.py stamps s_memtime, whose base differs
between CU groups (found in round 4: up to 16 ms apart inside one XCD), so its cross-workgroup figures ("stragglers", "boundary") are
not trustworthy; here the trace build stamps s_memrealtime (100 MHz, chip-wide) and the stamped launches are CAPTURED in a hipGraph
(eager launches are spaced by the host).  Per launch: dispatch ramp (first -> last workgroup entry), duration (first entry -> last
exit), the phases of the workgroup that exits LAST (what the next launch waits for), and the dead time to the next launch.

    python tools/chain_trace_rt.py [bs] [hw] [blocks]
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ['YM_LIB_PATH'] = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'trace', 'libyolact_hip_trace.so')
import torch  # noqa: E402
from yolact_minimal_amd import hip  # noqa: E402
from tools.conv_sweep import make_desc  # noqa: E402

bs = int(sys.argv[1]) if len(sys.argv) > 1 else 1
hw = int(sys.argv[2]) if len(sys.argv) > 2 else 34
nblocks = int(sys.argv[3]) if len(sys.argv) > 3 else 4
wide, mid = {136: (256, 64), 68: (512, 128), 34: (1024, 256), 17: (2048, 512)}[hw]
dev = torch.device('cuda:0')
REGION = 4096 * 4
ws = torch.empty(1 << 27, dtype=torch.uint8, device=dev)
counters = torch.zeros(hip.TILE_COUNTERS, dtype=torch.int32, device=dev)
tuned = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'yolact_minimal_amd', 'tuned_gfx950.json')))
specs = [(bs, hw, hw, wide, mid, 1, 1, 0), (bs, hw, hw, mid, mid, 3, 1, 0), (bs, hw, hw, mid, wide, 1, 1, 1)]
descs = []
x_wide = torch.randn(bs, hw, hw, wide, device=dev)
for spec in specs:
    d, keep = make_desc(*spec, dev)
    sig = f'M{spec[0] * d.Ho * d.Wo}_N{spec[4]}_C{spec[3]}_k{spec[5]}_s{spec[6]}_seg1_r{spec[7]}'
    hit = tuned.get(sig, [0, 0, 0, 0, 0, 0, 0])
    ov = os.environ.get('YM_CHAIN_CFG_' + str(len(descs)))          # e.g. "64,64,6,0,2,0,0": override the tuned entry of conv 0 / 1 / 2
    if ov:
        hit = [int(v) for v in ov.split(',')]
    d.tile_m, d.tile_n, d.ksplit, d.kwaves, d.stages = hit[0], hit[1], hit[2], hit[3], hit[4]
    d.tail_tiles, d.tail_ksplit = (hit[5], hit[6]) if len(hit) > 6 else (0, 0)
    d.tile_counters = counters.data_ptr()
    descs.append((sig, hit, d, keep))
descs[0][2].inp = x_wide.data_ptr()
descs[1][2].inp = descs[0][3][4].data_ptr()
descs[2][2].inp = descs[1][3][4].data_ptr()
descs[2][2].seg[0].out = x_wide.data_ptr()
descs[2][2].scale = None
n = 3 * nblocks
trace = torch.zeros(n * REGION, dtype=torch.int64, device=dev)


def run(stamp):
    for k in range(n):
        if stamp:
            os.environ['YM_TRACE_PTR'] = str(trace.data_ptr() + k * REGION * 8)
            os.environ['YM_TRACE_REALTIME'] = os.environ.get('YM_CHAIN_RT', '1')
        else:
            os.environ.pop('YM_TRACE_PTR', None)
        hip.conv2d_fwd(descs[k % 3][2], ws)


def graph_of(stamp):
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        run(stamp)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            run(stamp)
    return g


e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for name, stamp in (('no stamps', False), ('stamped', True)):
    g = graph_of(stamp)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(10):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    print(f'chain of {n} launches, hipGraph replay, {name}: {e0.elapsed_time(e1) * 1e2 / n:.2f} us per launch')
raw = trace.cpu().reshape(n, REGION // 4, 4).clone()
raw[:, :, 0] &= (1 << 60) - 1
raw = raw.double() / 100.0             # s_memrealtime: 100 MHz -> us
print('per launch [us]: WGs | ramp (first->last entry) | duration (first entry->last exit) | median exit, last exit after first entry | '
      'LAST workgroup: entry +, prologue / K loop / epilogue | mean prologue / K loop / epilogue | dead time to next launch')
tot = {}
for k in range(n):
    r = raw[k][(raw[k, :, 0] > 0) & (raw[k, :, 3] > 0)]
    if not r.shape[0]:
        continue
    t0 = float(r[:, 0].min())
    last = int(r[:, 3].argmax())
    L = r[last]
    nxt = raw[k + 1][(raw[k + 1, :, 0] > 0)] if k + 1 < n else None
    gap = float(nxt[:, 0].min() - r[:, 3].max()) if nxt is not None and nxt.shape[0] else float('nan')
    row = dict(wgs=r.shape[0], ramp=float(r[:, 0].max()) - t0, dur=float(r[:, 3].max()) - t0, med_exit=float(r[:, 3].median()) - t0,
               l_entry=float(L[0]) - t0, l_pro=float(L[1] - L[0]), l_k=float(L[2] - L[1]), l_epi=float(L[3] - L[2]),
               pro=float((r[:, 1] - r[:, 0]).mean()), kl=float((r[:, 2] - r[:, 1]).mean()), epi=float((r[:, 3] - r[:, 2]).mean()), gap=gap)
    if k >= 3:                                      # (skip the first block: cold)
        t = tot.setdefault(k % 3, [])
        t.append(row)
    print(f"  {k:2d} {descs[k % 3][0]:32s} {row['wgs']:4d} | {row['ramp']:5.2f} | {row['dur']:6.2f} | {row['med_exit']:6.2f} {row['dur']:6.2f} | "
          f"+{row['l_entry']:5.2f}  {row['l_pro']:5.2f} / {row['l_k']:5.2f} / {row['l_epi']:5.2f} | {row['pro']:5.2f} / {row['kl']:5.2f} / {row['epi']:5.2f} | {gap:6.2f}")
print('means over blocks 1..:')
for c, rows in sorted(tot.items()):
    m = {k_: sum(r_[k_] for r_ in rows if r_[k_] == r_[k_]) / max(1, sum(1 for r_ in rows if r_[k_] == r_[k_])) for k_ in rows[0]}
    print(f"  {descs[c][0]:32s} {descs[c][1]} dur {m['dur']:6.2f} = ramp {m['ramp']:5.2f}; last WG: entry +{m['l_entry']:5.2f}, prologue {m['l_pro']:5.2f}, "
          f"K loop {m['l_k']:5.2f}, epilogue {m['l_epi']:5.2f}; median exit {m['med_exit']:6.2f}; dead time to next {m['gap']:5.2f}")
