#!/usr/bin/env python3
"""In-kernel phase timing of conv_igemm_f32 (debug build tools/trace/libyolact_hip_trace.so): per workgroup
s_memtime stamps at kernel entry, after the prologue (first tile staged), after the K loop, after the epilogue."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ['YM_LIB_PATH'] = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'trace', 'libyolact_hip_trace.so')
import torch  # noqa: E402
from yolact_minimal_amd import hip  # noqa: E402
from tools.conv_sweep import make_desc  # noqa: E402

dev = torch.device('cuda:0')
trace = torch.zeros(8192 * 4, dtype=torch.int64, device=dev)
os.environ['YM_TRACE_PTR'] = str(trace.data_ptr())
ws = torch.empty(1 << 27, dtype=torch.uint8, device=dev)
counters = torch.zeros(hip.TILE_COUNTERS, dtype=torch.int32, device=dev)
import json  # noqa: E402
tuned = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'yolact_minimal_amd', 'tuned_gfx950.json')))
SPECS = {'bs1': ((1, 34, 34, 256, 1024, 1, 1, 1), (1, 34, 34, 1024, 256, 1, 1, 0), (1, 34, 34, 256, 256, 3, 1, 0),
                 (1, 68, 68, 256, 256, 3, 1, 0), (8, 34, 34, 256, 256, 3, 1, 0), (8, 136, 136, 256, 256, 3, 1, 0)),
         'bs8': ((8, 34, 34, 256, 1024, 1, 1, 1), (8, 34, 34, 1024, 256, 1, 1, 0), (8, 34, 34, 256, 256, 3, 1, 0))}
for spec in SPECS[sys.argv[1] if len(sys.argv) > 1 else 'bs1']:
    d, keep = make_desc(*spec, dev)
    sig = f'M{spec[0] * d.Ho * d.Wo}_N{spec[4]}_C{spec[3]}_k{spec[5]}_s{spec[6]}_seg1_r{spec[7]}'
    hit = tuned.get(sig, [0, 0, 0, 0, 0, 0, 0])
    tile, ks = (hit[0], hit[1]), hit[2]
    d.tile_m, d.tile_n, d.ksplit, d.kwaves, d.stages = hit[0], hit[1], hit[2], hit[3], hit[4]
    d.tail_tiles, d.tail_ksplit = (hit[5], hit[6]) if len(hit) > 6 else (0, 0)
    d.tile_counters = counters.data_ptr()
    print(sig, hit)
    for stg in (2, 22, 23, 33, 34):
        d.stages = stg
        for _ in range(3):
            hip.conv2d_fwd(d, ws)
        torch.cuda.synchronize()
        trace.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); hip.conv2d_fwd(d, ws); e1.record()
        torch.cuda.synchronize()
        raw = trace.cpu().reshape(-1, 4).double()
        M = spec[0] * d.Ho * d.Wo
        spans, spreads, rows = [], [], []
        for x in range(8):                                   # every XCD has its own s_memtime base: compare inside one XCD only
            r = raw[x::8]
            r = r[(r[:, 0] > 0) & (r[:, 3] > 0)]
            if r.shape[0]:
                rows.append(r)
                spans.append(float(r[:, 3].max() - r[:, 0].min()))
                spreads.append(float(r[:, 0].max() - r[:, 0].min()))
        t = torch.cat(rows)
        print(f'M={M} N={spec[4]} K={d.k_pad} tile={tile} ks={ks} stages={stg}: {t.shape[0]} WGs stamped, event {e0.elapsed_time(e1) * 1e3:.1f} us')
        print(f'   (shader clock cycles) WG start spread per XCD {max(spreads):.0f}; prologue {((t[:, 1] - t[:, 0]).mean()):.0f}; '
              f'K loop {((t[:, 2] - t[:, 1]).mean()):.0f} (min {((t[:, 2] - t[:, 1]).min()):.0f}, max {((t[:, 2] - t[:, 1]).max()):.0f}); '
              f'epilogue {((t[:, 3] - t[:, 2]).mean()):.0f}; first start -> last end per XCD {max(spans):.0f}')
