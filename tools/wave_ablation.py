#!/usr/bin/env python3
"""Where does a wave-DMA conv launch (conv_wdma_f32) lose its time?  Trace build + YM_PERS_ABL: complete (0), operand stream only
(1: the DMA ring and its counted waits, no LDS reads, no MFMAs), LDS reads + MFMAs only (2), MFMAs only (3), on the three convs of a
res101 layer3 bottleneck at batch 1 with their tuned configurations (and without the tail split).

    make -C yolact_minimal_amd/csrc trace && python tools/wave_ablation.py

Round 4 left the K loop of these launches at 0.8-1.0 us per K tile for 0.43 us of MFMAs with three suspects ruled out (LDS latency,
prefetch distance, tile order: DESIGN.md section 8); this table says which half -- the operand stream or the LDS / MFMA side -- holds
the time."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ['YM_LIB_PATH'] = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'trace', 'libyolact_hip_trace.so')
import torch  # noqa: E402
from yolact_minimal_amd import hip  # noqa: E402
from tools.conv_sweep import make_desc  # noqa: E402

dev = torch.device('cuda:0')
ws = torch.empty(1 << 26, dtype=torch.uint8, device=dev)
counters = torch.zeros(hip.TILE_COUNTERS, dtype=torch.int32, device=dev)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def timeit(d, iters=50):
    for _ in range(3):
        hip.conv2d_fwd(d, ws)
    best = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(iters):
            hip.conv2d_fwd(d, ws)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters * 1e3)
    return best


CASES = (  # (B, H, W, Cin, Cout, k, stride, residual), [(tile_m, tile_n, kwaves, stages, tail_tiles, tail_ksplit, waves per workgroup)]
    ((1, 34, 34, 1024, 256, 1, 1, 0), [(32, 32, 4, 22, 40, 4, 0), (32, 32, 4, 22, 0, 0, 0), (32, 32, 4, 23, 0, 0, 0)]),
    ((1, 34, 34, 256, 256, 3, 1, 0), [(32, 32, 4, 22, 40, 6, 0), (32, 32, 4, 22, 0, 0, 0), (32, 32, 4, 23, 0, 0, 0)]),
    ((1, 34, 34, 256, 1024, 1, 1, 1), [(32, 32, 1, 22, 0, 0, 1), (32, 32, 1, 22, 0, 0, 4), (32, 32, 1, 23, 0, 0, 1)]),
)
for spec, variants in CASES:
    d, keep = make_desc(*spec, dev)
    M = spec[0] * d.Ho * d.Wo
    flops = 2.0 * M * spec[4] * d.k_pad
    d.tile_counters = counters.data_ptr()
    print(f'== M{M} N{spec[4]} K{d.k_pad}: MFMA time at 155 TFLOP/s {flops / 155e6:.1f} us, {-(-M // 32) * -(-spec[4] // 32)} tiles of 32x32')
    for tm, tn, kwv, stg, tt, ts, wpb in variants:
        d.tile_m, d.tile_n, d.ksplit, d.kwaves, d.stages, d.tail_tiles, d.tail_ksplit, d.grid_wgs = tm, tn, 1, kwv, stg, tt, ts, wpb
        row = []
        for abl in (0, 1, 2, 3):
            os.environ['YM_PERS_ABL'] = str(abl)
            row.append(timeit(d))
        os.environ['YM_PERS_ABL'] = '0'
        print(f'   K waves {kwv}, ring {stg - 20}, tail {tt}x{ts}, {wpb or 4} waves/workgroup: complete {row[0]:6.1f} us ({flops / row[0] / 1e6:5.1f} TF) | '
              f'operand stream only {row[1]:6.1f} | LDS reads + MFMAs {row[2]:6.1f} | MFMAs only {row[3]:6.1f}', flush=True)
