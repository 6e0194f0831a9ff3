#!/usr/bin/env python3
"""Where does a 64x64-tile conv launch lose its time?  Trace build + YM_PERS_ABL: the persistent kernel with only its operand
stream (1), only its LDS reads + MFMAs (2), and complete (0), for ring depths / grid sizes, on bs=8 and bs=1 shapes."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ['YM_LIB_PATH'] = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'trace', 'libyolact_hip_trace.so')
import torch  # noqa: E402
from yolact_minimal_amd import hip  # noqa: E402
from tools.conv_sweep import make_desc  # noqa: E402

dev = torch.device('cuda:0')
ws = torch.empty(1 << 28, dtype=torch.uint8, device=dev)
counters = torch.zeros(hip.TILE_COUNTERS, dtype=torch.int32, device=dev)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def timeit(d, iters=30):
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


CASES = (  # spec, ksplit, tail, [(ring, grid)]
    ((8, 34, 34, 256, 1024, 1, 1, 1), 1, (16, 8), [(2, 768), (3, 512), (2, 1024)]),        # 2320 tiles = 3 x 768 + 16
    ((8, 34, 34, 256, 256, 3, 1, 0), 1, (68, 3), [(2, 512), (3, 512)]),                    # 580 tiles = 512 + 68
    ((8, 68, 68, 128, 512, 1, 1, 1), 1, (16, 4), [(2, 768)]),
)
for spec, ks, tail, variants in CASES:
    d, keep = make_desc(*spec, dev)
    M = spec[0] * d.Ho * d.Wo
    flops = 2.0 * M * spec[4] * d.k_pad
    d.tile_counters = counters.data_ptr()
    d.tile_m, d.tile_n, d.ksplit = 64, 64, ks
    d.tail_tiles, d.tail_ksplit = tail
    print(f'== M{M} N{spec[4]} K{d.k_pad}: MFMA time at 155 TFLOP/s {flops / 155e6:.1f} us')
    for ns, grid in variants:
        d.stages, d.grid_wgs = 40 + ns, grid
        row = []
        for abl in (0, 1, 2, 3, 5):
            os.environ['YM_PERS_ABL'] = str(abl)
            row.append(timeit(d))
        os.environ['YM_PERS_ABL'] = '0'
        print(f'   ring {ns}, grid {grid}: complete {row[0]:6.1f} us ({flops / row[0] / 1e6:5.1f} TF) | operand stream only {row[1]:6.1f} | '
              f'LDS reads + MFMAs {row[2]:6.1f} | MFMAs only {row[3]:6.1f} | MFMAs only, no per-tile barrier {row[4]:6.1f}', flush=True)
