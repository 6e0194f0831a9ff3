#!/usr/bin/env python3
"""How much of a Swin-T MLP fc1 launch is its exact-erf GELU epilogue?  Times the tuned launch of the four fc1 shapes of the bs=8 plan
with act = GELU and with act = ReLU (same kernel, same bytes)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from yolact_minimal_amd import hip  # noqa: E402
from tools.conv_sweep import make_desc  # noqa: E402
from tools import pers_bench as P  # noqa: E402

for (hw, c) in ((136, 96), (68, 192), (34, 384), (17, 768)):
    d, keep = make_desc(8, hw, hw, c, 4 * c, 1, 1, 0, P.dev)
    M = 8 * hw * hw
    sig = f'M{M}_N{4 * c}_C{c}_k1_s1_seg1_r0'
    hit = P.tuned.get(sig) or [0, 0, 0, 0, 0, 0, 0]
    d.tile_counters = P.counters.data_ptr()
    d.tile_m, d.tile_n, d.ksplit, d.kwaves, d.stages = hit[0], hit[1], hit[2], hit[3], hit[4]
    d.tail_tiles, d.tail_ksplit = (hit[5], hit[6]) if len(hit) > 6 else (0, 0)
    d.grid_wgs = hit[7] if len(hit) > 7 else 0
    d.scale = None
    row = []
    for name, act in (('relu', 1), ('gelu', 3), ('none', 0)):
        d.seg[0].act = act
        t = P.time_desc(d, iters=30, reps=4)
        row.append(f'{name} {t:7.1f} us')
    print(f'{sig:34s} {hit}  ' + ' | '.join(row), flush=True)
