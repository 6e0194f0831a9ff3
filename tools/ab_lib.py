#!/usr/bin/env python3
"""A/B of two builds of libyolact_hip.so on the conv launches of a plan (run once per build with YM_LIB_PATH set; same shapes, same
configurations): the tuned entry of every 64x64-regime shape of tools/pers_bench.py plus, for `pers`, the persistent kernel with
rings of 2 / 3 at 3 workgroups per CU.      python tools/ab_lib.py bs1|bs8|train [pers]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from yolact_minimal_amd import hip  # noqa: E402
from tools.conv_sweep import make_desc  # noqa: E402
from tools import pers_bench as P  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else 'bs8'
pers = 'pers' in sys.argv[2:]
tot = {}
for spec in P.shapes(which):
    if which == 'train':
        d, keep, sig = P.make_dgrad_desc(*spec[:7])
        M, n = spec[0] * d.Ho * d.Wo, d.Cout
    else:
        d, keep = make_desc(*spec, P.dev)
        M, n = spec[0] * d.Ho * d.Wo, spec[4]
        sig = f'M{M}_N{spec[4]}_C{spec[3]}_k{spec[5]}_s{spec[6]}_seg1_r{spec[7]}'
    hit = P.tuned.get(sig) or [0, 0, 0, 0, 0, 0, 0]
    flops = 2.0 * M * n * d.k_pad
    d.tile_counters = P.counters.data_ptr()
    cfgs = [('tuned', hit)]
    if pers:
        cfgs += [('pers2', [64, 64, 1, 0, 42, 0, 0, 768]), ('pers3', [64, 64, 1, 0, 43, 0, 0, 768]), ('dl2', [64, 64, 1, 0, 22, 0, 0, 0]),
                 ('reg2', [64, 64, 1, 0, 2, 0, 0, 0])]
    row = []
    for name, h in cfgs:
        d.tile_m, d.tile_n, d.ksplit, d.kwaves, d.stages = h[0], h[1], h[2], h[3], h[4]
        d.tail_tiles, d.tail_ksplit = (h[5], h[6]) if len(h) > 6 else (0, 0)
        d.grid_wgs = h[7] if len(h) > 7 else 0
        t = P.time_desc(d, iters=30, reps=4)
        tot[name] = tot.get(name, 0.0) + (t or 0.0)
        row.append(f'{name} {t:7.1f} us {flops / t / 1e6:6.1f} TF' if t else f'{name} -')
    print(f'{sig:36s} {hit[:2]}/{hit[4]:2d}  ' + ' | '.join(row), flush=True)
print('sum:', {k: round(v, 1) for k, v in tot.items()}, 'lib', hip.LIB_PATH)
