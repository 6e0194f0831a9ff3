#!/usr/bin/env python3
"""Upper bound for an epilogue that stores straight from the MFMA accumulator layout (no staging through LDS, no barrier): the
trace build's YM_PERS_ABL=9 skips the staging and stores accumulator registers as they are (wrong values, same stores).  Run once
without and once with the variable, both with YM_LIB_PATH=tools/trace/libyolact_hip_trace.so:
    python tools/epi_ablation.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from yolact_minimal_amd import hip  # noqa: E402
from tools.conv_sweep import make_desc  # noqa: E402
from tools import pers_bench as P  # noqa: E402

# short-K launches (Swin-T stage 1 / 2, ResNet layer1) and two long-K ones for contrast: (hw, cin, cout, k, residual)
SHAPES = ((136, 96, 384, 1, 0), (136, 96, 288, 1, 0), (136, 384, 96, 1, 1), (136, 64, 256, 1, 1), (136, 256, 64, 1, 0), (136, 64, 64, 3, 0),
          (68, 192, 768, 1, 0), (34, 1024, 256, 1, 0), (34, 256, 256, 3, 0))
tot = 0.0
for hw, cin, cout, k, res in SHAPES:
    d, keep = make_desc(8, hw, hw, cin, cout, k, 1, res, P.dev)
    M = 8 * hw * hw
    d.tile_counters = P.counters.data_ptr()
    d.tile_m, d.tile_n, d.ksplit, d.kwaves, d.stages = 64, 64, 1, 0, 22      # the per-item direct-to-LDS kernel, no K split
    t = P.time_desc(d, iters=30, reps=4)
    tot += t
    print(f'M{M}_N{cout}_C{cin}_k{k}_r{res}: {t:7.1f} us', flush=True)
print(f'sum {tot:.1f} us   YM_PERS_ABL={os.environ.get("YM_PERS_ABL", "")}')
