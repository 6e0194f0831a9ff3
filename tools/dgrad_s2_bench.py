#!/usr/bin/env python3
"""The stride-2 data gradients of ResNet (layer2-4: the 3x3 conv2 and the 1x1 downsample of each stage's first block) with and
without the parity-class row order (YM_DGRAD_CLASSES, read once per process): time the tuned entry and a sweep of tiles / staging /
tail splits, print the winners (`--out` writes them as T_ entries).   python tools/dgrad_s2_bench.py [batch] [--out file.json]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from yolact_minimal_amd import hip  # noqa: E402
from tools import pers_bench as P  # noqa: E402

batch = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 8
out = sys.argv[sys.argv.index('--out') + 1] if '--out' in sys.argv else ''
# (h, w of the forward INPUT, cin, cout, k): conv2 3x3 s2 and downsample 1x1 s2 of layer2.0 / layer3.0 / layer4.0
specs = [(136, 136, 128, 128, 3), (136, 136, 256, 512, 1), (68, 68, 256, 256, 3), (68, 68, 512, 1024, 1), (34, 34, 512, 512, 3), (34, 34, 1024, 2048, 1)]
new, tot0, tot1 = {}, 0.0, 0.0
for h, w, cin, cout, k in specs:
    d, keep, sig = P.make_dgrad_desc(batch, h, w, cin, cout, k, 2)
    d.tile_counters = P.counters.data_ptr()
    hit = P.tuned.get(sig) or [0, 0, 0, 0, 0, 0, 0]

    def run(v):
        d.tile_m, d.tile_n, d.ksplit, d.kwaves, d.stages = v[0], v[1], v[2], v[3], v[4]
        d.tail_tiles, d.tail_ksplit = v[5], v[6]
        try:
            if hip.conv_workspace_bytes(d) > P.ws.numel():
                return None
        except RuntimeError:
            return None
        return P.time_desc(d, iters=20, reps=3)

    t0 = run(hit)
    best = (t0, hit)
    M = batch * h * w
    for tm, tn in ((128, 128), (128, 64), (64, 128), (64, 64)):
        tiles = -(-M // tm) * -(-cin // tn)
        cands = [(tm, tn, 1, 0, st, 0, 0) for st in (2, 22)]
        for r in sorted({tiles % 256, tiles % 512} - {0}):
            for ts in (2, 3, 4, 6):
                cands += [(tm, tn, 1, 0, st, r, ts) for st in (2, 22)]
        for v in cands:
            t = run(list(v))
            if t is not None and t < best[0] * 0.97:
                best = (t, list(v))
    flops = 2.0 * (M // 4) * cin * cout * k * k                  # real work: a quarter of the dx pixels per tap set
    print(f'{sig:36s} tuned {hit} {t0:7.1f} us -> {best[1]} {best[0]:7.1f} us  ({flops / best[0] / 1e6:.0f} TFLOP/s of real work)', flush=True)
    tot0 += t0
    tot1 += best[0]
    if best[1] != hit:
        new[sig] = best[1]
print(f'YM_DGRAD_CLASSES={os.environ.get("YM_DGRAD_CLASSES", "1")}: tuned {tot0:.0f} us -> swept {tot1:.0f} us over the six launches of a step')
if out:
    json.dump(new, open(out, 'w'), indent=0, sort_keys=True)
