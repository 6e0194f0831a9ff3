"""K-slice exchange footprint of one forward, from the plan (no counters needed).

A launch whose K range is split over workgroups writes every slice's partial tile to the split-K scratch (sc1 stores, through the
fabric) and the tile's last arriver reads the slices back: `ym_conv2d_workspace_bytes(desc)` is exactly that footprint, so
2 x the sum over the plan's launches bounds the exchange traffic of one forward.  Launches of the wave-DMA kernel split K over the
waves of ONE workgroup (LDS hand-off, no scratch) unless they carry a tail split.

    python tools/exchange_bytes.py [cfg] [batch] [table.json ...]       # default: res101_coco 1, the committed table

Prints one row per table: launches, launches with an exchange, scratch bytes per forward (written once, read once).
"""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')

import torch                                                                     # noqa: E402

import bench                                                                     # noqa: E402
from yolact_minimal_amd import engine as eng_mod, hip                            # noqa: E402


def footprint(net, batch, size, device, mode):
    e = eng_mod.InferEngine(net, batch, size, size, device, use_graph=False, mode=mode)
    n = ex = wave = 0
    total = 0
    for kind, arg in e.ops:
        if kind != 'conv':
            continue
        n += 1
        nb = hip.conv_workspace_bytes(arg.desc)
        wave += int(arg.desc.kwaves > 0 and arg.desc.stages in (22, 23, 24))
        if nb:
            ex += 1
            total += nb
    return dict(launches=n, with_exchange=ex, wave_dma=wave, scratch_mb=round(total / 1e6, 2), traffic_mb=round(2 * total / 1e6, 2),
                per_launch_mb=round(2 * total / 1e6 / max(n, 1), 3))


def main():
    cfg_name = sys.argv[1] if len(sys.argv) > 1 else 'res101_coco'
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    tables = sys.argv[3:] or [eng_mod.TUNED_PATH]
    dev = torch.device('cuda:0')
    net, cfg = bench.build_net(cfg_name, 544, dev)
    out = {}
    for path in tables:
        with open(path) as f:
            eng_mod._tuned = json.load(f)
        for mode in ('latency', 'throughput'):
            out[f'{os.path.basename(path)}:{mode}'] = footprint(net, batch, 544, dev, mode)
    print(json.dumps(dict(cfg=cfg_name, batch=batch, note='traffic_mb = 2 x split-K scratch of every launch of one forward',
                          tables=out), indent=1))


if __name__ == '__main__':
    main()
