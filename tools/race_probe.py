"""Stress one conv configuration and localise run-to-run differences: which output tiles differ, are they tail tiles, how large.
    python tools/race_probe.py   (needs an MI355X)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from yolact_minimal_amd import hip
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(3)
    b, h, w, cin, cout = 8, 34, 34, 384, 1152
    x = torch.randn(b, h, w, cin, generator=g).to(dev)
    wt = (torch.randn(cout, cin, 1, 1, generator=g) * 0.05).to(dev)
    wp = hip.pack_conv_weight(wt, cin, cin)
    out = torch.empty(b, h, w, cout, device=dev)
    counters = torch.zeros(hip.TILE_COUNTERS, device=dev, dtype=torch.int32)
    ws = torch.empty(1 << 28, dtype=torch.uint8, device=dev)
    M = b * h * w
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
    for tile, ksplit, stages, tail in [((64, 64), 1, 22, (50, 2)), ((64, 64), 1, 0, (50, 2)), ((64, 64), 1, 23, (50, 2)),
                                       ((64, 64), 2, 22, (0, 0)), ((64, 64), 1, 22, (0, 0)), ((64, 64), 1, 22, (50, 3)),
                                       ((64, 64), 1, 22, (200, 2))]:
        d = hip.ConvDesc()
        d.inp, d.weight = x.data_ptr(), wp.data_ptr()
        d.B, d.H, d.W, d.Cin, d.Cout, d.KH, d.KW = b, h, w, cin, cout, 1, 1
        d.stride, d.pad, d.Ho, d.Wo, d.k_pad, d.nseg = 1, 0, h, w, cin, 1
        d.seg[0].n_begin, d.seg[0].n_end, d.seg[0].out = 0, cout, out.data_ptr()
        d.seg[0].batch_stride, d.seg[0].pitch, d.seg[0].act = h * w * cout, cout, 0
        d.tile_counters = counters.data_ptr()
        d.tile_m, d.tile_n, d.ksplit, d.stages = tile[0], tile[1], ksplit, stages
        d.tail_tiles, d.tail_ksplit = tail
        ws.zero_()
        hip.conv2d_fwd(d, ws)
        first = out.clone()
        tiles_m, tiles_n = -(-M // tile[0]), -(-cout // tile[1])
        main_tiles = tiles_m * tiles_n - tail[0]
        nbad, events = 0, []
        for it in range(reps):
            hip.conv2d_fwd(d, ws)
            diff = out != first
            if bool(diff.any()):
                nbad += 1
                if len(events) < 6:
                    idx = diff.reshape(M, cout).nonzero()
                    tm, tn = idx[:, 0] // tile[0], idx[:, 1] // tile[1]
                    ids = torch.unique(tm * tiles_n + tn).tolist()          # n-fastest tile order (m_fastest = 0 with a tail)
                    mx = float((out - first).abs().max())
                    rows = torch.unique(idx[:, 0] % tile[0]).tolist()
                    events.append(dict(it=it, n=int(diff.sum()), tiles=ids[:6], tail=[i >= main_tiles for i in ids[:6]], maxdiff=mx,
                                       rows_in_tile=rows[:10], cols=torch.unique(idx[:, 1] % tile[1]).tolist()[:10]))
        print(f'tile {tile} ksplit {ksplit} stages {stages} tail {tail}: {nbad}/{reps} launches differ', flush=True)
        for e in events:
            print('    ', e, flush=True)


if __name__ == '__main__':
    main()
