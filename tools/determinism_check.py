"""Per-layer check of a tuned inference plan at a benchmarked shape: (1) is every conv launch deterministic (same input, repeated
launches, bit-identical output)?  (2) does the tuned variant (tile / split-K / tail / staging from tuned_gfx950.json) agree with
the plain configuration of the same kernel (128x128 or 64x64 tile, no split, register staging) on the same input?

    python tools/determinism_check.py [cfg] [batch] [reps]      (needs an MI355X)
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from yolact_minimal_amd import hip
    from yolact_minimal_amd.config import build_cfg
    from yolact_minimal_amd.modules.yolact import Yolact
    name = sys.argv[1] if len(sys.argv) > 1 else 'res101_coco'
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 6
    dev = torch.device('cuda:0')
    os.environ['YM_GRAPH'] = '0'
    cfg = build_cfg(name, 'val', 544)
    torch.manual_seed(0)
    net = Yolact(cfg).eval().to(dev)
    img = torch.randn(batch, 3, 544, 544, generator=torch.Generator().manual_seed(1)).to(dev)
    with torch.no_grad():
        net(img)
    eng = net._engine(img)
    torch.cuda.synchronize()
    bad = 0
    ws = eng.workspace
    for i, (kind, arg) in enumerate(eng.ops):
        if kind != 'conv':
            eng._launch_one(kind, arg, ws)
            continue
        d = arg.desc
        outs = []
        for s in range(d.nseg):
            outs.append(d.seg[s].out)
        # the launch writes into plan-owned buffers: snapshot them through the torch tensors that own the memory
        owners = [t for t in eng._bufs if any(t.data_ptr() <= o < t.data_ptr() + t.numel() * 4 for o in outs)]
        snaps = []
        for r in range(reps):
            hip.conv2d_fwd(d, ws)
            torch.cuda.synchronize()
            snaps.append([t.clone() for t in owners])
        nondet = any(not torch.equal(a, b) for s in snaps[1:] for a, b in zip(snaps[0], s))
        # the same conv in its plain configuration
        keep = (d.tile_m, d.tile_n, d.ksplit, d.kwaves, d.stages, d.tail_tiles, d.tail_ksplit)
        d.tile_m, d.tile_n, d.ksplit, d.kwaves, d.stages, d.tail_tiles, d.tail_ksplit = 0, 0, 1, 0, 0, 0, 0
        big = torch.empty(max(hip.conv_workspace_bytes(d), 256), device=dev, dtype=torch.uint8)
        hip.conv2d_fwd(d, big)
        torch.cuda.synchronize()
        plain = [t.clone() for t in owners]
        d.tile_m, d.tile_n, d.ksplit, d.kwaves, d.stages, d.tail_tiles, d.tail_ksplit = keep
        hip.conv2d_fwd(d, ws)
        torch.cuda.synchronize()
        err = max(float((a - b).abs().max()) for a, b in zip(owners, plain))
        scale = max(float(b.abs().max()) for b in plain) + 1e-30
        flag = 'NONDETERMINISTIC' if nondet else ('MISMATCH' if err > 2e-5 * scale else '')
        if flag:
            bad += 1
            nd = 0
            if nondet:
                nd = max(int((a != b).sum()) for s in snaps[1:] for a, b in zip(snaps[0], s))
            print(f'{flag:17s} op {i:3d} {arg.name:45s} sig {arg.sig} cfg {keep} max|tuned-plain| {err:.3e} (scale {scale:.3e}) differing elems {nd}', flush=True)
    print(f'{name} bs={batch}: {sum(1 for k, _ in eng.ops if k == "conv")} convs checked, {bad} flagged')


if __name__ == '__main__':
    main()
