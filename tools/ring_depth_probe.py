"""Do deeper DMA rings pay where they cost no occupancy?  (tools/tune_forward.py only ever tried rings of 2.)

Rows of the batch-1 plan whose wave-DMA launch has at most one workgroup per CU, or one- / two-wave workgroups, get a ring of 3 / 4
(stages 23 / 24); prints the hipGraph forward time of each variant and whether the outputs are bit-identical to the committed table's
(the ring depth does not change the MFMA order).    python tools/ring_depth_probe.py
"""
import copy
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')

import torch                                                                     # noqa: E402

import bench                                                                     # noqa: E402
from yolact_minimal_amd import engine as E                                       # noqa: E402


def fwd_ms(eng, img, iters=100):
    for _ in range(10):
        eng.run(img)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(iters):
            eng.run(img)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters)
    return best


def variant(base, ns):
    t = copy.deepcopy(base)
    changed = []
    for k, v in base.items():
        if k.endswith('_tp') or not k.startswith('M') or len(v) < 7 or v[3] == 0 or v[4] != 22 or v[5]:
            continue
        M, N = int(k.split('_')[0][1:]), int(k.split('_')[1][1:])
        wgs = -(-M // v[0]) * -(-N // v[1])
        wpb = v[7] if len(v) > 7 and v[7] else 4
        lds = wpb * ns * (v[0] + v[1]) * 128
        per_cu = -(-(wgs * v[3] // wpb if wpb >= v[3] else wgs) // 256)
        if ns == 4 and (v[0], v[1]) != (32, 32):
            continue
        if per_cu * lds <= 160 * 1024:
            t[k] = v[:4] + [20 + ns] + v[5:]
            changed.append(k)
    return t, changed


def main():
    dev = torch.device('cuda:0')
    net, cfg = bench.build_net('res101_coco', 544, dev)
    img = torch.randn(1, 3, 544, 544, device=dev)
    base = json.load(open(E.TUNED_PATH))
    E._tuned = base
    e0 = E.InferEngine(net, 1, 544, 544, dev)
    t0 = fwd_ms(e0, img)
    ref = [o.clone() for o in e0.outputs()]
    print(f'committed table: {t0:.4f} ms per forward')
    for ns in (3, 4):
        tab, changed = variant(base, ns)
        E._tuned = tab
        e = E.InferEngine(net, 1, 544, 544, dev)
        t = fwd_ms(e, img)
        same = all(torch.equal(a, b) for a, b in zip(ref, e.outputs()))
        print(f'ring of {ns} on {len(changed)} shapes: {t:.4f} ms per forward, outputs bit-identical: {same}\n   {changed}')
    E._tuned = base


if __name__ == '__main__':
    main()
