#!/usr/bin/env python3
"""Experiment: bs=8 inference as ONE engine of batch 8 vs TWO engines of batch 4 replayed concurrently on two streams (two
independent dependent-kernel chains in flight).  Tunes the batch-4 shapes first (in memory only)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_net  # noqa: E402
from yolact_minimal_amd.engine import InferEngine, tuned_table  # noqa: E402

dev = torch.device('cuda:0')
name = sys.argv[1] if len(sys.argv) > 1 else 'res101_coco'
net, cfg = build_net(name, 544, dev)
img8 = torch.randn(8, 3, 544, 544, device=dev)


def timed(fn, n=20, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


eng8 = net._engine(img8)
t8 = timed(lambda: eng8.run(img8))
print(f'one engine, batch 8: {t8 * 1e3:.3f} ms -> {8 / t8:.1f} img/s')

img4 = [img8[:4].contiguous(), img8[4:].contiguous()]
tmp = InferEngine(net, 4, 544, 544, dev)
res = tmp.autotune(5, verbose=False)
tuned_table().update({k: v[:7] for k, v in res.items()})
del tmp
engs = [InferEngine(net, 4, 544, 544, dev) for _ in range(2)]
t4 = timed(lambda: engs[0].run(img4[0]))
print(f'one engine, batch 4 (tuned): {t4 * 1e3:.3f} ms -> {4 / t4:.1f} img/s')
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
for e, s, im in zip(engs, streams, img4):            # capture each engine's graph on its own stream
    with torch.cuda.stream(s):
        e.run(im)
torch.cuda.synchronize()


def both():
    for e, s, im in zip(engs, streams, img4):
        with torch.cuda.stream(s):
            e.run(im)


t2 = timed(both)
print(f'two engines of batch 4 on two streams: {t2 * 1e3:.3f} ms -> {8 / t2:.1f} img/s')
