#!/usr/bin/env python3
"""Experiment: bs=1 serving with S independent requests in flight — S engines (own activations, workspaces, counters, graphs)
replayed round-robin on S HIP streams.  A bs=1 forward is a chain of ~190 dependent launches that keeps ~35 % of the MFMA pipe
busy; a second chain fills the first one's launch boundaries, prologues and epilogues."""
import os
import sys
import time

os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
import torch  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_net  # noqa: E402
from yolact_minimal_amd.engine import InferEngine  # noqa: E402

dev = torch.device('cuda:0')
name = sys.argv[1] if len(sys.argv) > 1 else 'res101_coco'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
net, cfg = build_net(name, 544, dev)
img = torch.randn(B, 3, 544, 544, device=dev)
for S in (1, 2, 3, 4):
    engs = [InferEngine(net, B, 544, 544, dev) for _ in range(S)]
    streams = [torch.cuda.Stream() for _ in range(S)]
    for e, s in zip(engs, streams):
        with torch.cuda.stream(s):
            e.run(img)
    torch.cuda.synchronize()

    def step(i):
        with torch.cuda.stream(streams[i % S]):
            engs[i % S].run(img)
    for i in range(20):
        step(i)
    torch.cuda.synchronize()
    n = 200 if B == 1 else 40
    t0 = time.perf_counter()
    for i in range(n):
        step(i)
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / n
    print(f'{S} request(s) of batch {B} in flight: {t * 1e3:.3f} ms per batch -> {B / t:.1f} img/s', flush=True)
    del engs
