#!/usr/bin/env python3
import os, sys
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
dev = torch.device('cuda:0')
net, cfg = bench.build_net('swin_tiny_coco', 544, dev)
w = bench.Workload(net, cfg, 8, 544, dev, with_post=False)
t = min(bench.timed(w, 30, 5, lambda: None), bench.timed(w, 30, 0, lambda: None)) / 30
print(f'swin bs8 forward {t * 1e3:.3f} ms = {8 / t:.1f} img/s; digest {[round(float(o.double().sum()), 6) for o in w.engine.outputs()]}')
