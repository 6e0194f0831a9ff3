#!/usr/bin/env python3
"""nms / after_nms timings on the bench's dense synthetic head outputs (bench.post_bench), alone: for rocprofv3 per-kernel tables."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

dev = torch.device('cuda:0')
net, cfg = bench.build_net('res101_coco', 544, dev)
print(json.dumps(bench.post_bench(net, cfg, dev, 544, iters=50)))
