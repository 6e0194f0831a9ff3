#!/usr/bin/env python3
"""The reference's own training loop (dropin/reference_loops.py: torch DDP + torch.optim.SGD around the HIP autograd path) for K steps:
stage times as train.py prints them (t_fl / t_b / t_u), and the step time.  For `rocprofv3 --kernel-trace -- python tools/ref_loop_profile.py`."""
import argparse
import os
import socket
import sys
import time

os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, 'dropin'), REPO]
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
import reference_loops as L  # noqa: E402
from yolact_minimal_amd.utils.synthetic import synth_targets  # noqa: E402
from yolact_minimal_amd.config import build_cfg  # noqa: E402
from modules.yolact import Yolact  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--cfg', default='res101_coco')
ap.add_argument('--batch', type=int, default=8)
ap.add_argument('--steps', type=int, default=8)
ap.add_argument('--no-ddp', action='store_true')
args = ap.parse_args()
with socket.socket() as sk:
    sk.bind(('127.0.0.1', 0))
    port = sk.getsockname()[1]
dist.init_process_group(backend='nccl', init_method=f'tcp://127.0.0.1:{port}', rank=0, world_size=1)
dev = torch.device('cuda:0')
cfg = build_cfg(args.cfg, 'train', 544, train_bs=args.batch, bs_per_gpu=args.batch)
torch.manual_seed(0)
net = Yolact(cfg)
net.train()
optimizer = L.make_optimizer(net, cfg)
net = net.cuda() if args.no_ddp else L.wrap_ddp(net, 0)
if args.no_ddp:
    cfg.cuda = True
img = torch.randn(args.batch, 3, 544, 544, device=dev)
boxes, masks = synth_targets(args.batch, 544, seed=0)
boxes, masks = [b.to(dev) for b in boxes], [m.to(dev) for m in masks]
loader = lambda n: ((img, [b.clone() for b in boxes], masks) for _ in range(n))      # noqa: E731
nop = lambda *a: None                                                              # noqa: E731
fences = os.environ.get('REF_LOOP_FENCES', '1') != '0'
marks = []
mark = lambda step, losses, lr: marks.append(time.perf_counter())                   # noqa: E731
L.train_loop(net, optimizer, cfg, loader(args.steps + 3), max_steps=args.steps + 3, on_step=mark, fences=fences)
torch.cuda.synchronize()
dt = (time.perf_counter() - marks[2]) / args.steps                                  # the steps after the 3rd (timer started at step 1)
t_fl, t_b, t_u = L.timer.get_times(['for+loss', 'backward', 'update'])
print(f'{dt * 1e3:.2f} ms/step over {args.steps} steps ({"fenced like train.py" if fences else "no fences"}): for+loss {t_fl * 1e3:.2f}  backward {t_b * 1e3:.2f}  '
      f'update {t_u * 1e3:.2f} ms')
dist.destroy_process_group()
