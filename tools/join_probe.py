#!/usr/bin/env python3
"""Is every gradient complete (with respect to the caller's stream) when `loss.backward()` returns in the reference loop?  Right after
backward() a main-stream clone of the LAST gradients the side stream writes (the stem's) is taken, many steps deep into an unfenced
run; after a device synchronize the clones must equal the gradients."""
import os, socket, sys
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, 'dropin'), REPO]
import torch, torch.distributed as dist
import reference_loops as L
from yolact_minimal_amd.utils.synthetic import synth_targets
from yolact_minimal_amd.config import build_cfg
from modules.yolact import Yolact
with socket.socket() as sk:
    sk.bind(('127.0.0.1', 0)); port = sk.getsockname()[1]
dist.init_process_group(backend='nccl', init_method=f'tcp://127.0.0.1:{port}', rank=0, world_size=1)
dev = torch.device('cuda:0')
cfg = build_cfg('res101_coco', 'train', 544, train_bs=8, bs_per_gpu=8)
torch.manual_seed(0)
net = Yolact(cfg); net.train()
opt = L.make_optimizer(net, cfg)
net = L.wrap_ddp(net, 0)
img = torch.randn(8, 3, 544, 544, device=dev)
boxes, masks = synth_targets(8, 544, seed=0)
boxes, masks = [b.to(dev) for b in boxes], [m.to(dev) for m in masks]
watch = [p for n, p in net.module.named_parameters() if n in ('backbone.conv1.weight', 'backbone.layers.0.0.conv1.weight', 'backbone.layers.0.0.downsample.0.weight', 'backbone.layers.1.0.conv2.weight')]
snaps = []
for step in range(30):
    lc, lb, lm, ls = net(img, [b.clone() for b in boxes], masks)
    tot = lc + lb + lm + ls
    opt.zero_grad()
    tot.backward()
    clones = [p.grad.clone() for p in watch]          # main stream, immediately
    later = [p.grad for p in watch]
    if step in (5, 15, 25, 29):
        torch.cuda.synchronize()
        snaps.append((step, [bool(torch.equal(a, b)) for a, b in zip(clones, later)], [float(b.abs().max()) for b in later]))
    # no optimizer step: the weights stay fixed, so every step must ALSO reproduce the first step's gradient
    if step == 0:
        torch.cuda.synchronize(); first = [p.grad.clone() for p in watch]
    elif step in (5, 15, 25, 29):
        snaps[-1] += ([bool(torch.equal(a, b)) for a, b in zip(first, later)],)
for s in snaps:
    print('JOIN_PROBE', s)
dist.destroy_process_group()
