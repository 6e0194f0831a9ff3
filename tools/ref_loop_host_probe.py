#!/usr/bin/env python3
"""Host-side time of every phase of the reference loop's step (no device fences inside the step): how long the host needs to ENQUEUE
forward / backward / optimizer, against the device time of the step.  Modes: --no-ddp, env YM_AUTO_SIDE_STREAM / YM_AUTO_FLAT."""
import argparse, os, socket, sys, time
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, 'dropin'), REPO]
import torch, torch.distributed as dist
import reference_loops as L
from yolact_minimal_amd.utils.synthetic import synth_targets
from yolact_minimal_amd.config import build_cfg
from modules.yolact import Yolact
ap = argparse.ArgumentParser(); ap.add_argument('--no-ddp', action='store_true'); ap.add_argument('--steps', type=int, default=8)
args = ap.parse_args()
with socket.socket() as sk:
    sk.bind(('127.0.0.1', 0)); port = sk.getsockname()[1]
dist.init_process_group(backend='nccl', init_method=f'tcp://127.0.0.1:{port}', rank=0, world_size=1)
dev = torch.device('cuda:0')
cfg = build_cfg('res101_coco', 'train', 544, train_bs=8, bs_per_gpu=8)
torch.manual_seed(0)
net = Yolact(cfg); net.train()
opt = L.make_optimizer(net, cfg)
net = net.cuda() if args.no_ddp else L.wrap_ddp(net, 0)
img = torch.randn(8, 3, 544, 544, device=dev)
boxes, masks = synth_targets(8, 544, seed=0)
boxes, masks = [b.to(dev) for b in boxes], [m.to(dev) for m in masks]
if os.environ.get('PROBE_EVENTS', '0') == '1':
    # no fences at all: device-side duration of every phase from events, step by step (does a phase inflate as the host runs ahead?)
    evs = []
    t_host = []
    for step in range(args.steps + 3):
        e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        h0 = time.perf_counter()
        e[0].record()
        lc, lb, lm, ls = net(img, [b.clone() for b in boxes], masks)
        e[1].record()
        tot = lc + lb + lm + ls
        opt.zero_grad()
        tot.backward()
        e[2].record()
        opt.step()
        e[3].record()
        evs.append(e)
        t_host.append(time.perf_counter() - h0)
        if step in (5, 10, 20, 40):
            ms = torch.cuda.memory_stats()
            print(f'alloc at step {step}: device_alloc {ms["num_device_alloc"]} device_free {ms["num_device_free"]} reserved {ms["reserved_bytes.all.current"] / 2**30:.2f} GiB active {ms["active_bytes.all.peak"] / 2**30:.2f} GiB peak')
    torch.cuda.synchronize()
    for i in (3, 6, 10, 15, 20, 30, args.steps + 2):
        if i < len(evs):
            e = evs[i]
            nxt = evs[i + 1][0] if i + 1 < len(evs) else None
            print(f'step {i}: device ms forward {e[0].elapsed_time(e[1]):.2f} backward {e[1].elapsed_time(e[2]):.2f} optimizer {e[2].elapsed_time(e[3]):.2f} '
                  f'to next step {e[3].elapsed_time(nxt) if nxt else 0:.2f} | host enqueue {t_host[i] * 1e3:.2f} ms')
    dist.destroy_process_group()
    sys.exit(0)
rows = []
for step in range(args.steps + 3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    lc, lb, lm, ls = net(img, [b.clone() for b in boxes], masks)
    t1 = time.perf_counter()
    tot = lc + lb + lm + ls
    opt.zero_grad()
    tot.backward()
    t2 = time.perf_counter()
    opt.step()
    t3 = time.perf_counter()
    torch.cuda.synchronize(); t4 = time.perf_counter()
    if step >= 3:
        rows.append((t1 - t0, t2 - t1, t3 - t2, t4 - t3, t4 - t0))
import numpy as np
m = np.array(rows).mean(0) * 1e3
print(f'host ms: forward {m[0]:.2f}  backward {m[1]:.2f}  optimizer {m[2]:.2f}  drain {m[3]:.2f}  | step (fenced) {m[4]:.2f}  alloc reserved {torch.cuda.memory_reserved() / 2**30:.1f} GiB, retries {torch.cuda.memory_stats()["num_alloc_retries"]}, segments {torch.cuda.memory_stats()["segment.all.allocated"]}')
dist.destroy_process_group()
