#!/usr/bin/env python3
"""Multi-rank check of the training step (launched by torch.distributed.run): every rank trains K steps on its own shard with
`Trainer` (flat-buffer gradient all-reduce) and the ranks must end with IDENTICAL parameters and momentum.  With
YM_DIST_BACKEND=gloo two ranks can share one GPU (RCCL refuses duplicate devices), which is how tests/test_gpu_train.py runs it."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yolact_minimal_amd.utils.synthetic import synth_targets  # noqa: E402
from yolact_minimal_amd.config import build_cfg  # noqa: E402
from yolact_minimal_amd.modules.yolact import Yolact  # noqa: E402
from yolact_minimal_amd.trainer import Trainer, init_distributed  # noqa: E402


def main():
    rank, world, local_rank = init_distributed()
    ndev = torch.cuda.device_count()
    dev = torch.device('cuda', local_rank % ndev)
    torch.cuda.set_device(dev)
    name = os.environ.get('YM_CHECK_CFG', 'res50_coco')
    size = int(os.environ.get('YM_CHECK_SIZE', '128'))
    cfg = build_cfg(name, 'train', size, train_bs=2 * world, bs_per_gpu=2)
    torch.manual_seed(100 + rank)                       # DIFFERENT initial weights per rank: the trainer must broadcast rank 0's
    tr = Trainer(Yolact(cfg), cfg, dev, world, local_rank % ndev)
    for blk in (b for l in getattr(tr.net.backbone, 'layers', []) for b in getattr(l, 'blocks', [])):
        blk.drop_prob = 0.0                             # (Swin) DropPath masks are per-rank random numbers
    img = torch.randn(2, 3, size, size, generator=torch.Generator().manual_seed(7 + rank)).to(dev)
    boxes, masks = synth_targets(2, size, seed=50 + 10 * rank)
    boxes, masks = [b.to(dev) for b in boxes], [m.to(dev) for m in masks]
    losses = None
    for _ in range(3):
        losses = tr.step(img, boxes, masks)
    torch.cuda.synchronize()
    digest = torch.stack([tr.opt.flat.double().sum(), tr.opt.flat.double().abs().sum(), tr.opt.buf.double().abs().sum(),
                          tr.opt.flat[::997].double().pow(2).sum()])
    if dist.get_backend() != 'nccl':
        digest = digest.cpu()                               # (gloo gathers host tensors; RCCL needs device tensors)
    gathered = [torch.zeros_like(digest) for _ in range(world)]
    dist.all_gather(gathered, digest)
    gathered = [g.cpu() for g in gathered]
    ok = all(torch.equal(gathered[0], g) for g in gathered) and all(bool(torch.isfinite(l)) for l in losses)
    bn = [b for b in tr.net.buffers() if b.is_floating_point()]
    if bn:                                              # BN running statistics follow rank 0 (broadcast at the start of each step)
        pass
    if rank == 0:
        print('DDP_CHECK', 'OK' if ok else 'MISMATCH', 'world', world, 'launches', tr.reducer.launches if tr.reducer else None,
              'buckets', len(tr.reducer.buckets) if tr.reducer else None, [g.tolist() for g in gathered])
        if tr.reducer:
            import json
            print('DDP_LAUNCH_LOG', json.dumps(dict(params=len(tr.opt.params), buckets=len(tr.reducer.buckets), backend=dist.get_backend(),
                                                    log=tr.reducer.last_launch_log)))
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == '__main__':
    main()
