"""One DDP training step of YOLACT on MI355X.

Reference step (`/root/reference/train.py:76,102-130`): DDP(net, broadcast_buffers=True); losses = net(images,
targets, masks); 4-float loss all-reduce (logging); zero_grad; backward with bucketed gradient all-reduce;
SGD(momentum 0.9, weight_decay 5e-4) step; LR warm-up / step decay (`train.py:103-109`).

Here: one process per GPU; `torch.distributed` with backend 'nccl' (= RCCL over xGMI on ROCm) carries the
gradient all-reduce, overlapped with the HIP backward kernels by DDP's bucket hooks (25 MB buckets: ~8 buckets for
res101's 200 MB of fp32 gradients, ring all-reduce is per-link bound on xGMI so few large messages are preferred
over many small ones); BN buffers are broadcast from rank 0 every step like the reference; the optimizer is ONE
launch of `ym_sgd_step` over a flat parameter buffer (parameters are re-pointed at views of it).
"""
import os

import torch
import torch.distributed as dist

from . import hip


def init_distributed(backend=None):
    """Join the process group from torchrun's env (RANK/LOCAL_RANK/WORLD_SIZE/MASTER_*). Returns (rank, world, local_rank)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    force = os.environ.get('YM_FORCE_DIST', '0') == '1'      # exercise the collective path with a single rank (tests)
    if (world > 1 or force) and not dist.is_initialized():
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        dist.init_process_group(backend=backend, init_method='env://')
    return rank, world, local_rank


def lr_at(cfg, step):
    """LR schedule of the reference loop (train.py:103-109): linear warm-up to `warmup_until`, x0.1 at each lr_step."""
    lr = cfg.lr
    if cfg.warmup_until > 0 and step <= cfg.warmup_until:
        return (cfg.lr - cfg.warmup_init) * (step / cfg.warmup_until) + cfg.warmup_init
    for i, s in enumerate(cfg.lr_steps):
        if step >= s:
            lr = cfg.lr * 0.1 ** i
    return lr


def shard_batch(global_batch, rank, world):
    """Contiguous shard of a global batch for this rank (DistributedSampler-style equal split, train.py:77)."""
    if global_batch % world != 0:
        raise AssertionError('Total training batch size must be divisible by GPU number.')   # config.py:234
    per = global_batch // world
    return range(rank * per, (rank + 1) * per)


def reduce_max(value, device=None):
    """MAX over ranks of a python float (bench timing)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device or 'cpu')
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


class FlatSGD:
    """torch.optim.SGD(momentum, weight_decay) semantics as one HIP launch over a flat buffer."""

    def __init__(self, params, lr, momentum=0.9, weight_decay=5e-4):
        self.params = [p for p in params if p.requires_grad]
        self.lr, self.momentum, self.weight_decay = lr, momentum, weight_decay
        n = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self.flat = torch.empty(n, device=dev, dtype=torch.float32)
        off = 0
        for p in self.params:                       # parameters become views of the flat buffer
            k = p.numel()
            self.flat[off:off + k].copy_(p.data.reshape(-1))
            p.data = self.flat[off:off + k].view_as(p.data)
            off += k
        self.buf = torch.zeros_like(self.flat)
        self.grad = torch.empty_like(self.flat)
        off = 0
        for p in self.params:                       # the HIP wgrad kernels write straight into these views
            k = p.numel()
            p._ym_grad_slot = self.grad[off:off + k].view_as(p.data)
            p._ym_slot_free = True
            off += k
        self.steps = 0

    def zero_grad(self):
        for p in self.params:
            p.grad = None
            p._ym_slot_free = True

    def step(self):
        for p in self.params:                       # gather only gradients that did not land in their slot
            g = p.grad
            if g is None:
                p._ym_grad_slot.zero_()
            elif g.data_ptr() != p._ym_grad_slot.data_ptr():
                p._ym_grad_slot.copy_(g)
        hip.check(hip.lib().ym_sgd_step(hip.ptr(self.flat), hip.ptr(self.grad), hip.ptr(self.buf), self.flat.numel(),
                                        float(self.lr), float(self.momentum), float(self.weight_decay),
                                        int(self.steps == 0), hip.stream_ptr()), 'ym_sgd_step')
        self.steps += 1


class Trainer:
    def __init__(self, net, cfg, device, world=1, local_rank=0):
        self.net, self.cfg, self.device, self.world = net.train().to(device), cfg, device, world
        self.opt = FlatSGD(self.net.parameters(), cfg.lr)
        self.model = self.net
        self.ddp = world > 1 or (dist.is_initialized() and os.environ.get('YM_FORCE_DIST', '0') == '1')
        if self.ddp:
            from torch.nn.parallel import DistributedDataParallel as DDP
            self.model = DDP(self.net, device_ids=[local_rank], output_device=local_rank, broadcast_buffers=True,
                             bucket_cap_mb=25, gradient_as_bucket_view=False)
        self.step_idx = 0

    def step(self, images, targets, masks):
        self.opt.lr = lr_at(self.cfg, self.step_idx)
        losses = self.model(images, targets, masks)
        if self.ddp:
            all_loss = torch.stack([l.detach() for l in losses])
            dist.all_reduce(all_loss)                 # 16-byte logging collective, train.py:121-122
        total = losses[0] + losses[1] + losses[2] + losses[3]
        self.opt.zero_grad()
        total.backward()
        self.opt.step()
        self.net.mark_weights_changed()
        self.step_idx += 1
        return losses
