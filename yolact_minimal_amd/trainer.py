"""One DDP training step of YOLACT on MI355X.

Reference step (`/root/reference/train.py:76,102-130`): DDP(net, broadcast_buffers=True); losses = net(images,
targets, masks); 4-float loss all-reduce (logging); zero_grad; backward with bucketed gradient all-reduce;
SGD(momentum 0.9, weight_decay 5e-4) step; LR warm-up / step decay (`train.py:103-109`).

Here: one process per GPU; `torch.distributed` with backend 'nccl' (= RCCL over xGMI on ROCm) carries the
gradient all-reduce.  Parameters, gradients and momentum live in three flat fp32 buffers: the HIP wgrad kernels write
each gradient into its slice, `FlatGradReducer` all-reduces contiguous >=25 MB ranges of that buffer as backward fills
them (asynchronously on the RCCL stream, zero copies; ~8 messages for res101's 200 MB — ring all-reduce is per-link
bound on xGMI so few large messages are preferred), BN running statistics are broadcast from rank 0 every step as ONE
flat message like `DDP(broadcast_buffers=True)`, and the optimizer is ONE launch of `ym_sgd_step` over the flat buffers.
"""
import os

import torch
import torch.distributed as dist

from . import hip
from .train_engine import wgrad_stream_if_used, weights_changed, release_wgrad_scratch


def init_distributed(backend=None):
    """Join the process group from torchrun's env (RANK/LOCAL_RANK/WORLD_SIZE/MASTER_*). Returns (rank, world, local_rank)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    force = os.environ.get('YM_FORCE_DIST', '0') == '1'      # exercise the collective path with a single rank (tests)
    if (world > 1 or force) and not dist.is_initialized():
        if backend is None:
            backend = os.environ.get('YM_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
        dist.init_process_group(backend=backend, init_method='env://')
    return rank, world, local_rank


def lr_at(cfg, step):
    """Learning rate the reference loop has in force at `step` (train.py:103-109), stateless.  The reference keeps the rate in the
    optimizer and overwrites it on two events: every step <= warmup_until sets the linear warm-up value, and then — in the same
    iteration, so it wins — a step listed in cfg.lr_steps sets cfg.lr * 0.1**index.  Consequences kept here: step 0 trains at the
    FULL cfg.lr (0 is in lr_steps), an lr_step inside the warm-up window applies for that one step, and after the warm-up the
    rate is whatever the most recent event left."""
    steps = list(cfg.lr_steps)
    if step in steps:
        return cfg.lr * 0.1 ** steps.index(step)
    warm = cfg.warmup_until > 0
    if warm and step <= cfg.warmup_until:
        return (cfg.lr - cfg.warmup_init) * (step / cfg.warmup_until) + cfg.warmup_init
    past = [s for s in steps if s <= step]
    last = max(past) if past else None
    if warm and (last is None or last < cfg.warmup_until):
        return (cfg.lr - cfg.warmup_init) * 1.0 + cfg.warmup_init          # the value the last warm-up step (warmup_until) set
    return cfg.lr * 0.1 ** steps.index(last) if last is not None else cfg.lr


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
        # every parameter starts on a 64-byte boundary of the flat buffer (the conv epilogue wants 16-byte aligned bias /
        # scale vectors, and a misaligned one silently costs the vectorised epilogue); the gaps stay zero for ever
        starts, n = [], 0
        for p in self.params:
            starts.append(n)
            n += (p.numel() + 15) // 16 * 16
        dev = self.params[0].device
        self.flat = torch.zeros(n, device=dev, dtype=torch.float32)
        self.offsets = []
        for p, off in zip(self.params, starts):     # parameters become views of the flat buffer
            k = p.numel()
            self.flat[off:off + k].copy_(p.data.reshape(-1))
            p.data = self.flat[off:off + k].view_as(p.data)
            self.offsets.append((off, off + k))
        self.buf = torch.zeros_like(self.flat)
        self.grad = torch.zeros_like(self.flat)
        for p, (a, b) in zip(self.params, self.offsets):   # the HIP wgrad kernels write straight into these views
            p._ym_grad_slot = self.grad[a:b].view_as(p.data)
            p._ym_slot_free = True
            p._ym_in_slot = False
        self.steps = 0

    def zero_grad(self):
        for p in self.params:
            p.grad = None
            p._ym_slot_free = True
            p._ym_in_slot = False
            p._ym_side_written = False

    @staticmethod
    def gather(p):
        """Make sure this parameter's gradient sits in its slot of the flat buffer (no-op when autograd adopted the slot)."""
        if p._ym_in_slot:
            return
        g = p.grad
        if g is None:
            p._ym_grad_slot.zero_()
        elif g.data_ptr() != p._ym_grad_slot.data_ptr():
            if getattr(p, '_ym_side_written', False):
                # the slot was written from the side stream on the promise that autograd ADOPTS it unread; a sum / clone made on the
                # main stream raced with that write (train_engine._grad_slot joins the streams for every case it knows about)
                raise RuntimeError('a gradient written on the side stream was copied or summed by autograd on the main stream '
                                   '(AccumulateGrad did not adopt the slot view): run with YM_WGRAD_STREAM=0')
            p._ym_grad_slot.copy_(g)
        p._ym_in_slot = True

    def check_storage(self):
        """The parameters must still BE the flat buffer: `net.to(device)`, `.half()`, `.double()` after the trainer was built
        re-allocate parameter storage (Module._apply), after which this optimizer would step an orphaned buffer and the wgrad
        kernels would write stale slots — fail loudly instead."""
        base = self.flat.data_ptr()
        for p, (a, _) in zip(self.params, self.offsets):
            if p.data_ptr() != base + 4 * a:
                raise RuntimeError('a parameter no longer aliases the optimizer\'s flat buffer (was the module moved / cast after the '
                                   'Trainer was created?): build the Trainer after .to(device) and do not re-allocate parameters')

    def step(self):
        """(Deviation from torch.optim.SGD, irrelevant for YOLACT where every parameter receives a gradient each step: a parameter
        whose grad is None is treated as a zero gradient — it still gets weight decay and momentum — where torch skips it.)"""
        self.check_storage()
        for p in self.params:
            self.gather(p)
        hip.check(hip.lib().ym_sgd_step(hip.ptr(self.flat), hip.ptr(self.grad), hip.ptr(self.buf), self.flat.numel(),
                                        float(self.lr), float(self.momentum), float(self.weight_decay),
                                        int(self.steps == 0), hip.stream_ptr()), 'ym_sgd_step')
        self.steps += 1
        weights_changed()                           # the packed weight images of train_engine are stale now
        for p in self.params:
            p._ym_in_slot = False                   # the next step() gathers again unless a hook already did


class FlatAdamW(FlatSGD):
    """torch.optim.AdamW(lr, betas=(0.9, 0.999), eps=1e-8, weight_decay) as one HIP launch over the flat buffers — what the
    reference picks for swin_tiny_coco (train.py:62-63: weight_decay=0.05).  Shares FlatSGD's flat parameter / gradient layout."""

    def __init__(self, params, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.05):
        super().__init__(params, lr, momentum=0.0, weight_decay=weight_decay)
        self.betas, self.eps = betas, eps
        self.exp_avg_sq = torch.zeros_like(self.flat)          # `buf` is exp_avg

    def step(self):
        self.check_storage()
        for p in self.params:
            self.gather(p)
        self.steps += 1
        hip.check(hip.lib().ym_adamw_step(hip.ptr(self.flat), hip.ptr(self.grad), hip.ptr(self.buf), hip.ptr(self.exp_avg_sq),
                                          self.flat.numel(), float(self.lr), float(self.betas[0]), float(self.betas[1]),
                                          float(self.eps), float(self.weight_decay), int(self.steps), hip.stream_ptr()),
                  'ym_adamw_step')
        weights_changed()
        for p in self.params:
            p._ym_in_slot = False


_REDUCE_AT_END = os.environ.get('YM_REDUCE_AT_END', '0') == '1'      # diagnostic: every bucket is all-reduced after backward (no overlap)
_REDUCE_DRY = os.environ.get('YM_REDUCE_DRY', '0') == '1'            # diagnostic: the hooks and the bookkeeping run, the collectives do not


class FlatGradReducer:
    """Bucketed gradient all-reduce overlapped with backward, zero-copy on the optimizer's flat gradient buffer.

    Takes the place of DDP's reducer (reference `train.py:76`): the HIP wgrad kernels already write every gradient into its
    slice of one flat buffer, so a bucket is just a contiguous range of it — no per-parameter copies into and out of bucket
    storage (632 copy launches per res101 step with torch DDP).  Buckets are cut in REVERSE parameter order (the order
    backward produces gradients), >= `bucket_bytes` each (25 MB: ~8 messages for res101's 200 MB; ring all-reduce over xGMI
    is per-link bound, so few large messages beat many small ones), and are launched in bucket order on every rank as soon as
    their last gradient has landed (`register_post_accumulate_grad_hook`), asynchronously on the RCCL stream.
    """

    def __init__(self, opt, world, bucket_bytes=None, group=None):
        self.opt, self.world, self.group = opt, world, group
        if bucket_bytes is None:
            bucket_bytes = int(float(os.environ.get('YM_BUCKET_MB', '25')) * (1 << 20))
        self.buckets = []                            # [start, end, param indices]
        cur, size, end = [], 0, opt.flat.numel()
        for i in reversed(range(len(opt.params))):
            cur.append(i)
            size += opt.params[i].numel() * 4
            if size >= bucket_bytes or i == 0:
                self.buckets.append((opt.offsets[i][0], end, cur))
                cur, size, end = [], 0, opt.offsets[i][0]
        self.bucket_of = {}
        for b, (_, _, idx) in enumerate(self.buckets):
            for i in idx:
                self.bucket_of[i] = b
        backend = dist.get_backend(group) if dist.is_initialized() else 'none'
        self.avg_op = backend == 'nccl'                # RCCL averages in the collective; gloo has no AVG -> SUM then scale
        self._hooks = [p.register_post_accumulate_grad_hook(self._make_hook(i)) for i, p in enumerate(opt.params)]
        self.launches = 0
        self.enabled = True        # False: this backward keeps its gradients local (DDP.no_sync(); see train_state.ModuleTrainState)
        self.reset()

    def reset(self):
        self.pending = [len(idx) for _, _, idx in self.buckets]
        self.next_bucket = 0
        self.works = []
        self.grads_seen = 0
        self.in_finish = False
        self.launch_log = []        # per step: (bucket, gradients produced when it was launched, launched by finish()?)

    def _make_hook(self, i):
        def hook(p):
            if not self.enabled:
                return
            FlatSGD.gather(p)
            if p.grad is not None and p.grad.data_ptr() != p._ym_grad_slot.data_ptr():
                p.grad = p._ym_grad_slot.view_as(p._ym_grad_slot)     # the reduced value is what the caller must see
            b = self.bucket_of[i]
            self.pending[b] -= 1
            self.grads_seen += 1
            if not _REDUCE_AT_END:
                self._launch_ready()
        return hook

    def _launch_ready(self):
        while self.next_bucket < len(self.buckets) and self.pending[self.next_bucket] == 0:
            a, e, _ = self.buckets[self.next_bucket]
            op = dist.ReduceOp.AVG if self.avg_op else dist.ReduceOp.SUM
            # the collective orders itself after the CURRENT stream: weight gradients are written on train_engine's side stream,
            # everything else (BN / bias gradients) on the main one -> issue it from the side stream after making that wait for
            # the main stream; the main stream itself is not held up
            flat = self.opt.grad
            if _REDUCE_DRY:
                self.launch_log.append((self.next_bucket, self.grads_seen, self.in_finish))
                self.next_bucket += 1
                continue
            side = wgrad_stream_if_used(flat.device) if flat.is_cuda else None
            if side is not None:
                side.wait_stream(torch.cuda.current_stream(flat.device))
                with torch.cuda.stream(side):
                    work = dist.all_reduce(flat[a:e], op=op, group=self.group, async_op=True)
            else:
                work = dist.all_reduce(flat[a:e], op=op, group=self.group, async_op=True)
            self.works.append(work)
            self.launch_log.append((self.next_bucket, self.grads_seen, self.in_finish))
            self.launches += 1
            self.next_bucket += 1

    def finish(self):
        """After backward: reduce what is left (parameters that received no gradient count as zeros), wait for every
        bucket, and leave the AVERAGED gradients in the flat buffer."""
        if not self.enabled:
            self.reset()
            return
        self.in_finish = True
        for b in range(self.next_bucket, len(self.buckets)):
            for i in self.buckets[b][2]:
                p = self.opt.params[i]
                if not p._ym_in_slot:
                    FlatSGD.gather(p)
            self.pending[b] = 0
        self._launch_ready()
        for w in self.works:
            w.wait()
        if not self.avg_op and self.world > 1:
            self.opt.grad.mul_(1.0 / self.world)
        self.last_launch_log = self.launch_log
        self.reset()


def flatten_batch_counters(module):
    """All BatchNorm `num_batches_tracked` counters as views of one int64 tensor: one `+= 1` launch per step instead of one
    per layer (the modules are flagged so the forward skips its own increment)."""
    bns = [m for m in module.modules() if isinstance(m, torch.nn.modules.batchnorm._BatchNorm) and m.num_batches_tracked is not None]
    if not bns:
        return None
    flat = torch.stack([m.num_batches_tracked.reshape(()) for m in bns]).contiguous()
    for i, m in enumerate(bns):
        m.num_batches_tracked.data = flat[i]
        m._ym_nbt_flat = True
    return flat


def flatten_buffers(module):
    """Re-point every floating-point buffer (BN running statistics) at a view of one flat tensor so that the per-step
    rank-0 broadcast (`DDP(broadcast_buffers=True)`, reference train.py:76) is ONE message instead of 2 per BN layer."""
    bufs = [b for b in module.buffers() if b.is_floating_point()]
    if not bufs:
        return None
    flat = torch.empty(sum(b.numel() for b in bufs), device=bufs[0].device, dtype=bufs[0].dtype)
    off = 0
    for b in bufs:
        k = b.numel()
        flat[off:off + k].copy_(b.reshape(-1))
        b.data = flat[off:off + k].view_as(b)
        off += k
    return flat


class Trainer:
    def __init__(self, net, cfg, device, world=1, local_rank=0):
        self.net, self.cfg, self.device, self.world = net.train().to(device), cfg, device, world
        if getattr(self.net, '_train_state', None) is not None:
            self.net._drop_train_state()             # the module's own plumbing (train_state.py) gives way to this trainer's
        release_wgrad_scratch()                      # (per-layer scratch of a previous trainer in this process)
        if cfg.__class__.__name__ == 'swin_tiny_coco':           # optimizer choice of the reference (train.py:60-63)
            self.opt = FlatAdamW(self.net.parameters(), cfg.lr, weight_decay=0.05)
        else:
            self.opt = FlatSGD(self.net.parameters(), cfg.lr)
        self.model = self.net
        self.ddp = world > 1 or (dist.is_initialized() and os.environ.get('YM_FORCE_DIST', '0') == '1')
        self.reducer, self.buffers_flat = None, None
        self.nbt_flat = flatten_batch_counters(self.net)
        self.torch_ddp = self.ddp and os.environ.get('YM_TORCH_DDP', '0') == '1'
        if self.torch_ddp:                           # the reference's wrapper, kept as an option (per-parameter bucket copies)
            from torch.nn.parallel import DistributedDataParallel as DDP
            self.model = DDP(self.net, device_ids=[local_rank], output_device=local_rank, broadcast_buffers=True,
                             bucket_cap_mb=25, gradient_as_bucket_view=False)
        elif self.ddp:
            dist.broadcast(self.opt.flat, 0)         # every replica starts from rank 0's weights
            self.buffers_flat = flatten_buffers(self.net)
            self.reducer = FlatGradReducer(self.opt, world)
        self.step_idx = 0
        self._timing = None
        self.net.mark_weights_changed()              # parameters were re-pointed at the flat buffer
        weights_changed()

    def enable_timing(self):
        """Bracket the collectives of every following step with HIP events on the step's stream (bench.py `extra.ddp`)."""
        self._timing = []

    def timing_summary(self):
        """Means over the timed steps, in ms (synchronises): `buffer_broadcast_ms` (the one-message BN running-stat broadcast),
        `backward_ms`, `allreduce_exposed_ms` = end of backward on the device (main and weight-gradient streams joined) -> every
        bucket reduced, i.e. what the overlap with backward did NOT hide."""
        tm, self._timing = self._timing or [], None
        if not tm:
            return {}
        torch.cuda.synchronize(self.device)
        n = len(tm)
        return dict(steps=n, buffer_broadcast_ms=round(sum(e[0].elapsed_time(e[1]) for e in tm) / n, 4),
                    backward_ms=round(sum(e[2].elapsed_time(e[3]) for e in tm) / n, 4),
                    allreduce_exposed_ms=round(sum(e[3].elapsed_time(e[4]) for e in tm) / n, 4))

    @property
    def module(self):
        return self.net

    def close(self):
        """Give back what training holds beyond the parameters: the per-layer slab scratch of the deferred weight-gradient
        reductions (~2 GB for res101 at batch 8, growing with the batch; YM_WGRAD_REDUCE_BATCH=0 never allocates it) and the cached
        reduction tables.  Call when the process goes on to evaluate / serve; the next training step re-allocates them."""
        torch.cuda.synchronize(self.device)
        release_wgrad_scratch()

    # ---- full-state checkpoint (the reference's save_latest/save_best keep only net.state_dict(),
    # utils/common_utils.py:41-63, so a resumed run restarts the momentum buffers and the LR warm-up) -------------------
    def state_dict(self):
        """Model weights under the reference's key names (loadable by `Yolact.load_weights` / the reference) plus the
        optimizer's flat momentum buffer, the step counters and the random generator states."""
        return {'model': {k: v.detach().clone() for k, v in self.net.state_dict().items()},
                'momentum': self.opt.buf.detach().clone(), 'opt_steps': self.opt.steps, 'step_idx': self.step_idx,
                'exp_avg_sq': self.opt.exp_avg_sq.detach().clone() if hasattr(self.opt, 'exp_avg_sq') else None,
                'param_numel': [p.numel() for p in self.opt.params],
                # generator states: DropPath and the mask-loss sub-sampling draw from the device generator
                'cuda_rng': torch.cuda.get_rng_state(self.device), 'cpu_rng': torch.get_rng_state(),
                # the augmentation decisions (`train_aug`) draw from python's `random`; the saving rank is recorded so that the
                # other ranks can derive their own streams on resume (a shared state would give every rank identical DropPath
                # masks and mask-loss sub-samples)
                'py_random': __import__('random').getstate(), 'rank': dist.get_rank() if dist.is_initialized() else 0,
                'mask_rng': __import__('yolact_minimal_amd.loss', fromlist=['x']).mask_generator(self.device).get_state()}

    def load_state_dict(self, state):
        if state['param_numel'] != [p.numel() for p in self.opt.params]:
            raise RuntimeError('checkpoint does not match this model (parameter sizes differ)')
        with torch.no_grad():
            own = self.net.state_dict()                       # tensors alias the flat buffers: copy in place
            missing = set(own) ^ set(state['model'])
            if missing:
                raise RuntimeError(f'checkpoint keys differ: {sorted(missing)[:5]}')
            for k, v in state['model'].items():
                own[k].copy_(v)
            self.opt.buf.copy_(state['momentum'])
            if state.get('exp_avg_sq') is not None and hasattr(self.opt, 'exp_avg_sq'):
                self.opt.exp_avg_sq.copy_(state['exp_avg_sq'])
        self.opt.steps, self.step_idx = int(state['opt_steps']), int(state['step_idx'])
        if state.get('cuda_rng') is not None:
            torch.cuda.set_rng_state(state['cuda_rng'].cpu(), self.device)
            torch.set_rng_state(state['cpu_rng'].cpu())
            import random
            if state.get('py_random') is not None:
                random.setstate(state['py_random'])
            rank = dist.get_rank() if dist.is_initialized() else 0
            if state.get('mask_rng') is not None:
                from .loss import mask_generator
                mask_generator(self.device).set_state(state['mask_rng'].cpu())
            if rank != int(state.get('rank', 0)):
                # a different rank than the one that saved: same checkpoint, its OWN random streams (deterministic in step and rank)
                seed = (int(state['step_idx']) * 1000003 + 7919 * rank + 12345) % (2 ** 31)
                torch.cuda.manual_seed(seed)
                torch.manual_seed(seed)
                random.seed(seed)
                from .loss import mask_generator
                mask_generator(self.device).manual_seed(seed + 1)
        self.net.mark_weights_changed()
        weights_changed()

    def save(self, path):
        torch.save(self.state_dict(), path)

    def load(self, path):
        self.load_state_dict(torch.load(path, map_location=self.device))

    def step(self, images, targets, masks):
        self.opt.lr = lr_at(self.cfg, self.step_idx)
        ev = None
        if self._timing is not None:
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
            ev[0].record()
        if self.buffers_flat is not None and self.world > 1:
            dist.broadcast(self.buffers_flat, 0)      # BN running stats follow rank 0, one 0.4 MB message
        if ev is not None:
            ev[1].record()
        self.opt.zero_grad()                          # before the forward: the host is ahead of the device here
        losses = self.model(images, targets, masks)
        if self.nbt_flat is not None:
            self.nbt_flat += 1                        # every BatchNorm ran once (num_batches_tracked)
        if self.ddp:
            all_loss = torch.stack([l.detach() for l in losses])
            dist.all_reduce(all_loss)                 # 16-byte logging collective, train.py:121-122
        total = losses[0] + losses[1] + losses[2] + losses[3]
        from .loss import unit_loss_grads
        from .train_engine import wgrad_on_side_stream, check_links_drained
        # weight gradients run on a side stream next to the data-gradient chain; the main stream waits for them at the block's end
        if ev is not None:
            ev[2].record()
        with unit_loss_grads(), wgrad_on_side_stream(self.device):   # d(total)/d(loss_i) = 1: stored loss gradients pass through unscaled
            total.backward()
        check_links_drained()                         # every gradient parked for a later consumer was consumed (host-side check)
        if ev is not None:
            ev[3].record()                            # backward done on the device: the weight-gradient stream was joined at the block's end
        if self.reducer is not None:
            self.reducer.finish()
        if ev is not None:
            ev[4].record()                            # ... and every bucket reduced (the step's stream waited for the RCCL stream)
            self._timing.append(ev)
        self.opt.step()
        self.net.mark_weights_changed()
        self.step_idx += 1
        return losses
