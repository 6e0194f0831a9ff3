"""Training plumbing a `Yolact` brings along by itself, so that the reference's OWN loop runs at the speed of `Trainer`
(`/root/reference/train.py:60-63,76,102-130`: `optim.SGD(net.parameters())`, `DDP(net.cuda(), [local_rank], ...)`,
`net(images, targets, masks)`, `optimizer.zero_grad()`, `loss_total.backward()`, `optimizer.step()` — not a line of it changed).

Measured on one MI355X, res101_coco 544 px batch 8 (profiles/r06_reference_loop_*): that loop took 72.3 ms per step against 41.6 for
`Trainer.step` with bit-identical results; with this module 46.3 (fenced by the reference's timer) / 43.1 (fences removed).  Where the 31 ms went: 28 of them inside torch's DistributedDataParallel — per step 323
`mul` launches (gradient / world size into the bucket), 419 bucket -> gradient copies, and the coalesced broadcast of the 416
BatchNorm buffers (flatten + 416 copies back) — all issued one by one from the autograd thread; the rest: one weight re-pack launch
per conv and direction (no pack cache without an owner), weight gradients on the main stream, 104 `num_batches_tracked += 1` launches.

`ModuleTrainState` (created at the first train-mode forward on the GPU when no `Trainer` owns the parameters) gives the module what
`Trainer` gives it:

  * one flat gradient buffer; every parameter's slice is the slot the HIP weight-gradient kernels write and autograd adopts as
    `p.grad` (zero copies; torch optimizers read `p.grad` as usual).  `optimizer.zero_grad()` (set_to_none) frees a slot; a
    gradient that is still there (accumulation, `set_to_none=False`) is accumulated into, as autograd would;
  * the packed weight images live in the per-device pack cache (one batched re-pack launch per step);
  * weight gradients run on the side stream with batched slab reductions, joined by a callback on the autograd engine when the
    backward pass ends (`train_engine._auto_backward_end`) — `backward()` returns with ordinary stream semantics;
  * BatchNorm running statistics are views of one flat tensor, `num_batches_tracked` of one int64 tensor (one increment per step);
  * under `torch.nn.parallel.DistributedDataParallel` the module reduces its own gradients: `FlatGradReducer` all-reduces (AVG)
    contiguous >= 25 MB ranges of the flat buffer on the RCCL stream while backward is still running and the end-of-backward
    callback waits for them, the flat BatchNorm buffer is broadcast from rank 0 at every train-mode forward
    (`broadcast_buffers=True`), the parameters once.  torch's wrapper is told to leave these tensors alone through the hook it reads
    for that purpose, `module._ddp_params_and_buffers_to_ignore` (see `Yolact`); it keeps ONE 12-float parameter
    (`prediction_layers.bbox_layer.bias`), because it refuses to wrap a module it has nothing to do for.

`YM_AUTO_FLAT=0` switches all of it off (torch DDP then does everything itself, as in round 5).

Hardware queues: keep GPU_MAX_HW_QUEUES at its default (4) for training.  With 8 (what the serving pipeline wants) the default
stream, the side stream, RCCL's stream and DDP's own streams each get a queue, more than the part's four compute pipes: the reference
loop without its timer fences then takes 54.7 instead of 43.1 ms per step and the one-rank reducer costs 5 ms instead of 0.3
(profiles/r06_reference_loop_timings.txt; `dropin/run.py` exports 8 for eval.py / detect.py only).
"""
import os

import torch
import torch.distributed as dist

from .trainer import FlatGradReducer, flatten_buffers, flatten_batch_counters

AUTO = os.environ.get('YM_AUTO_FLAT', '1') != '0'
DDP_KEEPS = 'prediction_layers.bbox_layer.bias'      # the one parameter left to torch's DDP (a fresh main-stream gradient, 48 bytes)


class ModuleTrainState:
    def __init__(self, net, device):
        self.device = device
        named = [(n, p) for n, p in net.named_parameters() if p.requires_grad]
        self.params = [p for n, p in named if n != DDP_KEEPS]
        starts, n = [], 0
        for p in self.params:                      # 64-byte aligned slots (the conv epilogues want 16-byte aligned vectors)
            starts.append(n)
            n += (p.numel() + 15) // 16 * 16
        self.grad = torch.zeros(n, device=device, dtype=torch.float32)
        self.flat = self.grad                      # (FlatGradReducer reads `.flat.numel()` for the end of the first bucket)
        self.offsets = []
        for p, off in zip(self.params, starts):
            k = p.numel()
            self.offsets.append((off, off + k))
            p._ym_grad_slot = self.grad[off:off + k].view_as(p.data)
            p._ym_auto = self
        self.all_params = [p for _, p in named]
        self.buffers_flat = flatten_buffers(net)
        self.nbt_flat = flatten_batch_counters(net)
        self.bns = [m for m in net.modules() if getattr(m, '_ym_nbt_flat', False)]
        self.reducer = None
        self.synced_params = False
        self.side_stream = os.environ.get('YM_AUTO_SIDE_STREAM', '1') != '0'
        self.begin_forward()

    def release(self):
        """(`net.cuda()` / `.to()` re-allocate parameters and buffers, a `Trainer` brings its own slots: Yolact drops the state.)"""
        for m in self.bns:
            m._ym_nbt_flat = False                 # the BatchNorms count their own batches again
        for p in self.params:
            for a in ('_ym_grad_slot', '_ym_auto', '_ym_slot_free', '_ym_in_slot', '_ym_side_written'):
                if hasattr(p, a):
                    delattr(p, a)
        if self.reducer is not None:
            for h in self.reducer._hooks:
                h.remove()
            self.reducer = None

    # ---- per step ---------------------------------------------------------------------------------------------------------------
    def begin_forward(self):
        """Every slot is handed out at most once per backward (`train_engine._grad_slot`); whether it is free is decided there, when
        the optimizer's zero_grad() has already run (train.py:126 sits between the forward and the backward pass)."""
        for p in self.params:
            p._ym_slot_free = True
            p._ym_in_slot = False
            p._ym_side_written = False

    def distributed(self, wrapped):
        """Gradient / buffer synchronisation is the module's job iff torch's DDP wrapped it (and was told to ignore its tensors)."""
        if not (wrapped and dist.is_initialized()):
            return False
        return dist.get_world_size() > 1 or os.environ.get('YM_FORCE_DIST', '0') == '1'

    def sync_before_forward(self, wrapped, grad_sync=True):
        """`grad_sync`: the wrapper's `require_backward_grad_sync` at this forward — False inside `DDP.no_sync()` (gradient
        accumulation): the coming backward then leaves its gradients local, as torch's reducer would; the next synchronised backward
        accumulates into them and all-reduces the sum."""
        if not self.distributed(wrapped):
            return
        if self.reducer is None:
            self.reducer = FlatGradReducer(self, dist.get_world_size())
        self.reducer.enabled = bool(grad_sync)
        if not self.synced_params:
            # what DDP's constructor does for the tensors it manages (_sync_module_states): every replica starts from rank 0's weights
            if dist.get_world_size() > 1:
                dist._broadcast_coalesced(dist.group.WORLD, [p.data for p in self.all_params], 250 << 20, 0)
                if self.nbt_flat is not None:
                    dist.broadcast(self.nbt_flat, 0)
            self.synced_params = True
        if self.buffers_flat is not None and dist.get_world_size() > 1:
            dist.broadcast(self.buffers_flat, 0)          # DDP(broadcast_buffers=True): BN running statistics follow rank 0

    def after_forward(self):
        if self.nbt_flat is not None:
            self.nbt_flat += 1                            # every BatchNorm ran once

    def end_backward(self):
        """(from the autograd engine's end-of-backward callback, after the weight-gradient stream was joined)"""
        if self.reducer is not None:
            self.reducer.finish()
            # `finish()` treats a parameter without a gradient as zeros IN THE FLAT BUFFER; the caller's view is p.grad
        for p in self.params:
            p._ym_in_slot = False
