"""Training forward/backward of YOLACT on MI355X.

Reference: `Yolact.forward` train branch + `compute_loss` (`/root/reference/modules/yolact.py:141-203`) and the
step in `train.py:116-130` (`loss.backward()` through autograd/ATen/cuDNN, `optimizer.step()`).

Here every convolution (forward, data gradient, weight gradient), every train-mode BatchNorm (batch statistics,
running-stat update, backward), max-pool / bilinear backward and the SGD update run as hand-written HIP kernels
behind the C-ABI (`include/yolact_hip.h`); tensors stay NHWC fp32 between them.  `torch.autograd.Function` is
used only as the tape that orders those kernels and hands parameter gradients to `torch.distributed` (DDP /
RCCL) — no ATen convolution, batch-norm or pooling kernel is ever called.  The loss bookkeeping on the [B,N,*]
head outputs (matching, OHEM ranking, cross-entropy, smooth-L1, mask BCE) is `yolact_minimal_amd/loss.py`.
"""
import ctypes
import math
import os
import weakref

import torch
import torch.nn.functional as F

from . import hip
from .hip import ConvDesc, WgradDesc, ACT_NONE, ACT_RELU, ACT_TANH

_scratch = {}
_desc_cache = {}      # shape key -> (descriptor with every shape-dependent field set, ...): see _conv_forward
launch_counts = None  # tools/train_layer_table.py sets this to a dict: desc-cache key -> launches (per-shape accounting of a step)


def _count(key):
    if launch_counts is not None:
        launch_counts[key] = launch_counts.get(key, 0) + 1

# ---- per-shape kernel configuration (measured on MI355X; same JSON as the inference engine) ---------------------
_TUNING = os.environ.get('YM_TUNE_TRAIN', '0') == '1'     # sweep unseen shapes inline and remember the winner
_new_entries = {}


def _tuning():
    """Inline sweeps: tools/autotune_train.py (YM_TUNE_TRAIN=1 at import) or the opt-in tune-on-first-use mode (YM_AUTOTUNE=1, read
    per call like engine.autotune_on)."""
    return _TUNING or os.environ.get('YM_AUTOTUNE', '0') == '1'


def _remember(key, hit):
    """A row the inline sweep measured: kept for tools/autotune_train.py (dump_new_entries) and, with YM_AUTOTUNE=1, written through
    to the per-user cache that later processes overlay on the shipped table (engine.tuned_table)."""
    _table()[key] = hit
    _new_entries[key] = hit
    from .engine import autotune_on, _store_user_rows, user_cache_path
    if autotune_on():
        try:
            _store_user_rows({key: hit})
        except OSError as e:
            import sys
            print(f'yolact_minimal_amd: could not write {user_cache_path()}: {e}', file=sys.stderr)


def train_mma():
    return int(os.environ.get('YM_TRAIN_MMA', '0') or 0)


def _table():
    from .engine import tuned_table
    return tuned_table()


def tuned_table_changed():
    """The cached launch descriptors keep the tile / split / staging choice of their first use: call this after editing the tuned
    table (tests, tuning tools) so that the next launch of every shape reads its entry again."""
    _desc_cache.clear()


def _time_launch(fn, iters=5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); fn()
    best = 1e30
    for _ in range(2):
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters)
    return best


def _configure_conv(d, key, stats=False):
    """Set tile/ksplit/kwaves of a ConvDesc from the tuned table (or sweep it when YM_TUNE_TRAIN=1).  `stats`: the launch carries
    fused BatchNorm sums, which the persistent kernel does not do: `<key>_st` holds the per-item choice measured for such launches
    where the plain entry (shared with inference) selects the persistent kernel."""
    from . import plan_transfer
    M_rows = d.B * d.Ho * d.Wo
    hit = None
    if plan_transfer.mode() != 'only':
        hit = (_table().get(key + '_st') if stats else None) or _table().get(key)
    if hit is None and not _tuning() and plan_transfer.mode() != 'off':
        # another --img_size / batch: the row of the nearest tuned shape of the family, re-derived for this M (plan_transfer.py);
        # with fused statistics the `_st` family competes with the plain one, the donor nearer in M wins
        only = plan_transfer.mode() == 'only'
        donors = [(abs(math.log2(nb[1] / M_rows)), i, k) for i, k in enumerate(([key + '_st'] if stats else []) + [key])
                  for nb in [plan_transfer.nearest(_table(), k, only)] if nb is not None]
        if donors:
            k = min(donors)[2]
            hit, _ = plan_transfer.lookup(_table(), k, M_rows, d.Cout, d.k_pad // 32, d.nseg)
            if hit is not None and stats and k == key and len(hit) > 4 and 42 <= hit[4] <= 48:
                hit = list(hit[:7])
                hit[4] = 22 if hit[4] == 42 else 23     # the persistent walker does not cover launches with fused BatchNorm sums
    if hit is None and _tuning():
        M, nkt = d.B * d.Ho * d.Wo, d.k_pad // 32
        big = scratch(torch.device('cuda', torch.cuda.current_device()), 1 << 28)
        d.tile_counters = _tile_counters(torch.device('cuda', torch.cuda.current_device()))
        cands = [((0, 0), 0, 0, 0, (0, 0))]
        for tm, tn in ((128, 128), (128, 64), (64, 128), (64, 64)):
            wgs = -(-M // tm) * -(-d.Cout // tn)
            for ks in (1, 2, 3, 4, 6, 8, 12, 16):
                if ks > 1 and (wgs >= 1024 or ks * 2 > nkt or wgs * ks > 8192):
                    continue
                cands.append(((tm, tn), ks, 0, 2, (0, 0)))
                cands.append(((tm, tn), ks, 0, 22, (0, 0)))         # direct-to-LDS staging
                if (tm, tn) == (64, 64) and nkt // ks >= 3:
                    cands.append(((tm, tn), ks, 0, 3, (0, 0)))
                    cands.append(((tm, tn), ks, 0, 23, (0, 0)))
            if 256 < wgs <= hip.TILE_COUNTERS:          # split the last partial round of tiles (ym_conv_desc.tail_tiles)
                for r in sorted({wgs % 256, wgs % 512} - {0}):
                    for ts in (2, 3, 4, 6, 8):
                        if ts * 2 <= nkt and r * ts <= 2048:
                            cands.append(((tm, tn), 1, 0, 2, (r, ts)))
                            cands.append(((tm, tn), 1, 0, 22, (r, ts)))
        best = (1e30, (0, 0), 0, 0, 0, (0, 0))
        for tile, ks, kwv, stg, tail in cands:
            d.tile_m, d.tile_n, d.ksplit, d.kwaves, d.stages = tile[0], tile[1], ks, kwv, stg
            d.tail_tiles, d.tail_ksplit = tail
            if hip.conv_workspace_bytes(d) > big.numel():
                continue
            try:
                t = _time_launch(lambda: hip.conv2d_fwd(d, big))
            except RuntimeError:
                continue
            if t < best[0] * 0.98:
                best = (t, tile, ks, kwv, stg, tail)
        hit = [best[1][0], best[1][1], best[2], best[3], best[4], best[5][0], best[5][1]]
        _remember(key, hit)
    if hit is not None:
        d.tile_m, d.tile_n, d.ksplit = hit[0], hit[1], hit[2]
        d.kwaves = hit[3] if len(hit) > 3 else 0
        d.stages = hit[4] if len(hit) > 4 else 0
        d.tail_tiles, d.tail_ksplit = (hit[5], hit[6]) if len(hit) > 6 else (0, 0)
        from .engine import _grid_wgs
        d.grid_wgs = _grid_wgs(hit)                             # persistent kernel (stages 4x): workgroups launched
    force = os.environ.get('YM_FORCE_STAGES')        # experiments / tests: e.g. 43 = every conv the persistent kernel covers runs on it
    if force:
        d.tile_m, d.tile_n, d.kwaves, d.stages, d.grid_wgs = 64, 64, 0, int(force), int(os.environ.get('YM_FORCE_GRID', '0'))
        if d.tail_tiles and d.ksplit > 1:
            d.tail_tiles = d.tail_ksplit = 0
    mma = train_mma()
    if mma and d.Cin % 32 == 0 and d.nlevels == 0:
        # opt-in FAST training mode (YM_TRAIN_MMA=3): forward and data-gradient convs on the bf16 MFMA (split-bf16 products, see
        # ym_conv_desc.mma).  NOT the parity mode: per-product error ~2^-17 instead of 2^-24, which the ill-conditioned backward of
        # a random-init net amplifies beyond the fp32 reference's own noise (tests keep the default, f32).
        hit3 = _table().get(key + f'_mma{mma}')
        if hit3 is not None:
            d.tile_m, d.tile_n, d.ksplit, d.kwaves = hit3[0], hit3[1], hit3[2], (hit3[3] if len(hit3) > 3 else 0)
            d.tail_tiles, d.tail_ksplit = (hit3[5], hit3[6]) if len(hit3) > 6 else (0, 0)
        if d.kwaves == 0:
            d.mma, d.stages = mma, 0


_MSPLIT_SCALE = float(os.environ.get('YM_WGRAD_MSPLIT_SCALE', '1'))


def _configure_wgrad(d, key):
    from . import plan_transfer
    hit = _table().get(key) if plan_transfer.mode() != 'only' else None
    if hit is None and not _tuning():
        p = plan_transfer.parse(key)
        if p:
            hit, _ = plan_transfer.lookup(_table(), key, p[1], p[2], 0)
    if hit is None and _tuning():
        big = scratch(torch.device('cuda', torch.cuda.current_device()), 1 << 28)
        best = (1e30, 0, 2)
        tbn = 64 if d.Cout_real <= 64 else 128
        tiles = -(-d.Cout // tbn) * -(-(d.KH * d.KW * d.Cin) // 128)
        # (22 / 23 / 24: the DMA rings; layers with <= 64 output channels have the 32-pixel ring only, at 48 KB = 3 workgroups per CU)
        for nb, slots in ((2, 512), (1, 768)) + (((22, 512), (23, 768), (24, 512)) if d.Cout_real > 64 else ((22, 768),)):
            # pixel splits that make the grid a whole number of rounds of the chip (2 workgroups of 68 KB LDS per CU = 512 slots,
            # 3 of 34 KB with the single-buffer variant = 768): power-of-two splits alone left e.g. 756 workgroups = 1.5 rounds
            # for the layer3 3x3 (msplit 14 -> 504 = one round)
            half = slots // 2
            fill = {max(1, round(half * j / tiles)) for j in (1, 2, 3, 4, 6, 8, 12, 16)} | {max(1, (half * j) // tiles) for j in (2, 4, 6, 8)}
            for ms in sorted({0, 1, 2, 4, 8, 16, 32, 64, 128, 256} | {m_ for m_ in fill if m_ <= 256}):
                d.msplit, d.lds_buffers = ms, nb
                need = hip.lib().ym_conv2d_wgrad_workspace_bytes(ctypes.byref(d))
                if need == 0 or need > big.numel():
                    continue
                t = _time_launch(lambda: hip.check(hip.lib().ym_conv2d_wgrad(ctypes.byref(d), ctypes.c_void_p(big.data_ptr()),
                                                                            big.numel(), hip.stream_ptr()), 'wgrad'))
                if t < best[0] * 0.98:
                    best = (t, ms, nb)
        hit = [best[1], best[2]]
        _remember(key, hit)
    if hit is not None:
        d.msplit = hit[0]
        d.lds_buffers = hit[1] if len(hit) > 1 else 2
        if _MSPLIT_SCALE != 1.0 and d.msplit > 1:    # experiment knob: the table was tuned with the chip to itself
            d.msplit = max(1, int(round(d.msplit * _MSPLIT_SCALE)))


def dump_new_entries(path):
    import json
    with open(path, 'w') as f:
        json.dump(_new_entries, f, indent=0, sort_keys=True)


def scratch(device, nbytes):
    """Stream-ordered scratch: one buffer per (device, stream), grown on demand (kernels on one stream never overlap)."""
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    buf = _scratch.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
        _scratch[key] = buf
    return buf


def _ru(x, m):
    return (x + m - 1) // m * m


_counters = {}


def _tile_counters(device):
    """Arrival counters of the fused split-K finish (ym_conv_desc.tile_counters): zeroed once, kernels leave them zero; the
    training step is one stream, so every conv shares them."""
    t = _counters.get(device)
    if t is None:
        t = _counters[device] = torch.zeros(hip.TILE_COUNTERS, device=device, dtype=torch.int32)
    return t.data_ptr()


class _StatsPool:
    """fp64 accumulators for the fused BN statistics of every ConvBn of one forward: ONE buffer, zeroed by ONE fill per step,
    handed out in slices (instead of an allocation + fill launch per layer)."""

    def __init__(self):
        self.buf, self.off, self.want = None, 0, 0

    def begin(self, device):
        if self.buf is None or self.buf.device != device or self.buf.numel() < self.want:
            self.buf = torch.empty(max(self.want, 1 << 16), device=device, dtype=torch.float64)
        self.buf.zero_()
        self.off, self.want = 0, 0

    def take(self, n, device):
        self.want += n
        if self.buf is not None and self.buf.device == device and self.off + n <= self.buf.numel():
            out = self.buf[self.off:self.off + n]
            self.off += n
            return out
        return torch.zeros(n, device=device, dtype=torch.float64)      # pool too small this step: grows at the next begin()


_stats_pool = _StatsPool()
# ---- packed weight images ---------------------------------------------------------------------------------------------------
# Every conv needs its OIHW weight as a forward image ([Cout][k_pad]) and, in backward, as a dgrad image ([Cin][KH][KW][Cout]).
# After an optimizer step all of them are stale at once: when the trainer's flat optimizer owns the parameters (FlatSGD sets
# `_ym_grad_slot`, and is the only writer: it bumps `weights_epoch()`), the images live in persistent buffers and ONE
# `ym_pack_conv_weights_batch` launch refreshes all of them on the first request of a step (181 launches -> 1 for res101).
# Parameters driven by anything else (a torch optimizer, the reference loop verbatim) are packed per use, as before.
_EPOCH = [0]


def weights_changed():
    """Raw-pointer writers of parameter memory (FlatSGD / FlatAdamW step, checkpoint load, broadcast) call this."""
    _EPOCH[0] += 1


class _PackCache:
    def __init__(self):
        self.entries = {}          # key -> dict(ref, dst, item fields, stamp)
        self.table = None          # device copy of the ym_pack_item array
        self.order = []
        self.total_chunks = 0

    @staticmethod
    def _owner(weight):
        base = weight._base if weight._base is not None else weight
        return base if getattr(base, '_ym_grad_slot', None) is not None else None

    def get(self, weight, kind, pad_a, pad_b, rows):
        """Packed image of `weight` (kind 0: forward, 1: dgrad), or None when the parameter is not trainer-owned."""
        owner = self._owner(weight)
        if owner is None or not weight.is_contiguous() or os.environ.get('YM_PACK_CACHE', '1') == '0':
            return None
        cout, cin, kh, kw = weight.shape
        key = (weight.data_ptr(), cout, cin, kh, kw, kind, pad_a, pad_b, rows)
        e = self.entries.get(key)
        stamp = (weight._version, _EPOCH[0])
        if e is not None and e['ref']() is owner:
            if e['stamp'] != stamp:
                self.refresh()
            return e['dst']
        import weakref
        n = rows * pad_b if kind == 0 else cin * kh * kw * pad_a
        e = dict(ref=weakref.ref(owner),
                 dst=torch.empty(n, device=weight.device, dtype=torch.float32), src=weight.data_ptr(), dims=(cout, cin, kh, kw),
                 kind=kind, pad_a=pad_a, pad_b=pad_b, rows=rows, stamp=None,
                 chunks=(n + 1023) // 1024 if kind == 0 else -(-(cin * kh * kw) // 32) * (pad_a // 32))
        self.entries[key] = e
        self.table = None                                   # rebuilt at the next refresh
        L = hip.lib()                                       # a new image is packed on its own
        if kind == 0:
            hip.check(L.ym_pack_conv_weight(hip.ptr(weight.detach()), hip.ptr(e['dst']), cout, cin, kh, kw, pad_a, pad_b, hip.stream_ptr()),
                      'ym_pack_conv_weight')
            if rows > cout:
                e['dst'][cout * pad_b:].zero_()
        else:
            hip.check(L.ym_pack_conv_weight_dgrad(hip.ptr(weight.detach()), hip.ptr(e['dst']), cout, cin, kh, kw, pad_a, hip.stream_ptr()),
                      'ym_pack_conv_weight_dgrad')
        e['stamp'] = stamp
        return e['dst']

    def refresh(self):
        """Re-pack every live image in one launch."""
        for ref in list(_pre_refresh_hooks):                 # derived weights (the concatenated head filter) follow their sources first
            obj = ref()
            if obj is None:                                  # its module is gone: drop the hook
                _pre_refresh_hooks.remove(ref)
            else:
                obj.sync()
        # (an owner that lost its gradient slot was re-allocated or released -- Module._apply, ModuleTrainState.release: its source
        #  pointer is stale)
        dead = [k for k, e in self.entries.items() if e['ref']() is None or getattr(e['ref'](), '_ym_grad_slot', None) is None]
        for k in dead:
            del self.entries[k]
            self.table = None
        if not self.entries:
            return
        any_e = next(iter(self.entries.values()))
        dev = any_e['dst'].device
        if self.table is None:
            self.order = list(self.entries.values())
            arr = (hip.PackItem * len(self.order))()
            at = 0
            for it, e in zip(arr, self.order):
                it.src, it.dst = e['src'], e['dst'].data_ptr()
                it.cout, it.cin, it.kh, it.kw = e['dims']
                it.pad_a, it.pad_b, it.rows, it.kind, it.first_chunk = e['pad_a'], e['pad_b'], e['rows'], e['kind'], at
                at += e['chunks']
            self.total_chunks = at
            host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
            self.table = host.to(dev)
        with torch.cuda.device(dev):
            hip.check(hip.lib().ym_pack_conv_weights_batch(ctypes.c_void_p(self.table.data_ptr()), len(self.order), self.total_chunks,
                                                           hip.stream_ptr()), 'ym_pack_conv_weights_batch')
        for e in self.order:
            owner = e['ref']()
            e['stamp'] = (owner._version, _EPOCH[0])


_pre_refresh_hooks = []     # weak references to objects whose .sync() runs before a batched re-pack (see _HeadWeights)
_pack_caches = {}          # one cache (and one device-side table) per device


def _pack_cache_for(device):
    c = _pack_caches.get(device)
    if c is None:
        c = _pack_caches[device] = _PackCache()
    return c


def _pack_fwd(weight, cin_pad, cout_pad):
    cout, cin, kh, kw = weight.shape
    k_pad = _ru(kh * kw * cin_pad, 32)
    if kh == 1 and kw == 1 and cin_pad == cin and cout_pad == cout and cin % 32 == 0 and weight.is_contiguous():
        return weight.detach().view(cout, cin), cin        # OIHW of a 1x1 conv IS the packed [Cout][K] image: no copy
    cached = _pack_cache_for(weight.device).get(weight, 0, cin_pad, k_pad, cout_pad)
    if cached is not None:
        return cached.view(cout_pad, k_pad), k_pad
    if cout_pad == cout:
        return hip.pack_conv_weight(weight.detach(), cin_pad, k_pad), k_pad
    wp = torch.zeros(cout_pad, k_pad, device=weight.device, dtype=torch.float32)
    hip.check(hip.lib().ym_pack_conv_weight(hip.ptr(weight.detach().contiguous()), hip.ptr(wp), cout, cin, kh, kw, cin_pad,
                                            k_pad, hip.stream_ptr()), 'ym_pack_conv_weight')
    return wp, k_pad


def _pack_dgrad(weight, cout_pad):
    cout, cin, kh, kw = weight.shape
    cached = _pack_cache_for(weight.device).get(weight, 1, cout_pad, 0, 0)
    if cached is not None:
        return cached
    wd = torch.empty(cin, kh * kw * cout_pad, device=weight.device, dtype=torch.float32)
    hip.check(hip.lib().ym_pack_conv_weight_dgrad(hip.ptr(weight.detach().contiguous()), hip.ptr(wd), cout, cin, kh, kw,
                                                  cout_pad, hip.stream_ptr()), 'ym_pack_conv_weight_dgrad')
    return wd


def _conv_forward(x, wp, k_pad, cout_pad, kh, kw, stride, pad, shift, act, residual=None, bn_stats=None, out=None, segs=None):
    """`out`: write into this [B,Ho,Wo,cout_pad] tensor instead of a fresh one.  `segs`: [(n0, n1, tensor_ptr, batch_stride, pitch,
    act)] routes output-channel ranges to separate tensors (the fused prediction head); then nothing is returned."""
    b, h, w, cin = x.shape
    ho, wo = (h + 2 * pad - kh) // stride + 1, (w + 2 * pad - kw) // stride + 1
    y = None
    # Everything of the descriptor that depends only on the layer's SHAPE (sizes, tile / split-K choice, workspace size, whether
    # the BN statistics fuse) is built once per shape and reused: per call only the pointers change.  (The host spends ~35 us per
    # launch in a step of ~1100 launches; with the convs on the bf16 pipe the step is host-bound.)
    key = ('f', x.device.index, b, h, w, cin, cout_pad, kh, kw, stride, pad, act, residual is not None, bn_stats is not None,
           None if segs is None else tuple((n0, n1, bs, pt, a) for n0, n1, _, bs, pt, a in segs), train_mma())
    ent = _desc_cache.get(key)
    if ent is None:
        d = ConvDesc()
        d.B, d.H, d.W, d.Cin, d.Cout, d.KH, d.KW = b, h, w, cin, cout_pad, kh, kw
        d.stride, d.pad, d.Ho, d.Wo, d.k_pad = stride, pad, ho, wo, k_pad
        if segs is None:
            d.nseg = 1
            d.seg[0].n_begin, d.seg[0].n_end = 0, cout_pad
            d.seg[0].batch_stride, d.seg[0].pitch, d.seg[0].act = ho * wo * cout_pad, cout_pad, act
        else:
            d.nseg = len(segs)
            for i, (n0, n1, ptr, bstride, pitch, a) in enumerate(segs):
                d.seg[i].n_begin, d.seg[i].n_end = n0, n1
                d.seg[i].batch_stride, d.seg[i].pitch, d.seg[i].act = bstride, pitch, a
        # pointers of this first call: the inline sweep (YM_TUNE_TRAIN=1) launches with them, and the two queries below look at
        # their alignment (torch allocations are 256-byte aligned, so every later call answers the same)
        d.inp, d.weight = x.data_ptr(), wp.data_ptr()
        d.shift = shift.data_ptr() if shift is not None else None
        d.residual = residual.data_ptr() if residual is not None else None
        if segs is None:
            y = out if out is not None else torch.empty(b, ho, wo, cout_pad, device=x.device, dtype=torch.float32)
            d.seg[0].out = y.data_ptr()
        else:
            for i, sg in enumerate(segs):
                d.seg[i].out = sg[2]
        if cin != 4:
            _configure_conv(d, f'M{b * ho * wo}_N{cout_pad}_C{cin}_k{kh}_s{stride}_seg{d.nseg}_r{int(residual is not None)}', stats=bn_stats is not None)
        d.tile_counters = _tile_counters(x.device)
        fuses = bn_stats is not None and hip.lib().ym_conv2d_fuses_bn_stats(ctypes.byref(d)) == 1
        ent = _desc_cache[key] = (d, fuses, hip.conv_workspace_bytes(d))
    d, fused, ws_bytes = ent
    d.inp, d.weight, d.k_pad = x.data_ptr(), wp.data_ptr(), k_pad
    d.shift = shift.data_ptr() if shift is not None else None
    d.residual = residual.data_ptr() if residual is not None else None
    if segs is None:
        if y is None:
            y = out if out is not None else torch.empty(b, ho, wo, cout_pad, device=x.device, dtype=torch.float32)
        d.seg[0].out = y.data_ptr()
    else:
        for i, sg in enumerate(segs):
            d.seg[i].out = sg[2]
    if fused:
        d.bn_sum, d.bn_sumsq = bn_stats.data_ptr(), bn_stats.data_ptr() + cout_pad * 8
    ws = scratch(x.device, ws_bytes)
    _count(key)
    hip.conv2d_fwd(d, ws)
    if bn_stats is not None:
        return y, fused
    return y


def _conv_dgrad(dz, weight, cout_pad, x_shape, stride, pad, add=None, out=None, bn_bwd=None):
    """dx [B,H,W,Cin] from dz [B,Ho,Wo,cout_pad] (cout_pad % 32 == 0) and the OIHW weight; `add` [B,H,W,Cin] (another
    consumer's gradient of the same tensor) is summed in the epilogue; `out`: destination instead of a fresh tensor.
    `bn_bwd`: the BnGradLink of the ConvBn that PRODUCED x, when dx is the whole gradient of that tensor: the epilogue then
    also accumulates that BatchNorm's backward sums (ym_conv_desc.bnb_*) and leaves them in the link."""
    cout, cin, kh, kw = weight.shape
    b, h, w, cin_x = x_shape
    assert cin_x == cin and cout_pad % 32 == 0
    wd = _pack_dgrad(weight, cout_pad)
    dx = out if out is not None else torch.empty(b, h, w, cin, device=dz.device, dtype=torch.float32)
    key = ('d', dz.device.index, b, dz.shape[1], dz.shape[2], cout_pad, cin, kh, kw, stride, pad, h, w, add is not None, train_mma(),
           bn_bwd is not None)
    ent = _desc_cache.get(key)
    if ent is None:
        d = ConvDesc()
        d.B, d.H, d.W, d.Cin, d.Cout, d.KH, d.KW = b, dz.shape[1], dz.shape[2], cout_pad, cin, kh, kw
        d.stride, d.pad, d.Ho, d.Wo, d.k_pad, d.nseg = stride, pad, h, w, kh * kw * cout_pad, 1
        d.seg[0].n_begin, d.seg[0].n_end = 0, cin
        d.seg[0].batch_stride, d.seg[0].pitch, d.seg[0].act = h * w * cin, cin, ACT_NONE
        d.transposed = 1
        d.inp, d.weight, d.seg[0].out = dz.data_ptr(), wd.data_ptr(), dx.data_ptr()
        d.residual = add.data_ptr() if add is not None else None
        _configure_conv(d, f'T_M{b * h * w}_N{cin}_C{cout_pad}_k{kh}_s{stride}')
        d.tile_counters = _tile_counters(dz.device)
        fuses = bn_bwd is not None and hip.lib().ym_conv2d_fuses_bn_stats(ctypes.byref(d)) == 1
        ent = _desc_cache[key] = (d, hip.conv_workspace_bytes(d), fuses)
    d, ws_bytes, fuses = ent
    d.inp, d.weight, d.seg[0].out = dz.data_ptr(), wd.data_ptr(), dx.data_ptr()
    if add is not None:
        assert tuple(add.shape) == (b, h, w, cin) and add.is_contiguous()
        d.residual = add.data_ptr()
    if fuses:
        assert bn_bwd.c == cin and bn_bwd.m == b * h * w
        stats = _stats_pool.take(2 * cin, dz.device)                 # zeroed
        d.bn_sum, d.bn_sumsq = stats.data_ptr(), stats.data_ptr() + cin * 8
        d.bnb_y, d.bnb_out, d.bnb_mean, d.bnb_invstd = bn_bwd.y, bn_bwd.out, bn_bwd.mean, bn_bwd.invstd
        d.bnb_gamma, d.bnb_beta, d.bnb_relu = bn_bwd.gamma, bn_bwd.beta, bn_bwd.relu
    ws = scratch(dz.device, ws_bytes)
    _count(key)
    hip.conv2d_fwd(d, ws)
    if fuses:
        bn_bwd.stats, bn_bwd.dout_ptr = stats, dx.data_ptr()
    return dx


def _grad_slot(param, shape):
    """Gradient destination: the parameter's slice of the optimizer's flat gradient buffer when the trainer installed
    one (`FlatSGD` sets `param._ym_grad_slot`), so autograd adopts the view and no gather copy is needed; else fresh.
    The slot view is tagged `_ym_is_slot`: only such a destination may be written from the side stream (nothing reads it before
    the optimizer / the bucket all-reduce, which order themselves after that stream).  A fresh tensor is summed / cloned by
    autograd on the main stream, so it is always produced there — and when it is a SECOND gradient of a parameter whose slot was
    already handed out in this backward, the main stream first waits for the side stream (autograd is about to add the two)."""
    slot = getattr(param, '_ym_grad_slot', None) if param is not None else None
    if slot is not None:
        _auto_backward_begin(param)
        owner = getattr(param, '_ym_owner', param)
        if getattr(owner, '_ym_auto', None) is not None and owner.grad is not None:
            # module-owned slots (train_state.ModuleTrainState): the gradient of an earlier backward is still in place (accumulation
            # without zero_grad, or zero_grad(set_to_none=False)) -> a fresh tensor, which autograd adds to it
            param._ym_slot_free = False
    if slot is not None and getattr(param, '_ym_slot_free', False):
        param._ym_slot_free = False           # a second use in the same step must accumulate into a fresh tensor
        v = slot.view_as(slot)                # a FRESH view: AccumulateGrad only adopts (instead of cloning) an unshared tensor
        v._ym_is_slot = True
        return v
    if slot is not None and getattr(getattr(param, '_ym_owner', param), '_ym_side_written', False):
        join_wgrad_stream(param.device)
        getattr(param, '_ym_owner', param)._ym_side_written = False
    return torch.empty(shape, device=param.device if param is not None else None, dtype=torch.float32)


def _side_ok(param, *dsts):
    """May a gradient of `param` be written into `dsts` from the side stream?  Only when every destination is the optimizer's
    slot (see `_grad_slot`) and this autograd node is the parameter's only gradient producer (`_ym_multi_producer`: e.g. Swin's
    qkv.bias also receives a gradient from the window-attention node — autograd sums the two on the main stream)."""
    if not _side_active[0] or param is None or getattr(param, '_ym_multi_producer', False):
        return False
    return all(d is not None and d.is_cuda and getattr(d, '_ym_is_slot', False) for d in dsts)


# ---- weight gradients on a side stream ---------------------------------------------------------------------------------------
# Nothing in backward waits for a weight gradient (only the optimizer step and the bucket all-reduce do), while the data-gradient
# chain is the critical path and neither kernel keeps the MFMA pipe full on its own (0.5-0.8 busy).  So every `_conv_wgrad` is
# enqueued on a second stream that waits for its operands (it is issued BEFORE the layer's data-gradient conv, so the two run side
# by side), with its own scratch; `join_wgrad_stream` makes the main stream wait for it before the optimizer step, and the
# gradient reducer orders each bucket's all-reduce after both streams (`wgrad_stream_if_used`).  Measured on res101 bs=8: 47.7 ->
# 45.5 ms/step with the original call order.  YM_WGRAD_STREAM=0 puts everything back on one stream.
_WGRAD_STREAM = os.environ.get('YM_WGRAD_STREAM', '1') != '0'
_side_streams = {}
_side_active = [False]


class wgrad_on_side_stream:
    """`with wgrad_on_side_stream(device): loss.backward()` — inside, weight gradients go to the side stream; on exit the current
    stream waits for them, so the caller sees ordinary stream semantics.  (Outside such a block — a bare `.backward()` in a test or
    in the reference's loop — everything stays on the current stream: a gradient read right after backward must be complete.)"""

    def __init__(self, device):
        self.device = device

    def __enter__(self):
        self.prev = _side_active[0]
        _side_active[0] = _WGRAD_STREAM
        return self

    def __exit__(self, *exc):
        _side_active[0] = self.prev
        join_wgrad_stream(self.device)
        return False


# ---- a backward pass nobody brackets (the reference loop: a bare `loss_total.backward()`, train.py:127) ---------------------------
# When the parameters' gradient slots belong to a `train_state.ModuleTrainState`, the first gradient request of a backward pass
# (`_grad_slot` / `_conv_wgrad`) switches the weight-gradient side stream on and queues `_auto_backward_end` on the autograd engine,
# which runs when this backward's graph task has finished (with the caller's current stream current): it joins the side stream
# (pending slab reductions first), lets the module's gradient reducer wait for its buckets, and restores the switch.  `backward()`
# therefore returns exactly like torch's own: every `p.grad` complete with respect to the caller's stream.
_auto = []            # the ModuleTrainStates whose parameters took part in the backward pass that is running (usually one)
_auto_prev_side = [False]


def _auto_backward_begin(param):
    if param is None:
        return
    st = getattr(getattr(param, '_ym_owner', param), '_ym_auto', None)
    if st is None or st in _auto:
        return
    if not _auto:
        try:
            torch.autograd.Variable._execution_engine.queue_callback(_auto_backward_end)
        except RuntimeError:             # not inside a backward pass: nothing to bracket
            return
        _auto_prev_side[0] = _side_active[0]
    _auto.append(st)
    _side_active[0] = _side_active[0] or (_WGRAD_STREAM and st.side_stream)


def _auto_backward_end():
    states = list(_auto)
    del _auto[:]
    if not states:
        return
    _side_active[0] = _auto_prev_side[0]
    for dev in {st.device for st in states}:
        join_wgrad_stream(dev)
    for st in states:
        st.end_backward()


def _dev_key(device):
    device = device if isinstance(device, torch.device) else torch.device(device)
    return torch.device('cuda', device.index if device.index is not None else torch.cuda.current_device())


def wgrad_stream_if_used(device):
    """The (first) side stream if a weight gradient was ever enqueued, after making it wait for the other side streams — None
    otherwise.  Collectives over gradients are issued from it (trainer.FlatGradReducer)."""
    s = _side_streams.get(_dev_key(device))
    if s is not None:
        flush_wgrad_reduces(device)                  # the bucket about to be all-reduced must hold finished gradients
        for e in _extra_streams.get(_dev_key(device), ()):
            s.wait_stream(e)
    return s


# YM_WGRAD_STREAMS=n takes n side streams in turn (consecutive layers' weight gradients are independent of each other as well).
# Measured with 2, same box: res101 bs=8 45.4 -> 44.6 ms/step, but bs=16 83.3 -> 84.4 and Swin-T 35.6 -> 35.8; a third brings
# nothing -> the default stays one.  Launches into a caller-owned destination (the shared head's accumulating gradients) always use
# the first stream, which keeps them ordered.
_N_SIDE = int(os.environ.get('YM_WGRAD_STREAMS', '1'))
# YM_WGRAD_STREAM_PRIORITY=-1 creates the side stream(s) on a high-priority hardware queue (0 = like any other stream):
# both orders were measured in round 4 (the data-gradient chain ahead of the weight gradients and the reverse): no change.
_SIDE_PRIO = int(os.environ.get('YM_WGRAD_STREAM_PRIORITY', '0'))
_extra_streams = {}
_rr = [0]


def wgrad_stream(device, ordered=True):
    device = _dev_key(device)
    s = _side_streams.get(device)
    if s is None:
        s = _side_streams[device] = torch.cuda.Stream(device=device, priority=_SIDE_PRIO)
    if ordered or _N_SIDE <= 1:
        return s
    ex = _extra_streams.setdefault(device, [torch.cuda.Stream(device=device, priority=_SIDE_PRIO) for _ in range(_N_SIDE - 1)])
    _rr[0] = (_rr[0] + 1) % _N_SIDE
    return s if _rr[0] == 0 else ex[_rr[0] - 1]


def join_wgrad_stream(device):
    """Main stream waits for every weight gradient enqueued so far (call before the optimizer step / a gradient all-reduce)."""
    flush_wgrad_reduces(device)
    s = _side_streams.get(_dev_key(device))
    if s is not None:
        torch.cuda.current_stream(_dev_key(device)).wait_stream(s)
    for e in _extra_streams.get(_dev_key(device), ()):
        torch.cuda.current_stream(_dev_key(device)).wait_stream(e)
    _side_keep.pop(_dev_key(device), None)         # everything the side streams read is ordered before the main stream from here on


# Operands of a side-stream launch (x, dz, y) are OWNED by the main stream's autograd pass, which may
#   (a) free them as soon as the node returns -> the caching allocator would hand the block to the next main-stream kernel, and
#   (b) ACCUMULATE IN PLACE into a tensor it holds the only reference to (InputBuffer: `old += new` when use_count == 1) -- e.g. the
#       `dz` a residual conv returns as the gradient of its residual input, which autograd then sums with the other branch's;
# both while a lagging side stream has not read them yet.  Holding a reference until the side stream has passed the launch covers
# both (a block in use is not recycled; use_count > 1 forces the out-of-place sum) without `record_stream`, whose deferred frees
# had inflated the allocator's reservation from 8 to 30 GiB.  Entries are dropped as soon as their event has completed.
_side_keep = {}


def _keep_for_side(device, side, *tensors):
    import collections
    q = _side_keep.setdefault(_dev_key(device), collections.deque())
    ev = torch.cuda.Event()
    ev.record(side)
    q.append((ev, tensors))
    while len(q) > 1 and q[0][0].query():
        q.popleft()


# ---- batched slab reductions ----------------------------------------------------------------------------------------------------
# A weight gradient is a slab pass (msplit partial gradients over pixel ranges) + a reduction that sums the slabs in order and
# scatters into OIHW.  On the side stream the reduction of every layer was its own 5-20 us launch (128 per res101 step, 3 ms of
# kernel time next to the data-gradient chain).  Inside `wgrad_on_side_stream` the slabs of single-destination gradients stay in
# per-layer scratch instead (persistent: ~2 GB for res101 at batch 8) and ONE `ym_wgrad_reduce_batch` launch reduces every pending
# layer: when YM_WGRAD_REDUCE_BATCH layers are pending, before a gradient bucket is all-reduced (`wgrad_stream_if_used`) and when
# the streams join.  Same sums in the same order as the per-layer launch.  YM_WGRAD_REDUCE_BATCH=0: per-layer launches.
_REDUCE_BATCH = int(os.environ.get('YM_WGRAD_REDUCE_BATCH', '24'))
_slab_arena = {}          # (device, destination pointer) -> that layer's private slab scratch
_pending_reduce = {}      # device -> [WgradReduceItem]
_reduce_tables = {}       # (device, the items' bytes) -> (device copy of the item table, total blocks)
wgrad_reduce_launches = [0, 0]       # [batched launches, layers reduced by them] (tests / tools look at it)


def release_wgrad_scratch():
    """Drop the per-layer slab scratch and the cached item tables (a new Trainer calls this: the arena is keyed by gradient-slot
    addresses, which belong to the previous trainer's flat buffer)."""
    for dev, items in _pending_reduce.items():
        if items:
            raise RuntimeError('release_wgrad_scratch() with slab reductions still pending')
    _slab_arena.clear()
    _reduce_tables.clear()


def flush_wgrad_reduces(device):
    """Launch the pending slab reductions (on the side stream: that is where their slab passes ran)."""
    device = _dev_key(device)
    items = _pending_reduce.get(device)
    if not items:
        return
    _pending_reduce[device] = []
    key = (device, b''.join(bytes(it) for it in items))          # (pointers and plans are the same every step: the table is built once)
    ent = _reduce_tables.get(key)
    if ent is None:
        arr = (hip.WgradReduceItem * len(items))()
        at = 0
        for dst, it in zip(arr, items):
            ctypes.memmove(ctypes.byref(dst), ctypes.byref(it), ctypes.sizeof(it))
            dst.first_block = at
            at += it.blocks
        host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
        if len(_reduce_tables) >= 64:                               # (a gradient reducer's bucket timing can regroup the flushes: bounded)
            _reduce_tables.clear()
        ent = _reduce_tables[key] = (host.to(device), at)          # (pageable H2D: synchronous, once per table)
    table, total = ent
    side = _side_streams[device]
    with torch.cuda.stream(side):
        hip.check(hip.lib().ym_wgrad_reduce_batch(ctypes.c_void_p(table.data_ptr()), len(items), total, hip.stream_ptr()),
                  'ym_wgrad_reduce_batch')
    wgrad_reduce_launches[0] += 1
    wgrad_reduce_launches[1] += len(items)


def _conv_wgrad(x, dz, weight_shape, stride, pad, weight_param=None, dw=None, accumulate=False, segments=None, owners=None):
    """`owners`: the parameters a caller-owned `dw` (+ `segments` destinations) belong to (default: `weight_param`)."""
    _auto_backward_begin(weight_param if weight_param is not None else (owners[0] if owners else None))
    if not _side_active[0] or not x.is_cuda:
        return _conv_wgrad_now(x, dz, weight_shape, stride, pad, weight_param, dw, accumulate, segments)
    cout, cin, kh, kw = weight_shape
    shared_dst = dw is not None                      # a caller-owned destination may be written by several launches: keep them ordered
    if dw is None:                                   # (allocated / adopted on the main stream, written on the side stream)
        dw = _grad_slot(weight_param, (cout, cin, kh, kw)) if weight_param is not None else \
            torch.empty(cout, cin, kh, kw, device=x.device, dtype=torch.float32)
    owners = owners if owners is not None else (weight_param,)
    dsts = (dw,) if segments is None else (dw, segments[2], segments[3])
    if len(owners) != len(dsts) or not all(_side_ok(o, d) for o, d in zip(owners, dsts)):
        # a fresh tensor (second use of a parameter in this step, a non-leaf weight, no flat optimizer): autograd reads it on the
        # main stream right after this node returns -> produce it there
        return _conv_wgrad_now(x, dz, weight_shape, stride, pad, weight_param, dw, accumulate, segments)
    side = wgrad_stream(x.device, ordered=shared_dst)
    side.wait_stream(torch.cuda.current_stream(x.device))          # x, dz (and earlier accumulations into dw) are ready
    # (a caller-owned destination is accumulated into by several launches of one step: those reduce at once, in order)
    defer = _REDUCE_BATCH > 0 and _N_SIDE <= 1 and not shared_dst and segments is None and not accumulate
    with torch.cuda.stream(side):
        _conv_wgrad_now(x, dz, weight_shape, stride, pad, weight_param, dw, accumulate, segments, defer=defer)
    if defer and len(_pending_reduce.get(_dev_key(x.device), ())) >= _REDUCE_BATCH:
        flush_wgrad_reduces(x.device)
    _keep_for_side(x.device, side, x, dz)                          # neither recycled nor summed into in place until the side stream is past
    for o in owners:
        getattr(o, '_ym_owner', o)._ym_side_written = True
    return dw


def _conv_wgrad_now(x, dz, weight_shape, stride, pad, weight_param=None, dw=None, accumulate=False, segments=None, defer=False):
    """`dw`: destination OIHW tensor (default: the parameter's gradient slot / a fresh tensor).  `accumulate`: dw += gradient.
    `segments` = (row_end0, row_end1, dw1, dw2): output channels [0,row_end0) -> dw, [row_end0,row_end1) -> dw1, the rest -> dw2.
    `defer`: slab pass only, into this layer's private scratch; the reduction joins the next `flush_wgrad_reduces`."""
    cout, cin, kh, kw = weight_shape
    b, h, w, cin_p = x.shape
    if dw is None:
        dw = _grad_slot(weight_param, (cout, cin, kh, kw)) if weight_param is not None else \
            torch.empty(cout, cin, kh, kw, device=x.device, dtype=torch.float32)
    key = ('w', x.device.index, b, h, w, cin_p, cin, dz.shape[1], dz.shape[2], dz.shape[3], cout, kh, kw, stride, pad,
           None if segments is None else (segments[0], segments[1]))
    ent = _desc_cache.get(key)
    if ent is None:
        d = WgradDesc()
        d.x, d.dy, d.dw = x.data_ptr(), dz.data_ptr(), dw.data_ptr()
        d.B, d.H, d.W, d.Cin, d.Cin_real, d.Cout, d.Cout_real = b, h, w, cin_p, cin, dz.shape[3], cout
        d.KH, d.KW, d.stride, d.pad, d.Ho, d.Wo, d.msplit = kh, kw, stride, pad, dz.shape[1], dz.shape[2], 0
        if segments is not None:
            d.row_end[0], d.row_end[1] = segments[0], segments[1]
            d.dw_seg[0], d.dw_seg[1] = segments[2].data_ptr(), segments[3].data_ptr()
        _configure_wgrad(d, f'W_M{b * dz.shape[1] * dz.shape[2]}_N{dz.shape[3]}_C{cin_p}_k{kh}_s{stride}')
        nbytes = hip.lib().ym_conv2d_wgrad_workspace_bytes(ctypes.byref(d))
        if nbytes == 0:
            raise RuntimeError('ym_conv2d_wgrad_workspace_bytes: ' + hip.lib().ym_last_error().decode())
        ent = _desc_cache[key] = (d, nbytes)
    d, nbytes = ent
    d.x, d.dy, d.dw = x.data_ptr(), dz.data_ptr(), dw.data_ptr()
    d.accumulate = int(accumulate)
    _count(key)
    if segments is not None:
        d.dw_seg[0], d.dw_seg[1] = segments[2].data_ptr(), segments[3].data_ptr()
    if defer:
        dev = _dev_key(x.device)
        akey = (dev, dw.data_ptr())                      # per DESTINATION (= per layer: layers of one shape share `key`)
        ws = _slab_arena.get(akey)
        if ws is None or ws.numel() < nbytes:
            ws = _slab_arena[akey] = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
        item = hip.WgradReduceItem()
        hip.check(hip.lib().ym_conv2d_wgrad_slabs(ctypes.byref(d), ctypes.c_void_p(ws.data_ptr()), ws.numel(), ctypes.byref(item),
                                                  hip.stream_ptr()), 'ym_conv2d_wgrad_slabs')
        _pending_reduce.setdefault(dev, []).append(item)
        return dw
    ws = scratch(x.device, nbytes)
    hip.check(hip.lib().ym_conv2d_wgrad(ctypes.byref(d), ctypes.c_void_p(ws.data_ptr()), ws.numel(), hip.stream_ptr()),
              'ym_conv2d_wgrad')
    return dw


class ConvBias(torch.autograd.Function):
    """y = act(conv(x, W) + b) on NHWC; output channels zero-padded to `cout_pad` (multiple of 32 when a data
    gradient is needed).  Replaces conv+bias+ReLU/tanh of FPN / ProtoNet / head (modules/yolact.py:18-30,37-47,62-68)."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride, pad, act, cout_pad, residual=None, xjoin=None, xrole=None, producer=None):
        ctx.xjoin, ctx.xrole = xjoin, xrole
        ctx.producer = producer          # BnGradLink of the ConvBn that produced x, when this conv's dgrad writes x's WHOLE gradient
        cout, cin, kh, kw = weight.shape
        wp, k_pad = _pack_fwd(weight, x.shape[-1], cout_pad)
        shift = None
        if bias is not None:
            shift = bias.detach() if cout_pad == cout else F.pad(bias.detach(), (0, cout_pad - cout))
            shift = shift.contiguous()
        y = _conv_forward(x, wp, k_pad, cout_pad, kh, kw, stride, pad, shift, act,
                          residual.contiguous() if residual is not None else None)
        ctx.save_for_backward(x, weight, y if act != ACT_NONE else None)
        ctx.meta = (stride, pad, act, cout_pad, bias is not None, residual is not None)
        ctx.bias_param = bias if isinstance(bias, torch.nn.Parameter) else None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, y = ctx.saved_tensors
        stride, pad, act, cout_pad, has_bias, has_res = ctx.meta
        dy = dy.contiguous()
        m, c = dy.numel() // cout_pad, cout_pad
        dz = torch.empty_like(dy) if act != ACT_NONE else dy
        bias_param = ctx.bias_param
        dbias = None
        if has_bias:
            dbias = _grad_slot(bias_param, (c,)) if (bias_param is not None and cout_pad == weight.shape[0]) else \
                torch.empty(c, device=dy.device, dtype=torch.float32)
        y_ptr = hip.ptr(y) if y is not None else None

        def bias_grad():            # column sums of dy * act'(y) (the reduction re-derives act' itself: it does not need dz)
            ws = scratch(dy.device, hip.lib().ym_bn_train_bwd_workspace_bytes(m, c))                  # partial-sum path
            hip.check(hip.lib().ym_act_bias_bwd(hip.ptr(dy), y_ptr, m, c, act, None, hip.ptr(dbias), ctypes.c_void_p(ws.data_ptr()),
                                                ws.numel(), hip.stream_ptr()), 'ym_act_bias_bwd')
        if has_bias and cout_pad == weight.shape[0] and _side_ok(bias_param, dbias):
            # like the weight gradient, the bias gradient is off the critical path: side stream (it lands in the optimizer's slot)
            side = wgrad_stream(dy.device)
            side.wait_stream(torch.cuda.current_stream(dy.device))
            with torch.cuda.stream(side):
                bias_grad()
            _keep_for_side(dy.device, side, dy, y)
            bias_param._ym_side_written = True
        elif has_bias:
            bias_grad()
        if act != ACT_NONE:
            hip.check(hip.lib().ym_act_bias_bwd(hip.ptr(dy), y_ptr, m, c, act, hip.ptr(dz), None, None, 0, hip.stream_ptr()),
                      'ym_act_bias_bwd')
        dw = _conv_wgrad(x, dz, weight.shape, stride, pad, weight)          # (side stream: overlaps the data gradient below)
        dx = None
        if ctx.needs_input_grad[0]:
            add = _join_add(ctx.xjoin)
            dx = _join_result(ctx.xjoin, ctx.xrole, _conv_dgrad(dz, weight, cout_pad, x.shape, stride, pad, add, bn_bwd=ctx.producer))
        if dbias is not None and cout_pad != weight.shape[0]:
            dbias = dbias[:weight.shape[0]].contiguous()
        return dx, dw, dbias, None, None, None, None, (dz if has_res else None), None, None, None


class _HeadWeights:
    """The three sibling convs of the PredictionModule (conf | bbox | coef, modules/yolact.py:22-24) run as ONE 351-channel conv.
    Their OIHW weights / biases are kept concatenated in persistent buffers that follow the parameters: three contiguous
    device-to-device copies per optimizer step (before the batched re-pack of the weight images) instead of a `torch.cat` with
    its autograd node, allocation and backward split every forward."""
    _by_module = weakref.WeakKeyDictionary()      # PredictionModule -> its _HeadWeights: a rebuilt / collected net leaves nothing behind

    def __init__(self, hd, pad):
        convs = (hd.conf_layer, hd.bbox_layer, hd.coef_layer[0])
        self.convs = convs
        w0 = convs[0].weight
        self.couts = [c.out_channels for c in convs]
        self.cout = sum(self.couts)
        self.w = torch.zeros(self.cout, *w0.shape[1:], device=w0.device, dtype=torch.float32)
        self.b = torch.zeros(pad, device=w0.device, dtype=torch.float32)
        self.w._ym_grad_slot = self.w           # marks the buffer as "refreshed only by its owner" for the pack cache
        self.stamp = None
        _pre_refresh_hooks.append(weakref.ref(self))

    @classmethod
    def of(cls, hd, pad):
        e = cls._by_module.get(hd)
        if e is None or e.convs[0].weight.device != e.w.device or e.convs[0] is not hd.conf_layer:
            e = cls._by_module[hd] = cls(hd, pad)
        return e

    def sync(self):
        stamp = tuple(c.weight._version for c in self.convs) + tuple(c.bias._version for c in self.convs) + (_EPOCH[0],)
        if stamp == self.stamp:
            return
        with torch.no_grad():
            r = 0
            for c, n in zip(self.convs, self.couts):
                self.w[r:r + n].copy_(c.weight.detach())
                self.b[r:r + n].copy_(c.bias.detach())
                r += n
        self.stamp = stamp


class PredictionHead(torch.autograd.Function):
    """The shared PredictionModule over all FPN levels as ONE autograd node (modules/yolact.py:26-31,149-157): per level the
    upfeature conv (+ReLU) and one 351-channel conv whose segmented epilogue writes conf / box / tanh(coef) straight into the
    concatenated [B, N, *] tensors (no permute / reshape / cat / tanh kernels).  Backward: one gather turns the loss's gradients
    of those tensors into the per-level [rows][352] conv-output gradients (tanh' included), one column-sum pair gives the two
    bias gradients for all levels, and the per-level weight-gradient launches ACCUMULATE into the parameters' gradient slots
    (the reference's autograd sums five per-level gradients with separate kernels)."""

    @staticmethod
    def forward(ctx, w_up, b_up, w_conf, b_conf, w_box, b_box, w_coef, b_coef, hd, nc, cd, na, joins, *levels):
        ctx.joins = joins if joins is not None else (None,) * len(levels)
        dev = levels[0].device
        bsz = levels[0].shape[0]
        shapes = [(lv.shape[1], lv.shape[2]) for lv in levels]
        row_off, anc_off = [0], [0]
        for h, w in shapes:
            row_off.append(row_off[-1] + bsz * h * w)
            anc_off.append(anc_off[-1] + h * w * na)
        n_total = anc_off[-1]
        c_conf, c_box, c_coef = na * nc, na * 4, na * cd
        cout = c_conf + c_box + c_coef
        pad = _ru(cout, 32)
        hw_ = _HeadWeights.of(hd, pad)
        hw_.sync()
        wp_up, k_up = _pack_fwd(w_up, 256, 256)
        wp_hd, k_hd = _pack_fwd(hw_.w, 256, pad)
        xh_all = torch.empty(row_off[-1], 256, device=dev, dtype=torch.float32)
        conf = torch.empty(bsz, n_total, nc, device=dev, dtype=torch.float32)
        box = torch.empty(bsz, n_total, 4, device=dev, dtype=torch.float32)
        coef = torch.empty(bsz, n_total, cd, device=dev, dtype=torch.float32)
        b_up_c = b_up.detach().contiguous()
        for l, lv in enumerate(levels):
            h, w = shapes[l]
            xh = xh_all[row_off[l]:row_off[l + 1]].view(bsz, h, w, 256)
            _conv_forward(lv, wp_up, k_up, 256, 3, 3, 1, 1, b_up_c, ACT_RELU, out=xh)
            off = anc_off[l]
            segs = [(0, c_conf, conf.data_ptr() + off * nc * 4, n_total * nc, c_conf, ACT_NONE),
                    (c_conf, c_conf + c_box, box.data_ptr() + off * 4 * 4, n_total * 4, c_box, ACT_NONE),
                    (c_conf + c_box, cout, coef.data_ptr() + off * cd * 4, n_total * cd, c_coef, ACT_TANH)]
            _conv_forward(xh, wp_hd, k_hd, cout, 3, 3, 1, 1, hw_.b, ACT_NONE, segs=segs)
        ctx.save_for_backward(w_up, xh_all, coef, *levels)
        ctx.meta = (hd, nc, cd, na, shapes, row_off, anc_off, pad, b_up, (w_conf, b_conf, w_box, b_box, w_coef, b_coef))
        return conf, box, coef

    @staticmethod
    def backward(ctx, dconf, dbox, dcoef):
        w_up, xh_all, coef, *levels = ctx.saved_tensors
        hd, nc, cd, na, shapes, row_off, anc_off, pad, b_up, (w_conf, b_conf, w_box, b_box, w_coef, b_coef) = ctx.meta
        dev = xh_all.device
        bsz = levels[0].shape[0]
        n_total = anc_off[-1]
        mtot = row_off[-1]
        c_conf, c_box, c_coef = na * nc, na * 4, na * cd
        hw_ = _HeadWeights.of(hd, pad)
        L = hip.lib()
        nlev = len(levels)
        rows_c = (ctypes.c_int32 * (nlev + 1))(*row_off)
        ancs_c = (ctypes.c_int32 * (nlev + 1))(*anc_off)
        dz_all = torch.empty(mtot, pad, device=dev, dtype=torch.float32)
        hip.check(L.ym_head_grad_gather(hip.ptr(dconf.contiguous()), hip.ptr(dbox.contiguous()), hip.ptr(dcoef.contiguous()), hip.ptr(coef),
                                        bsz, n_total, nc, cd, na, nlev, rows_c, ancs_c, pad, None, hip.ptr(dz_all), hip.stream_ptr()),
                  'ym_head_grad_gather')
        # head biases: column sums over all levels' rows, then split into the three parameters' slots
        ws = scratch(dev, L.ym_bn_train_bwd_workspace_bytes(mtot, pad))
        db_cat = torch.empty(pad, device=dev, dtype=torch.float32)
        hip.check(L.ym_act_bias_bwd(hip.ptr(dz_all), None, mtot, pad, ACT_NONE, None, hip.ptr(db_cat), ctypes.c_void_p(ws.data_ptr()),
                                    ws.numel(), hip.stream_ptr()), 'ym_act_bias_bwd')
        db = [_grad_slot(b_conf, (c_conf,)), _grad_slot(b_box, (c_box,)), _grad_slot(b_coef, (c_coef,))]
        hip.check(L.ym_scatter3(hip.ptr(db_cat), hip.ptr(db[0]), c_conf, hip.ptr(db[1]), c_box, hip.ptr(db[2]), c_coef, 0,
                                hip.stream_ptr()), 'ym_scatter3')
        dw = [_grad_slot(w_conf, tuple(w_conf.shape)), _grad_slot(w_box, tuple(w_box.shape)), _grad_slot(w_coef, tuple(w_coef.shape))]
        dxh_all = torch.empty(mtot, 256, device=dev, dtype=torch.float32)
        for l in range(nlev):
            h, w = shapes[l]
            dz = dz_all[row_off[l]:row_off[l + 1]].view(bsz, h, w, pad)
            xh = xh_all[row_off[l]:row_off[l + 1]].view(bsz, h, w, 256)
            _conv_dgrad(dz, hw_.w, pad, (bsz, h, w, 256), 1, 1, out=dxh_all[row_off[l]:row_off[l + 1]].view(bsz, h, w, 256))
            _conv_wgrad(xh, dz, tuple(hw_.w.shape), 1, 1, dw=dw[0], accumulate=l > 0, segments=(c_conf, c_conf + c_box, dw[1], dw[2]),
                        owners=(w_conf, w_box, w_coef))
        # upfeature: ReLU backward + bias column sums for all levels at once, then per-level data / weight gradients
        dzu_all = torch.empty_like(dxh_all)
        db_up = _grad_slot(b_up, (256,))
        ws = scratch(dev, L.ym_bn_train_bwd_workspace_bytes(mtot, 256))
        hip.check(L.ym_act_bias_bwd(hip.ptr(dxh_all), hip.ptr(xh_all), mtot, 256, ACT_RELU, hip.ptr(dzu_all), hip.ptr(db_up),
                                    ctypes.c_void_p(ws.data_ptr()), ws.numel(), hip.stream_ptr()), 'ym_act_bias_bwd')
        dw_up = _grad_slot(w_up, tuple(w_up.shape))
        dlevels = []
        for l, lv in enumerate(levels):
            h, w = shapes[l]
            dzu = dzu_all[row_off[l]:row_off[l + 1]].view(bsz, h, w, 256)
            if ctx.needs_input_grad[13 + l]:
                # (a level with other consumers -- P3: protonet + semantic conv, P5 / P6: the stride-2 convs -- hands its gradient on
                # through the level's GradJoin instead of returning it: no autograd sum)
                j = ctx.joins[l]
                dlevels.append(_join_result(j, 'pass' if j is not None else None,
                                            _conv_dgrad(dzu, w_up, 256, lv.shape, 1, 1, _join_add(j))))
            else:
                dlevels.append(None)
            _conv_wgrad(lv, dzu, tuple(w_up.shape), 1, 1, dw=dw_up, accumulate=l > 0, owners=(w_up,))
        return (dw_up, db_up, dw[0], db[0], dw[1], db[1], dw[2], db[2], None, None, None, None, None, *dlevels)


# Every link of the current forward (ResGradLink / GradJoin): `check_links_drained()` after backward proves that no parked gradient
# was left behind -- i.e. that every taker really ran AFTER its givers (the order autograd's engine is relied on for).
_live_links = []


def check_links_drained():
    """Call after `backward()` (Trainer.step does): a gradient still parked in a link was never added by its taker -- the taker ran
    before a giver, or never ran -- and would be silently MISSING from the step.  Raises instead."""
    left = [type(l).__name__ for l in _live_links if l.grad is not None]
    for l in _live_links:
        l.grad = None
    _live_links.clear()
    if left:
        raise RuntimeError(f'{len(left)} gradient(s) parked in {sorted(set(left))} were never consumed: autograd did not run the '
                           f'consumers of a shared tensor in decreasing creation order (set YM_GRAD_JOIN=0 YM_FUSE_RES_GRAD=0)')


class ResGradLink:
    """A Bottleneck's input feeds conv1 AND the residual add of conv3 (modules/resnet.py:21,35-37); autograd would sum the two
    gradients with an extra elementwise pass.  conv3's backward parks its residual gradient here (`give`) and conv1's backward,
    which necessarily runs later, adds it in the epilogue of its data-gradient conv (`take`)."""
    __slots__ = ('grad',)

    def __init__(self):
        self.grad = None
        _live_links.append(self)


class GradJoin:
    """A tensor with several consumers (a stage input feeds conv1 and the downsample conv, C3 / C4 also the FPN lateral conv; P3 feeds
    the protonet, the prediction head and the semantic conv; ...): autograd would sum the consumers' gradients with one elementwise
    kernel per extra consumer (ATen's, 14 per res101 step).  Instead the consumers form a chain through this object: autograd runs
    the nodes of one device in strictly decreasing creation order, so the consumer created LAST runs its backward FIRST; every
    consumer but the first-created one has the role 'pass' (its data-gradient launch adds what is parked here in its epilogue,
    parks the sum here and returns None), and the first-created one 'take's: its launch adds the parked sum and returns the WHOLE
    gradient of the tensor -- which also lets it carry the BatchNorm-backward sums of the tensor's producer (`sole_grad`).
    Same summation order as autograd's (arrival order).  ResGradLink is the two-consumer special case where the giver is a
    residual add rather than a convolution."""
    __slots__ = ('grad',)

    def __init__(self):
        self.grad = None
        _live_links.append(self)


_GRAD_JOIN = os.environ.get('YM_GRAD_JOIN', '1') != '0'
grad_join_passes = [0]           # gradients handed on through a GradJoin instead of an autograd sum (tests look at it)


def _join_add(join):
    """What the earlier-run consumers of the tensor parked (None for the first one to run)."""
    if join is None:
        return None
    add, join.grad = join.grad, None
    return add


_drain_armed = [False]


def _arm_drain_check():
    """Every backward that parks a gradient ends with `check_links_drained()`, whoever called `.backward()`: a callback queued on the
    running autograd engine (it fires when this backward's graph task has finished and raises out of `.backward()` /
    `autograd.grad`).  A PARTIAL backward — `loss_s.backward()`, `torch.autograd.grad` of one loss term, one backward per loss —
    never reaches the 'take' consumer of a tensor whose 'pass' consumers did run; without this the parked gradient was dropped in
    silence outside `Trainer.step`.  For partial backward set YM_GRAD_JOIN=0 YM_FUSE_RES_GRAD=0 (autograd's own sums)."""
    if _drain_armed[0]:
        return

    def _cb():
        _drain_armed[0] = False
        check_links_drained()
    try:
        torch.autograd.Variable._execution_engine.queue_callback(_cb)
        _drain_armed[0] = True
    except RuntimeError:                 # not inside a backward pass (a unit test calling the helpers directly)
        pass


def _join_result(join, role, dx):
    """'pass': park the running sum and return nothing to autograd; 'take' / no join: the gradient itself."""
    if join is not None and role == 'pass':
        join.grad = dx
        grad_join_passes[0] += 1
        _arm_drain_check()
        return None
    return dx


class BnGradLink:
    """Joins a ConvBn with the ONE backward launch that writes the gradient of its output (the data-gradient conv of its only
    consumer, or of the consumer that collects every contribution: ResGradLink 'take').  Forward fills in where the BN's saved
    tensors live (raw pointers: the tensors themselves are owned by the node's saved_tensors, which outlive the consumer's
    backward); the consumer's backward leaves the two fp64 column sums in `stats` and the address of the gradient it wrote in
    `dout_ptr`, and the BN's own backward skips its statistics pass when that is the gradient it was handed."""
    __slots__ = ('y', 'out', 'mean', 'invstd', 'gamma', 'beta', 'relu', 'c', 'm', 'stats', 'dout_ptr')

    def __init__(self):
        self.stats = self.dout_ptr = None


_FUSE_BN_BWD = os.environ.get('YM_FUSE_BN_BWD', '1') != '0'
bn_bwd_fused_launches = [0]      # BN backward passes that skipped their statistics pass (tests / tools look at it)


class ConvBn(torch.autograd.Function):
    """out = relu?(BN_train(conv(x, W)) + residual?) — one Bottleneck stage (modules/resnet.py:23-38) in train mode."""

    @staticmethod
    def forward(ctx, x, weight, gamma, beta, running_mean, running_var, residual, stride, pad, relu, momentum, eps, link=None,
                role=None, producer=None, own=None, xjoin=None, xrole=None):
        ctx.link, ctx.role = link, role
        ctx.xjoin, ctx.xrole = xjoin, xrole
        ctx.producer, ctx.own = producer, own
        cout, cin, kh, kw = weight.shape
        wp, k_pad = _pack_fwd(weight, x.shape[-1], cout)
        stats = _stats_pool.take(2 * cout, x.device)           # zeroed (the fused epilogue accumulates into it)
        y, fused = _conv_forward(x, wp, k_pad, cout, kh, kw, stride, pad, None, ACT_NONE, bn_stats=stats)
        m = y.numel() // cout
        out = torch.empty_like(y)
        mean = torch.empty(cout, device=x.device, dtype=torch.float32)
        invstd = torch.empty_like(mean)
        res_ptr = hip.ptr(residual.contiguous()) if residual is not None else None
        if fused:
            hip.check(hip.lib().ym_bn_train_fwd_stats(hip.ptr(y), m, cout, hip.ptr(gamma.detach()), hip.ptr(beta.detach()), eps,
                                                      momentum, hip.ptr(running_mean), hip.ptr(running_var), res_ptr, int(relu),
                                                      hip.ptr(out), hip.ptr(mean), hip.ptr(invstd),
                                                      ctypes.c_void_p(stats.data_ptr()), hip.stream_ptr()), 'ym_bn_train_fwd_stats')
        else:
            hip.check(hip.lib().ym_bn_train_fwd(hip.ptr(y), m, cout, hip.ptr(gamma.detach()), hip.ptr(beta.detach()), eps, momentum,
                                                hip.ptr(running_mean), hip.ptr(running_var), res_ptr, int(relu),
                                                hip.ptr(out), hip.ptr(mean), hip.ptr(invstd), ctypes.c_void_p(stats.data_ptr()),
                                                stats.numel() * 8, hip.stream_ptr()), 'ym_bn_train_fwd')
        # `out` is only needed for the ReLU mask when a residual was added; otherwise backward re-derives the mask from y
        mask_out = out if (relu and (residual is not None or beta is None)) else None
        ctx.save_for_backward(x, weight, gamma, y, mask_out, mean, invstd)
        ctx.meta = (stride, pad, relu, residual is not None)
        ctx.beta_param = beta
        if own is not None:
            own.y, own.out = y.data_ptr(), (mask_out.data_ptr() if mask_out is not None else None)
            own.mean, own.invstd, own.gamma = mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr()
            own.beta, own.relu, own.c, own.m = (beta.data_ptr() if beta is not None else None), int(relu), cout, m
        return out

    @staticmethod
    def backward(ctx, dout):
        x, weight, gamma, y, out, mean, invstd = ctx.saved_tensors
        stride, pad, relu, has_res = ctx.meta
        dout = dout.contiguous()
        cout = weight.shape[0]
        m = y.numel() // cout
        dy = torch.empty_like(y)
        dres = torch.empty_like(y) if has_res else None
        dgamma = _grad_slot(gamma, (cout,))
        dbeta = _grad_slot(ctx.beta_param, (cout,)) if ctx.beta_param is not None else torch.empty(cout, device=y.device)
        own = ctx.own
        out_ptr = hip.ptr(out) if (relu and out is not None) else None
        beta_ptr = hip.ptr(ctx.beta_param.detach()) if ctx.beta_param is not None else None
        if own is not None and own.stats is not None and own.dout_ptr == dout.data_ptr():
            # the launch that wrote `dout` already summed dz and dz * xhat over the pixels (BnGradLink): apply pass only
            hip.check(hip.lib().ym_bn_train_bwd_apply(hip.ptr(dout), out_ptr, hip.ptr(y), m, cout, hip.ptr(gamma.detach()), beta_ptr,
                                                      hip.ptr(mean), hip.ptr(invstd), int(relu), hip.ptr(dy),
                                                      hip.ptr(dres) if has_res else None, hip.ptr(dgamma), hip.ptr(dbeta),
                                                      ctypes.c_void_p(own.stats.data_ptr()), hip.stream_ptr()), 'ym_bn_train_bwd_apply')
            bn_bwd_fused_launches[0] += 1
        else:
            ws = scratch(y.device, hip.lib().ym_bn_train_bwd_workspace_bytes(m, cout))
            hip.check(hip.lib().ym_bn_train_bwd(hip.ptr(dout), out_ptr, hip.ptr(y), m, cout, hip.ptr(gamma.detach()), beta_ptr,
                                                hip.ptr(mean), hip.ptr(invstd), int(relu), hip.ptr(dy),
                                                hip.ptr(dres) if has_res else None, hip.ptr(dgamma), hip.ptr(dbeta),
                                                ctypes.c_void_p(ws.data_ptr()), ws.numel(), hip.stream_ptr()), 'ym_bn_train_bwd')
        if own is not None:
            own.stats = own.dout_ptr = None
        need_dx = ctx.needs_input_grad[0]
        if need_dx and cout % 32 != 0:
            raise RuntimeError('ConvBn dgrad needs Cout % 32 == 0')
        add = None
        if ctx.link is not None:
            if ctx.role == 'give' and has_res:
                ctx.link.grad, dres = dres, None                  # handed to the consumer that shares the tensor
                _arm_drain_check()
            elif ctx.role == 'take' and need_dx:
                add, ctx.link.grad = ctx.link.grad, None
        if need_dx and ctx.xjoin is not None:
            assert add is None
            add = _join_add(ctx.xjoin)
        dw = _conv_wgrad(x, dy, weight.shape, stride, pad, weight)          # (side stream: overlaps the data gradient below)
        dx = _conv_dgrad(dy, weight, cout, x.shape, stride, pad, add, bn_bwd=ctx.producer) if need_dx else None
        if need_dx:
            dx = _join_result(ctx.xjoin, ctx.xrole, dx)
        return dx, dw, dgamma, dbeta, None, None, dres, None, None, None, None, None, None, None, None, None, None, None


class MaxPool(torch.autograd.Function):
    """MaxPool2d(3, 2, 1) (modules/resnet.py:91).  The forward keeps the argmax position of every window (one byte per output
    element) so that backward is a gather from it instead of re-scanning the windows of x."""

    @staticmethod
    def forward(ctx, x):
        b, h, w, c = x.shape
        ho, wo = (h + 2 - 3) // 2 + 1, (w + 2 - 3) // 2 + 1
        out = torch.empty(b, ho, wo, c, device=x.device, dtype=torch.float32)
        idx = torch.empty(b, ho, wo, c, device=x.device, dtype=torch.uint8)
        hip.check(hip.lib().ym_maxpool3x3s2_fwd_idx(hip.ptr(x.contiguous()), hip.ptr(out), hip.ptr(idx, torch.uint8), b, h, w, c,
                                                     hip.stream_ptr()), 'ym_maxpool3x3s2_fwd_idx')
        ctx.save_for_backward(idx)
        ctx.shape = (b, h, w, c)
        return out

    @staticmethod
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        b, h, w, c = ctx.shape
        dx = torch.empty(b, h, w, c, device=dy.device, dtype=torch.float32)
        hip.check(hip.lib().ym_maxpool3x3s2_bwd_idx(hip.ptr(idx, torch.uint8), hip.ptr(dy.contiguous()), hip.ptr(dx), b, h, w, c,
                                                     hip.stream_ptr()), 'ym_maxpool3x3s2_bwd_idx')
        return dx


class Bilinear2x(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, align, xjoin=None):
        ctx.xjoin = xjoin                 # x has an earlier-created consumer that 'take's (GradJoin): this node passes its gradient on
        b, h, w, c = x.shape
        out = torch.empty(b, 2 * h, 2 * w, c, device=x.device, dtype=torch.float32)
        hip.bilinear2x(x, out, align)
        ctx.meta = (x.shape, align)
        return out

    @staticmethod
    def backward(ctx, dy):
        (b, h, w, c), align = ctx.meta
        dx = torch.empty(b, h, w, c, device=dy.device, dtype=torch.float32)
        hip.check(hip.lib().ym_bilinear2x_bwd(hip.ptr(dy.contiguous()), hip.ptr(dx), b, h, w, c, int(align), hip.stream_ptr()),
                  'ym_bilinear2x_bwd')
        if ctx.xjoin is not None:
            assert ctx.xjoin.grad is None, 'Bilinear2x must be the last-created consumer of its input'
            return _join_result(ctx.xjoin, 'pass', dx), None, None
        return dx, None, None


def _conv_bn(x, conv, bn, relu=True, residual=None, link=None, role=None, sole_grad=False, xjoin=None, xrole=None):
    """`sole_grad`: this conv's data-gradient launch writes the WHOLE gradient of x (x has no other consumer, or this is the
    ResGradLink / GradJoin 'take' end that folds the others in) -> it may carry the backward statistics of the BN that produced x.
    `xjoin`, `xrole`: x has several consumers, chained through a GradJoin (this one 'take's or 'pass'es)."""
    producer = getattr(x, '_ym_bn_link', None) if (sole_grad and _FUSE_BN_BWD) else None
    own = BnGradLink() if _FUSE_BN_BWD else None
    out = ConvBn.apply(x, conv.weight, bn.weight, bn.bias, bn.running_mean, bn.running_var, residual, conv.stride[0],
                       conv.padding[0], relu, float(bn.momentum), float(bn.eps), link, role, producer, own, xjoin, xrole)
    if own is not None:
        out._ym_bn_link = own
    if not getattr(bn, '_ym_nbt_flat', False):               # the trainer bumps all counters with one launch
        bn.num_batches_tracked += 1
    return out


def _conv_bias(x, conv, act=ACT_NONE, cout_pad=None, residual=None, xjoin=None, xrole=None, sole_grad=False):
    """`sole_grad`: as in `_conv_bn` (C5's only consumer is the FPN lateral conv: its dgrad carries layer4's last bn3 sums)."""
    cout = conv.out_channels
    producer = getattr(x, '_ym_bn_link', None) if (sole_grad and _FUSE_BN_BWD) else None
    return ConvBias.apply(x, conv.weight, conv.bias, conv.stride[0], conv.padding[0], act, cout_pad or cout, residual, xjoin, xrole,
                          producer)


_FUSE_RES_GRAD = os.environ.get('YM_FUSE_RES_GRAD', '1') != '0'


def train_features(net, img):
    """Train-mode network forward (HIP kernels), returns (class logits [B,N,C], box [B,N,4], coef [B,N,32] (tanh),
    proto [B,Hp,Wp,32], seg logits [B,C-1,H3,W3]) with an autograd tape attached."""
    b, _, h, w = img.shape
    x = torch.empty(b, h, w, 4, device=img.device, dtype=torch.float32)
    hip.nchw_to_nhwc4(img.contiguous().float(), x)
    _stats_pool.begin(img.device)
    _live_links.clear()                                          # (links of a forward whose backward never ran)
    _drain_armed[0] = False                                      # (a backward that died before its callbacks ran)
    if _auto:                                                    # (likewise: its end-of-backward callback never ran)
        _side_active[0] = _auto_prev_side[0]
        del _auto[:]
    bb = net.backbone
    if hasattr(bb, 'patch_embed'):                                # Swin-T (modules/swin_transformer.py)
        from .swin_train import swin_backbone_train
        c3, c4, c5 = swin_backbone_train(bb, x, net.training)
    else:
        x = _conv_bn(x, bb.conv1, bb.bn1)
        x = MaxPool.apply(x)
        outs = []
        for stage in bb.layers:
            for blk in stage:
                # identity blocks: x feeds conv1 and the residual add; their two gradients meet in conv1's dgrad epilogue
                link = ResGradLink() if (blk.downsample is None and x.requires_grad and _FUSE_RES_GRAD) else None
                # a stage's first block: x feeds conv1 and the downsample conv (C3 / C4 later also the FPN lateral conv): GradJoin,
                # conv1 'take's.  Either way conv1's dgrad writes the whole gradient of x and carries the previous bn3's backward sums
                xj = GradJoin() if (blk.downsample is not None and x.requires_grad and _GRAD_JOIN) else None
                if xj is not None:
                    x._ym_join = xj
                y = _conv_bn(x, blk.conv1, blk.bn1, link=link, role='take', sole_grad=link is not None or xj is not None,
                             xjoin=xj, xrole='take')
                y = _conv_bn(y, blk.conv2, blk.bn2, sole_grad=True)
                skip = _conv_bn(x, blk.downsample[0], blk.downsample[1], relu=False, xjoin=xj, xrole='pass') \
                    if blk.downsample is not None else x
                x = _conv_bn(y, blk.conv3, blk.bn3, relu=True, residual=skip, link=link, role='give', sole_grad=True)
            outs.append(x)
        c3, c4, c5 = outs[1:4]

    def join_of(t):                  # the GradJoin a backbone tensor already has (C3 / C4: the next stage's conv1 takes), for a 'pass'
        return getattr(t, '_ym_join', None)

    def new_join(t):
        return GradJoin() if (_GRAD_JOIN and t.requires_grad) else None

    # Consumers of a shared tensor are created taker first (see GradJoin): each pred conv before the upsample of its input, the
    # protonet / stride-2 convs before the prediction head.
    fpn = net.fpn
    p5_1 = _conv_bias(c5, fpn.lat_layers[2], sole_grad=True)
    j = new_join(p5_1)
    p5 = _conv_bias(p5_1, fpn.pred_layers[2][0], ACT_RELU, xjoin=j, xrole='take')
    p4_1 = _conv_bias(c4, fpn.lat_layers[1], residual=Bilinear2x.apply(p5_1, False, j),      # top-down add fused
                      xjoin=join_of(c4), xrole='pass')
    j = new_join(p4_1)
    p4 = _conv_bias(p4_1, fpn.pred_layers[1][0], ACT_RELU, xjoin=j, xrole='take')
    p3_1 = _conv_bias(c3, fpn.lat_layers[0], residual=Bilinear2x.apply(p4_1, False, j), xjoin=join_of(c3), xrole='pass')
    p3 = _conv_bias(p3_1, fpn.pred_layers[0][0], ACT_RELU)
    j5, j6, j3 = new_join(p5), None, new_join(p3)
    p6 = _conv_bias(p5, fpn.downsample_layers[0][0], ACT_RELU, xjoin=j5, xrole='take')
    j6 = new_join(p6)
    p7 = _conv_bias(p6, fpn.downsample_layers[1][0], ACT_RELU, xjoin=j6, xrole='take')
    levels = [p3, p4, p5, p6, p7]
    level_joins = (j3, None, j5, j6, None)

    pn = net.proto_net
    y = p3
    for i in (0, 2, 4):
        y = _conv_bias(y, pn.proto1[i], ACT_RELU, xjoin=j3 if i == 0 else None, xrole='take')
    y = Bilinear2x.apply(y, True)
    y = _conv_bias(y, pn.proto2[0], ACT_RELU)
    proto = _conv_bias(y, pn.proto2[2], ACT_RELU)                 # NHWC [B,Hp,Wp,32] == proto_out layout

    hd = net.prediction_layers
    nc, cd, na = net.cfg.num_classes, net.coef_dim, len(net.cfg.aspect_ratios)
    if os.environ.get('YM_FUSED_HEAD', '1') != '0':
        conf_all, box_all, coef_all = PredictionHead.apply(
            hd.upfeature[0].weight, hd.upfeature[0].bias, hd.conf_layer.weight, hd.conf_layer.bias, hd.bbox_layer.weight,
            hd.bbox_layer.bias, hd.coef_layer[0].weight, hd.coef_layer[0].bias, hd, nc, cd, na, level_joins, *levels)
        seg = _conv_bias(p3, net.semantic_seg_conv, ACT_NONE, _ru(nc - 1, 32), xjoin=j3, xrole='pass')[..., :nc - 1].permute(0, 3, 1, 2)
        return conf_all, box_all, coef_all, proto, seg
    c_conf, c_box, c_coef = na * nc, na * 4, na * cd
    w_head = torch.cat([hd.conf_layer.weight, hd.bbox_layer.weight, hd.coef_layer[0].weight], 0)
    b_head = torch.cat([hd.conf_layer.bias, hd.bbox_layer.bias, hd.coef_layer[0].bias], 0)
    head_pad = _ru(c_conf + c_box + c_coef, 32)
    confs, boxes, coefs = [], [], []
    for lv in levels:
        xh = _conv_bias(lv, hd.upfeature[0], ACT_RELU)
        o = ConvBias.apply(xh, w_head, b_head, 1, 1, ACT_NONE, head_pad, None)    # [B,H,W,352]
        bsz = o.shape[0]
        confs.append(o[..., :c_conf].reshape(bsz, -1, nc))
        boxes.append(o[..., c_conf:c_conf + c_box].reshape(bsz, -1, 4))
        coefs.append(torch.tanh(o[..., c_conf + c_box:c_conf + c_box + c_coef]).reshape(bsz, -1, cd))
    seg = _conv_bias(p3, net.semantic_seg_conv, ACT_NONE, _ru(nc - 1, 32))[..., :nc - 1].permute(0, 3, 1, 2)
    return torch.cat(confs, 1), torch.cat(boxes, 1), torch.cat(coefs, 1), proto, seg
