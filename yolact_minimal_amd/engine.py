"""Inference plan for `Yolact.forward` on one MI355X.

Reference path being replaced: `/root/reference/modules/yolact.py:141-164` (backbone `modules/resnet.py:86-98`,
FPN `modules/yolact.py:73-89`, ProtoNet `:49-53`, PredictionModule `:26-31`, cat + softmax `:155-163`).

The plan is built once per (batch, H, W): it owns every activation buffer in HBM (NHWC fp32, never freed
or re-allocated while serving), the packed weight images ([Cout][kh][kw][Cin], BN folded to scale/shift) and
one `ym_conv_desc` per layer, and then replays a fixed list of C-ABI launches on the current stream —
optionally captured once into a hipGraph (`YM_GRAPH=1`, default) so a bs=1 forward is one graph launch
instead of ~150 kernel launches.

Fusions (what never touches HBM as a separate tensor):
  conv + BN + ReLU                  -> one launch          (resnet.py:23-31)
  conv + BN + residual add + ReLU   -> one launch          (resnet.py:33-38)
  conv + bias + ReLU / + top-down add (FPN lateral)        (yolact.py:74-84)
  bbox | conf | coef convs + tanh + permute + reshape + cat over levels -> one 351-channel launch per level
                                     writing straight into the [B, N, C] outputs (yolact.py:27-30,155-157)
"""
import json
import os
import sys

import torch
import torch.nn as nn

from . import hip, plan_transfer
from .hip import ConvDesc, ACT_NONE, ACT_RELU, ACT_TANH, ACT_GELU


def _round_up(x, m):
    return (x + m - 1) // m * m


TUNED_PATH = os.environ.get('YM_TUNED_PATH', os.path.join(os.path.dirname(os.path.abspath(__file__)), 'tuned_gfx950.json'))
_tuned = None


def conv_mma():
    """YM_CONV_MMA = 0 (default: f32 MFMA, the parity mode) | 3 (bf16x3) | 6 (bf16x6): see ym_conv_desc.mma."""
    return int(os.environ.get('YM_CONV_MMA', '0') or 0)


def _grid_wgs(hit):
    """Eighth field of a tuned entry = `grid_wgs` of the persistent kernel.  (Old detail rows of `InferEngine.autotune` carried a
    timing there: anything that is not a non-negative integer is rejected instead of landing in a c_int32 field.)"""
    if len(hit) <= 7:
        return 0
    g = hit[7]
    if isinstance(g, bool) or not isinstance(g, int) or g < 0:
        raise ValueError(f'tuned entry {hit}: field 7 must be grid_wgs (a non-negative integer)')
    if hit[3] > 0 and 22 <= hit[4] <= 24:
        # wave kernel with DMA rings (conv_wdma_f32): the field is WAVES PER WORKGROUP, not a grid size -- a row edited over from
        # the persistent kernel (hundreds of workgroups) would be rejected by the launch, or silently lose its tail split
        if g not in (0, 1, 2, 4) or (g and g < hit[3]) or ((hit[5] or hit[6]) and g not in (0, 4)):
            raise ValueError(f'tuned entry {hit}: field 7 of a wave-DMA row is waves per workgroup (0 / 1 / 2 / 4, >= kwaves; 0 / 4 with a tail)')
    return g


def autotune_on():
    """YM_AUTOTUNE=1: an engine whose plan has launches without a row of their own (another --img_size / batch) measures them on
    first use (InferEngine.autotune, ~0.5 s per distinct shape) and keeps the rows in a per-user cache file that later processes
    read back: the find-db pattern.  Off by default: a first request then never pays for a sweep, and plans do not depend on
    files outside the package."""
    return os.environ.get('YM_AUTOTUNE', '0') == '1'


def user_cache_path():
    return os.environ.get('YM_TUNED_CACHE') or os.path.join(os.path.expanduser('~'), '.cache', 'yolact_minimal_amd', 'tuned_gfx950_user.json')


def tuned_table():
    """Per-shape (tile_m, tile_n, ksplit) choices measured on an MI355X by tools/autotune.py; with YM_AUTOTUNE=1 overlaid by the
    rows this user's earlier runs measured (user_cache_path())."""
    global _tuned
    if _tuned is None:
        _tuned = {}
        if os.path.exists(TUNED_PATH) and os.environ.get('YM_NO_TUNED', '0') != '1':
            with open(TUNED_PATH) as f:
                _tuned = json.load(f)
            if autotune_on() and os.path.exists(user_cache_path()):
                try:
                    with open(user_cache_path()) as f:
                        _tuned.update({k: v for k, v in json.load(f).items() if isinstance(v, list)})
                except (OSError, ValueError) as e:      # a torn / foreign file must not stop inference
                    print(f'yolact_minimal_amd: ignoring {user_cache_path()}: {e}', file=sys.stderr)
    return _tuned


def _store_user_rows(rows):
    """Merge `rows` into the user cache (read-modify-write through a temp file + rename: concurrent ranks lose rows at worst)."""
    path = user_cache_path()
    os.makedirs(os.path.dirname(path), exist_ok=True)
    cur = {}
    if os.path.exists(path):
        try:
            with open(path) as f:
                cur = json.load(f)
        except (OSError, ValueError):
            cur = {}
    cur.update(rows)
    tmp = f'{path}.{os.getpid()}.tmp'
    with open(tmp, 'w') as f:
        json.dump(cur, f, indent=0, sort_keys=True)
    os.replace(tmp, path)


_build_mode = ['latency']


def _entry(sig, M=0, N=0, nkt=0, nseg=1, with_source=False):
    """Tuned entry of a conv shape for the engine being built.  'latency' (one request at a time: the default) reads `sig`;
    'throughput' (the slots of a RequestPipeline with several requests in flight) reads `sig + '_tp'` first: choices that spread a
    launch over every CU (tail splits, one-wave workgroups) shorten a lone request and cost throughput when other requests want
    those CUs -- tools/tune_forward.py --inflight N measures them on the pipeline's own img/s.
    A shape without a row (another --img_size / batch) takes the row of the nearest tuned shape of its family, re-derived for its
    M (plan_transfer.py); `with_source` also returns where the row came from ('table' / 'nearest:<key>' / 'heuristic')."""
    t = tuned_table()
    hit, src = None, 'heuristic'
    if _build_mode[0] == 'throughput' and plan_transfer.mode() != 'only':
        hit = t.get(sig + '_tp')
        src = 'table'
    if hit is None:
        if M > 0:
            hit, src = plan_transfer.lookup(t, sig, M, N, nkt, nseg)
        else:
            hit, src = t.get(sig), 'table'
            if hit is None:
                src = 'heuristic'
    return (hit, src) if with_source else hit


class _LinearAsConv:
    """nn.Linear viewed as a 1x1 convolution of the [tokens][C] (= NHWC) tensor."""

    def __init__(self, lin):
        self.lin = lin
        self.in_channels, self.out_channels = lin.in_features, lin.out_features
        self.kernel_size, self.stride, self.padding = (1, 1), (1, 1), (0, 0)

    @property
    def weight(self):
        return self.lin.weight.reshape(self.out_channels, self.in_channels, 1, 1)

    @property
    def bias(self):
        return self.lin.bias


class _Conv:
    """One fused convolution: packed parameters + static descriptor."""

    def __init__(self, name, convs, bn=None, act=ACT_NONE, stem=False):
        self.name = name
        self.convs = convs if isinstance(convs, (list, tuple)) else [convs]   # >1: concatenated along Cout
        self.bn = bn
        self.act = act
        self.stem = stem
        c0 = self.convs[0]
        self.cin = c0.in_channels
        self.kh, self.kw = c0.kernel_size
        self.stride = c0.stride[0]
        self.pad = c0.padding[0]
        self.cout = sum(c.out_channels for c in self.convs)
        self.cin_pad = 4 if stem else self.cin
        self.k_pad = _round_up(self.kh * self.kw * self.cin_pad, 32)
        self.weight = self.scale = self.shift = None
        self.desc = None
        self.tile = (0, 0)
        self.ksplit = 0
        self.kwaves = 0
        self.stages = 0
        self.tail = (0, 0)           # (tail_tiles, tail_ksplit): see ym_conv_desc
        self.grid_wgs = 0            # persistent kernel (stages 4x): workgroups launched, 0 = as many as the CUs hold
        self.mma = 0                 # 0 = f32 MFMA (parity mode); 3 / 6 = split-bf16 products (ym_conv_desc.mma)
        self._hit = None             # the plan row this conv was bound with, and where it came from (_entry)
        self._shape = (0, 0, 0, 1)   # (M, N, K tiles, output segments) of the bound launch
        self.plan_source = 'heuristic'

    def refresh(self):
        """(Re)pack parameters from the nn.Modules into the kernel layout, on device."""
        packed = [hip.pack_conv_weight(c.weight.detach().float(), self.cin_pad, self.k_pad) for c in self.convs]
        self.weight = packed[0] if len(packed) == 1 else torch.cat(packed, 0).contiguous()
        if self.bn is not None:
            bn = self.bn
            self.scale, self.shift = hip.fold_bn(bn.weight.detach().float().contiguous(),
                                                 bn.bias.detach().float().contiguous(),
                                                 bn.running_mean.float().contiguous(),
                                                 bn.running_var.float().contiguous(), float(bn.eps))
        else:
            self.scale = None
            biases = [c.bias for c in self.convs]
            if all(b is None for b in biases):
                self.shift = None
            else:
                self.shift = torch.cat([b.detach().float() for b in biases]).contiguous()
        if self.desc is not None:
            self._bind_params()

    def _bind_params(self):
        d = self.desc
        d.weight = self.weight.data_ptr()
        d.scale = self.scale.data_ptr() if self.scale is not None else None
        d.shift = self.shift.data_ptr() if self.shift is not None else None

    def bind(self, x, segs, residual=None):
        """x: NHWC input buffer; segs: list of (n_begin, n_end, tensor_base_ptr, batch_stride, pitch, act)."""
        b, h, w, cin = x.shape
        assert cin == self.cin_pad, (self.name, cin, self.cin_pad)
        ho = (h + 2 * self.pad - self.kh) // self.stride + 1
        wo = (w + 2 * self.pad - self.kw) // self.stride + 1
        d = ConvDesc()
        d.inp = x.data_ptr()
        d.residual = residual.data_ptr() if residual is not None else None
        d.B, d.H, d.W, d.Cin, d.Cout = b, h, w, cin, self.cout
        d.KH, d.KW, d.stride, d.pad, d.Ho, d.Wo, d.k_pad = self.kh, self.kw, self.stride, self.pad, ho, wo, self.k_pad
        d.nseg = len(segs)
        for i, (n0, n1, base_ptr, bstride, pitch, act) in enumerate(segs):
            d.seg[i].n_begin, d.seg[i].n_end = n0, n1
            d.seg[i].out = base_ptr
            d.seg[i].batch_stride, d.seg[i].pitch, d.seg[i].act = bstride, pitch, act
        d.tile_m, d.tile_n = self.tile
        d.ksplit = self.ksplit
        d.kwaves = self.kwaves
        d.stages = self.stages
        self.desc = d
        self._bind_params()
        self.out_hw = (ho, wo)
        self.flops = 2.0 * b * ho * wo * self.cout * self.kh * self.kw * self.cin
        self.sig = f'M{b * ho * wo}_N{self.cout}_C{cin}_k{self.kh}_s{self.stride}_seg{len(segs)}_r{int(residual is not None)}'
        self._shape = (b * ho * wo, self.cout, self.k_pad // 32, len(segs))
        hit, self.plan_source = _entry(self.sig, *self._shape, with_source=True)
        self._hit = hit
        if hit and self.tile == (0, 0) and self.ksplit == 0 and self.kwaves == 0:
            self.tile, self.ksplit, self.kwaves = (hit[0], hit[1]), hit[2], (hit[3] if len(hit) > 3 else 0)
            self.stages = hit[4] if len(hit) > 4 else 0
            self.tail = (hit[5], hit[6]) if len(hit) > 6 else (0, 0)
            d.tile_m, d.tile_n, d.ksplit, d.kwaves, d.stages = hit[0], hit[1], hit[2], self.kwaves, self.stages
            d.tail_tiles, d.tail_ksplit = self.tail
            self.grid_wgs = d.grid_wgs = _grid_wgs(hit)         # persistent kernel (stages 4x): workgroups launched
            cap = int(os.environ.get('YM_MAX_KSPLIT', '0') or 0)     # experiment knob: cap the K split of the tuned choice
            if cap and self.ksplit > cap:
                self.ksplit = d.ksplit = cap
        self.apply_mma(conv_mma())
        return ho, wo

    def tuned_key(self):
        return self.sig + (f'_mma{self.mma}' if self.mma else '')

    def apply_mma(self, mma):
        """Select the matrix pipe of this conv (ym_conv_desc.mma) where the split-bf16 kernel applies: Cin % 32 == 0, workgroup
        kernel.  Tile / split-K / tail choices are kept; the operand staging falls back to the register double buffer."""
        d = self.desc
        ok = mma in (3, 6) and not self.stem and d.Cin % 32 == 0 and d.nlevels == 0
        self.mma = mma if ok else 0
        hit = None
        if self.mma and os.environ.get('YM_NO_TUNED', '0') != '1':      # this pipe's own row: exact, or the nearest tuned shape's
            hit, _ = plan_transfer.lookup(tuned_table(), self.tuned_key(), *self._shape)
        if hit is None:
            hit = self._hit                 # (no entry for this pipe: the f32 choice, incl. its wave kernel for tiny layers)
            if self.mma and hit and len(hit) > 4 and 52 <= hit[4] <= 54:
                hit = [0, 0, 0, 0, 0, 0, 0]  # (the f32 choice is the weight-stationary kernel, whose tiles the split-bf16 kernel does not have)
            if self.mma and hit and self.plan_source != 'table' and len(hit) > 3 and hit[3] > 0:
                hit = [0, 0, 0, 0, 0, 0, 0]  # (a TRANSFERRED f32 row naming the wave kernel is no measurement against this pipe)
        d.mma = self.mma
        if hit and os.environ.get('YM_NO_TUNED', '0') != '1':     # each matrix pipe has its own measured tile / split-K / tail choice
            self.tile, self.ksplit, self.kwaves = (hit[0], hit[1]), hit[2], (hit[3] if len(hit) > 3 else 0)
            self.stages = hit[4] if len(hit) > 4 else 0
            self.tail = (hit[5], hit[6]) if len(hit) > 6 else (0, 0)
            cap = int(os.environ.get('YM_MAX_KSPLIT', '0') or 0)     # experiment knob: cap the K split of the tuned choice
            if cap and self.ksplit > cap:
                self.ksplit = cap
            d.tile_m, d.tile_n, d.ksplit, d.kwaves = self.tile[0], self.tile[1], self.ksplit, self.kwaves
            d.tail_tiles, d.tail_ksplit = self.tail
            self.grid_wgs = d.grid_wgs = _grid_wgs(hit)
            if self.kwaves:                                         # the tuner may prefer the f32 wave kernel for a tiny layer
                self.mma = 0
                d.mma = 0
        if self.mma:
            # split modes stage through registers: 0/2 = one register set (fewer VGPRs: two workgroups per CU on the big tiles),
            # 3 = two sets (the tile being converted arrived an iteration earlier: wins where occupancy is one wave per SIMD anyway)
            self.stages = 3 if self.stages == 3 else 0
            if os.environ.get('YM_MMA_STAGES'):
                self.stages = int(os.environ['YM_MMA_STAGES'])
        d.stages = self.stages


def _bind_pyramid(layer, pyr, batch, shapes, segs):
    """Bind a `_Conv` to a pyramid input (ym_conv_desc.nlevels): `pyr` [sum_l B*h_l*w_l, Cin] holds the levels back to back."""
    rows, cin = pyr.shape
    assert cin == layer.cin_pad and layer.stride == 1 and layer.pad == layer.kh // 2 and len(shapes) <= 5
    assert rows == sum(batch * h * w for h, w in shapes)
    d = ConvDesc()
    d.inp = pyr.data_ptr()
    d.B, d.H, d.W, d.Cin, d.Cout = batch, shapes[0][0], shapes[0][1], cin, layer.cout
    d.KH, d.KW, d.stride, d.pad, d.Ho, d.Wo, d.k_pad = layer.kh, layer.kw, 1, layer.pad, shapes[0][0], shapes[0][1], layer.k_pad
    d.nlevels = len(shapes)
    for l, (h, w) in enumerate(shapes):
        d.level_h[l], d.level_w[l] = h, w
    d.nseg = len(segs)
    for i, (n0, n1, base_ptr, bstride, pitch, act) in enumerate(segs):
        d.seg[i].n_begin, d.seg[i].n_end = n0, n1
        d.seg[i].out = base_ptr
        d.seg[i].batch_stride, d.seg[i].pitch, d.seg[i].act = bstride, pitch, act
    layer.desc = d
    layer._bind_params()
    layer.out_hw = shapes[0]
    layer.flops = 2.0 * rows * layer.cout * layer.kh * layer.kw * layer.cin
    layer.sig = f'M{rows}_N{layer.cout}_C{cin}_k{layer.kh}_s1_seg{len(segs)}_r0_L{len(shapes)}'
    layer._shape = (rows, layer.cout, layer.k_pad // 32, len(segs))
    hit, layer.plan_source = _entry(layer.sig, *layer._shape, with_source=True)
    layer._hit = hit
    if hit:
        layer.tile, layer.ksplit, layer.kwaves = (hit[0], hit[1]), hit[2], 0
        layer.stages = 0
        layer.tail = (hit[5], hit[6]) if len(hit) > 6 else (0, 0)
    d.tile_m, d.tile_n = layer.tile
    d.ksplit = layer.ksplit
    d.tail_tiles, d.tail_ksplit = layer.tail


class InferEngine:
    def __init__(self, net, batch, height, width, device, use_graph=None, mode='latency'):
        self.net = net
        self.mode = mode                   # which tuned entries the plan reads: see _entry
        self.device = device
        self.B, self.H, self.W = batch, height, width
        self.num_classes = net.cfg.num_classes
        self.coef_dim = net.coef_dim
        self.na = len(net.cfg.aspect_ratios)
        if use_graph is None:
            use_graph = os.environ.get('YM_GRAPH', '1') != '0'
        self.use_graph = use_graph
        self.graph = None
        self.ops = []          # (kind, arg): one C-ABI launch each, in a valid sequential order
        self.op_stream = {}    # op index -> branch stream id (0 = the caller's stream)
        self.op_deps = {}      # op index -> indices of producer ops on OTHER streams (fork/join edges of the graph)
        self._producers = {}   # buffer data_ptr -> [op indices that write it]
        self._cur_stream = 0
        self._side_streams = {}
        self._events = {}
        self.use_branches = os.environ.get('YM_BRANCHES', '0') == '1'   # measured neutral on ROCm 7.2 hipGraph: off by default
        self.convs = []        # every _Conv launched through ym_conv2d_fwd, for weight refresh / tuning / flop accounting
        self.fused_stem = None  # the ResNet stem's _Conv when it runs inside the fused stem + max-pool launch (not in `convs`)
        self._img = None
        self._weights_epoch = -1
        self._bufs = []
        hip.lib()              # fail loudly here if the .so is missing
        prev, _build_mode[0] = _build_mode[0], mode
        try:
            with torch.cuda.device(device):
                self._build()
                if autotune_on() and os.environ.get('YM_NO_TUNED', '0') != '1':
                    self._autotune_missing()
        finally:
            _build_mode[0] = prev

    def _autotune_missing(self):
        """YM_AUTOTUNE=1: measure the launches of this plan that have no row of their own, keep the rows (process + user cache)."""
        have = {c.sig for c in self.convs if c.plan_source == 'table'}
        todo = {c.sig for c in self.convs} - have
        if not todo:
            return
        rows = self.autotune(iters=10, skip=have)
        rows = {k: (v if len(v) > 7 and v[7] else v[:7]) for k, v in rows.items()}
        tuned_table().update(rows)
        plan_transfer._index_cache.clear()
        for c in self.convs:
            if c.sig in rows:
                c.plan_source, c._hit = 'autotuned', rows[c.sig]
        try:
            _store_user_rows(rows)
        except OSError as e:
            print(f'yolact_minimal_amd: could not write {user_cache_path()}: {e}', file=sys.stderr)

    # ---- construction ------------------------------------------------------------------------
    def _buf(self, *shape):
        t = torch.empty(*shape, device=self.device, dtype=torch.float32)
        self._bufs.append(t)
        return t

    def _add_op(self, kind, arg, inputs, outputs):
        idx = len(self.ops)
        self.ops.append((kind, arg))
        sid = self._cur_stream if self.use_branches else 0
        self.op_stream[idx] = sid
        deps = []
        for t in inputs:
            if t is None:
                continue
            for j in self._producers.get(t.data_ptr(), ()):
                if self.op_stream[j] != sid and j not in deps:
                    deps.append(j)
        if deps:
            self.op_deps[idx] = deps
        for t in outputs:
            self._producers.setdefault(t.data_ptr(), []).append(idx)
        return idx

    def _conv(self, layer, x, out=None, residual=None, segs=None, seg_outputs=None):
        """Register a fused conv launch; returns its NHWC output buffer."""
        b, h, w, _ = x.shape
        ho = (h + 2 * layer.pad - layer.kh) // layer.stride + 1
        wo = (w + 2 * layer.pad - layer.kw) // layer.stride + 1
        if segs is None:
            if out is None:
                out = self._buf(b, ho, wo, layer.cout)
            segs = [(0, layer.cout, out.data_ptr(), ho * wo * layer.cout, layer.cout, layer.act)]
        layer.refresh()
        layer.bind(x, segs, residual)
        self.convs.append(layer)
        self._add_op('conv', layer, [x, residual], seg_outputs if seg_outputs is not None else [out])
        return out

    def _build_swin(self):
        """Swin-T backbone (reference modules/swin_transformer.py:500-518): returns the stage-1..3 maps, layer-normed, NHWC."""
        net, B = self.net, self.B
        bb = net.backbone
        ws = bb.window_size
        assert self.H % 4 == 0 and self.W % 4 == 0
        pe = bb.patch_embed
        x = self._conv(_Conv('backbone.patch_embed.proj', pe.proj, act=ACT_NONE, stem=True), self.x_in)     # [B,H/4,W/4,96]
        self._add_op('layernorm', (x, pe.norm, x), [x], [x])
        feats = []
        for li, layer in enumerate(bb.layers):
            _, h, w, c = x.shape
            heads = bb.heads[li]
            for bi, blk in enumerate(layer.blocks):
                p = f'backbone.layers.{li}.blocks.{bi}'
                n1 = self._buf(B, h, w, c)
                self._add_op('layernorm', (x, blk.norm1, n1), [x], [n1])
                qkv = self._conv(_Conv(p + '.attn.qkv', _LinearAsConv(blk.attn.qkv)), n1)
                att = self._buf(B, h, w, c)
                self._add_op('attn', (qkv, blk.attn, att, (B, h, w, c, heads, ws, blk.shift_size)), [qkv], [att])
                x = self._conv(_Conv(p + '.attn.proj', _LinearAsConv(blk.attn.proj)), att, residual=x)
                n2 = self._buf(B, h, w, c)
                self._add_op('layernorm', (x, blk.norm2, n2), [x], [n2])
                hid = self._conv(_Conv(p + '.mlp.fc1', _LinearAsConv(blk.mlp.fc1), act=ACT_GELU), n2)
                x = self._conv(_Conv(p + '.mlp.fc2', _LinearAsConv(blk.mlp.fc2)), hid, residual=x)
            if li in bb.out_norm_indices:
                f = self._buf(B, h, w, c)
                self._add_op('layernorm', (x, getattr(bb, f'norm{li}'), f), [x], [f])
                feats.append(f)
            if layer.downsample is not None:
                ho, wo = (h + 1) // 2, (w + 1) // 2
                merged = self._buf(B, ho, wo, 4 * c)
                self._add_op('merge_ln', (x, layer.downsample.norm, merged), [x], [merged])
                x = self._conv(_Conv(f'backbone.layers.{li}.downsample.reduction', _LinearAsConv(layer.downsample.reduction)), merged)
        return feats

    def _build(self):
        net, B, H, W = self.net, self.B, self.H, self.W
        bb = net.backbone
        self.x_in = self._buf(B, H, W, 4)
        self.static_img = None
        if hasattr(bb, 'patch_embed'):
            c3, c4, c5 = self._build_swin()
        else:
            c3, c4, c5 = self._build_resnet()
        self._build_neck_and_heads(c3, c4, c5)

    def _build_resnet(self):
        net, B, H, W = self.net, self.B, self.H, self.W
        bb = net.backbone

        # stem + maxpool: one launch straight from the NCHW image (csrc/stem.hip; YM_FUSED_STEM=0: NHWC4 copy + conv + pool, the
        # same bits from three launches and an 18.9 MB round trip per image)
        stem = _Conv('backbone.conv1', bb.conv1, bb.bn1, ACT_RELU, stem=True)
        conv1 = bb.conv1
        fusable = (conv1.out_channels == 64 and conv1.in_channels == 3 and tuple(conv1.kernel_size) == (7, 7) and
                   tuple(conv1.stride) == (2, 2) and tuple(conv1.padding) == (3, 3))
        if fusable and os.environ.get('YM_FUSED_STEM', '1') != '0':
            stem.refresh()
            ho, wo = (H + 6 - 7) // 2 + 1, (W + 6 - 7) // 2 + 1
            hp, wp = (ho + 2 - 3) // 2 + 1, (wo + 2 - 3) // 2 + 1
            stem.flops = 2.0 * B * ho * wo * stem.cout * stem.kh * stem.kw * stem.cin
            self.fused_stem = stem
            pooled = self._buf(B, hp, wp, 64)
            self._add_op('stem_pool', (stem, pooled), [], [pooled])
            x = pooled
        else:
            x = self._conv(stem, self.x_in)
            hp, wp = (x.shape[1] + 2 - 3) // 2 + 1, (x.shape[2] + 2 - 3) // 2 + 1
            pooled = self._buf(B, hp, wp, 64)
            self._add_op('maxpool', (x, pooled), [x], [pooled])
            x = pooled

        # residual stages
        stage_outs = []
        for li, stage in enumerate(bb.layers):
            for bi, blk in enumerate(stage):
                p = f'backbone.layers.{li}.{bi}'
                y = self._conv(_Conv(p + '.conv1', blk.conv1, blk.bn1, ACT_RELU), x)
                y = self._conv(_Conv(p + '.conv2', blk.conv2, blk.bn2, ACT_RELU), y)
                if blk.downsample is not None:
                    skip = self._conv(_Conv(p + '.downsample', blk.downsample[0], blk.downsample[1], ACT_NONE), x)
                else:
                    skip = x
                x = self._conv(_Conv(p + '.conv3', blk.conv3, blk.bn3, ACT_RELU), y, residual=skip)
            stage_outs.append(x)
        return stage_outs[1:4]

    def _build_neck_and_heads(self, c3, c4, c5):
        net, B = self.net, self.B
        # FPN (top-down adds fused as the lateral conv's residual)
        fpn = net.fpn
        p5_1 = self._conv(_Conv('fpn.lat_layers.2', fpn.lat_layers[2]), c5)
        u5 = self._buf(B, p5_1.shape[1] * 2, p5_1.shape[2] * 2, 256)
        self._add_op('bilinear', (p5_1, u5, False), [p5_1], [u5])
        assert u5.shape[1:3] == c4.shape[1:3], 'img_size must be divisible by 32 (reference config.py:75)'
        p4_1 = self._conv(_Conv('fpn.lat_layers.1', fpn.lat_layers[1]), c4, residual=u5)
        u4 = self._buf(B, p4_1.shape[1] * 2, p4_1.shape[2] * 2, 256)
        self._add_op('bilinear', (p4_1, u4, False), [p4_1], [u4])
        p3_1 = self._conv(_Conv('fpn.lat_layers.0', fpn.lat_layers[0]), c3, residual=u4)
        # Independent branches run on side streams (fork/join edges come from the data flow, see _add_op):
        #   stream 0: P3 + ProtoNet (the long chain)    stream 1: P5, P6, P7 + their heads
        #   stream 2: P4 + its head                       stream 3: the P3 head
        # P3..P7 live back to back in ONE buffer (level-major), so that the shared prediction head can run over the whole
        # pyramid in one launch per conv (ym_conv_desc.nlevels) instead of one launch per level
        # (measured: -3 % forward time at bs=1, where every launch is latency bound; at bs=8 the per-level launches with their
        #  individually tuned direct-to-LDS kernels are 1-2 % faster, so the pyramid launch is used for small batches only)
        pyr_env = os.environ.get('YM_PYRAMID_HEAD', 'auto')
        pyramid = pyr_env == '1' or (pyr_env == 'auto' and B <= 2)
        h5, w5 = p5_1.shape[1:3]
        shapes = [tuple(p3_1.shape[1:3]), tuple(p4_1.shape[1:3]), (h5, w5), ((h5 + 1) // 2, (w5 + 1) // 2),
                  (((h5 + 1) // 2 + 1) // 2, ((w5 + 1) // 2 + 1) // 2)]
        rows = [B * h * w for h, w in shapes]
        pyr = self._buf(sum(rows), 256)
        views, r0 = [], 0
        for (h, w), n in zip(shapes, rows):
            views.append(pyr[r0:r0 + n].view(B, h, w, 256))
            r0 += n
        self._cur_stream = 1
        p5 = self._conv(_Conv('fpn.pred_layers.2', fpn.pred_layers[2][0], act=ACT_RELU), p5_1, out=views[2])
        p6 = self._conv(_Conv('fpn.downsample_layers.0', fpn.downsample_layers[0][0], act=ACT_RELU), p5, out=views[3])
        p7 = self._conv(_Conv('fpn.downsample_layers.1', fpn.downsample_layers[1][0], act=ACT_RELU), p6, out=views[4])
        self._cur_stream = 2
        p4 = self._conv(_Conv('fpn.pred_layers.1', fpn.pred_layers[1][0], act=ACT_RELU), p4_1, out=views[1])
        self._cur_stream = 0
        p3 = self._conv(_Conv('fpn.pred_layers.0', fpn.pred_layers[0][0], act=ACT_RELU), p3_1, out=views[0])
        levels = [p3, p4, p5, p6, p7]
        assert [tuple(lv.shape[1:3]) for lv in levels] == shapes
        level_stream = [3, 2, 1, 1, 1]

        # ProtoNet
        pn = net.proto_net
        y = p3
        for i in (0, 2, 4):
            y = self._conv(_Conv(f'proto_net.proto1.{i}', pn.proto1[i], act=ACT_RELU), y)
        up = self._buf(B, y.shape[1] * 2, y.shape[2] * 2, 256)
        self._add_op('bilinear', (y, up, True), [y], [up])
        y = self._conv(_Conv('proto_net.proto2.0', pn.proto2[0], act=ACT_RELU), up)
        self.proto_out = self._conv(_Conv('proto_net.proto2.2', pn.proto2[2], act=ACT_RELU), y)  # NHWC = [B,Hp,Wp,32]

        # shared head, written straight into the concatenated outputs
        hd = net.prediction_layers
        nc, cd, na = self.num_classes, self.coef_dim, self.na
        n_total = sum(lv.shape[1] * lv.shape[2] * na for lv in levels)
        assert n_total * 4 == len(net.anchors) or not isinstance(net.anchors, list), 'anchor count mismatch'
        self.n_anchors = n_total
        self.class_logits = self._buf(B, n_total, nc)
        self.class_pred = self._buf(B, n_total, nc)
        self.box_pred = self._buf(B, n_total, 4)
        self.coef_pred = self._buf(B, n_total, cd)
        c_conf, c_box, c_coef = na * nc, na * 4, na * cd
        if pyramid:
            self._cur_stream = 0
            up = _Conv('prediction_layers.upfeature@P3-7', hd.upfeature[0], act=ACT_RELU)
            xh = self._buf(sum(rows), 256)
            up.refresh()
            _bind_pyramid(up, pyr, B, shapes, [(0, 256, xh.data_ptr(), 0, 256, ACT_RELU)])
            self.convs.append(up)
            self._add_op('conv', up, levels, [xh])
            fused = _Conv('prediction_layers.conf|bbox|coef@P3-7', [hd.conf_layer, hd.bbox_layer, hd.coef_layer[0]])
            segs = [(0, c_conf, self.class_logits.data_ptr(), n_total * nc, c_conf, ACT_NONE),
                    (c_conf, c_conf + c_box, self.box_pred.data_ptr(), n_total * 4, c_box, ACT_NONE),
                    (c_conf + c_box, c_conf + c_box + c_coef, self.coef_pred.data_ptr(), n_total * cd, c_coef, ACT_TANH)]
            fused.refresh()
            _bind_pyramid(fused, xh, B, shapes, segs)
            self.convs.append(fused)
            self._add_op('conv', fused, [xh], [self.class_logits, self.box_pred, self.coef_pred])
        off = 0
        for li, lv in enumerate(levels if not pyramid else []):
            self._cur_stream = level_stream[li]
            xh = self._conv(_Conv(f'prediction_layers.upfeature@P{li + 3}', hd.upfeature[0], act=ACT_RELU), lv)
            fused = _Conv(f'prediction_layers.conf|bbox|coef@P{li + 3}', [hd.conf_layer, hd.bbox_layer, hd.coef_layer[0]])
            es = 4  # bytes per float
            segs = [
                (0, c_conf, self.class_logits.data_ptr() + off * nc * es, n_total * nc, c_conf, ACT_NONE),
                (c_conf, c_conf + c_box, self.box_pred.data_ptr() + off * 4 * es, n_total * 4, c_box, ACT_NONE),
                (c_conf + c_box, c_conf + c_box + c_coef, self.coef_pred.data_ptr() + off * cd * es, n_total * cd,
                 c_coef, ACT_TANH),
            ]
            self._conv(fused, xh, segs=segs, seg_outputs=[self.class_logits, self.box_pred, self.coef_pred])
            off += lv.shape[1] * lv.shape[2] * na
        self._cur_stream = 0
        self._add_op('softmax', (self.class_logits, self.class_pred), [self.class_logits], [self.class_pred])

        self._alloc_workspaces()
        self.total_flops = sum(c.flops for c in self.convs) + (self.fused_stem.flops if self.fused_stem is not None else 0.0)
        self._weights_epoch = self.net._weights_epoch

    def _alloc_workspaces(self):
        """Split-K scratch + arrival counters: one set per branch stream (concurrent branches must not share them)."""
        sids = {self.op_stream.get(i, 0) for i, (kind, _) in enumerate(self.ops) if kind == 'conv'} | {0}
        # split-K arrival counters (fused finish, ym_conv_desc.tile_counters): zeroed once, the kernels leave them zero
        fused = os.environ.get('YM_FUSED_SPLITK', '1') != '0'
        oldc = getattr(self, 'counters', {})
        self.counters = {sid: oldc[sid] if sid in oldc else torch.zeros(hip.TILE_COUNTERS, device=self.device, dtype=torch.int32)
                         for sid in sids}
        need = {sid: 256 for sid in sids}
        for i, (kind, arg) in enumerate(self.ops):
            if kind == 'conv':
                sid = self.op_stream.get(i, 0)
                arg.desc.tile_counters = self.counters[sid].data_ptr() if fused else None
                if not fused:
                    arg.desc.tail_tiles, arg.desc.tail_ksplit = 0, 0
                nb = hip.conv_workspace_bytes(arg.desc)
                need[sid] = max(need[sid], nb)
        need[0] = max(need.values())          # stream 0's buffer also serves sequential (eager / profiling) replays
        old = getattr(self, 'workspaces', {})
        self.workspaces = {sid: (old[sid] if sid in old and old[sid].numel() >= nb else
                                 torch.empty(nb, device=self.device, dtype=torch.uint8)) for sid, nb in need.items()}
        self.workspace = self.workspaces[0]

    # ---- execution ---------------------------------------------------------------------------
    def refresh_weights(self):
        for c in self.convs:
            c.refresh()
        if self.fused_stem is not None:
            self.fused_stem.refresh()
        self._weights_epoch = self.net._weights_epoch
        self.graph = None

    def retune(self):
        """Re-read tile/ksplit knobs of every conv into its descriptor (after changing `layer.tile/ksplit`)."""
        for c in self.convs:
            c.desc.tile_m, c.desc.tile_n = c.tile
            c.desc.ksplit = c.ksplit
            c.desc.kwaves = c.kwaves
            c.desc.stages = c.stages
            c.desc.tail_tiles, c.desc.tail_ksplit = c.tail
            c.desc.grid_wgs = c.grid_wgs
            c.desc.mma = c.mma
        self._alloc_workspaces()
        self.graph = None

    def set_mma(self, mma):
        """Switch every eligible conv of the plan between the f32 MFMA (0) and the split-bf16 modes (3 / 6)."""
        prev, _build_mode[0] = _build_mode[0], self.mode
        try:
            for c in self.convs:
                c.apply_mma(mma)
        finally:
            _build_mode[0] = prev
        self.retune()

    def autotune(self, iters=10, verbose=False, mma=0, concurrent=False, skip=(), cus=256):
        """Time every (tile, ksplit) candidate of every distinct conv shape on this GPU; keep the fastest.
        Returns {signature: [tile_m, tile_n, ksplit, kwaves, stages, tail_tiles, tail_ksplit, grid_wgs]} = rows of the tuned table;
        the timings are left in `self.autotune_detail` = {signature: (best_us, default_us)}.  Persistent candidates (stages 4x) are
        timed and kept with grid_wgs = 0 (as many workgroups as the CUs hold; tools/pers_bench.py sweeps the grid).
        `concurrent`: tune for THROUGHPUT with requests in flight (bench.py --inflight 2) instead of for the latency of a launch
        that has the chip to itself: every candidate is timed as two copies of the launch running side by side on two streams
        (own split-K scratch and arrival counters each); the figure is the wall time per PAIR, so a choice that wins by spreading
        thin over all CUs (many K slices + an exchange) loses to one that does the same work with fewer resources.
        `skip`: signatures (table keys) that are left as they are (tools/autotune.py --skip-known).
        `cus`: compute units the launches can use (a CU-masked stream, cu_mask.py: the tail-split candidates follow it)."""
        results = {}
        self.autotune_detail = {}
        big_ws = torch.empty(1 << 28, device=self.device, dtype=torch.uint8)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ncopy = int(concurrent) if concurrent else 1        # True = 2 copies
        if ncopy == 1 and concurrent:
            ncopy = 2
        if concurrent:
            from .pipeline import _stream_set
            sides = _stream_set(self.device, ncopy - 1)
            big_ws2 = [torch.empty(1 << 27, device=self.device, dtype=torch.uint8) for _ in sides]
            counters2 = [torch.zeros(hip.TILE_COUNTERS, device=self.device, dtype=torch.int32) for _ in sides]
            main = torch.cuda.current_stream(self.device)

        def launch(d, d2):
            hip.conv2d_fwd(d, big_ws)
            if concurrent:                       # (same inputs and outputs: all copies write the same values; scratch + counters differ)
                for i, side in enumerate(sides):
                    with torch.cuda.stream(side):
                        hip.conv2d_fwd(d2[i], big_ws2[i])

        def time_cfg(c, tile, ks, kwv=0, stg=0, tail=(0, 0), gw=0):
            d = c.desc
            d.tile_m, d.tile_n, d.ksplit, d.kwaves, d.stages = tile[0], tile[1], ks, kwv, stg
            d.mma = mma if (split_ok(c) and kwv == 0) else 0
            d.tail_tiles, d.tail_ksplit = tail
            d.grid_wgs = gw                      # (never the previous table entry's grid; wave kernel with DMA rings: waves per workgroup)
            need = hip.conv_workspace_bytes(d)
            if need > big_ws.numel():
                return None
            d2 = None
            if concurrent:
                if need > big_ws2[0].numel():
                    return None
                d2 = [type(d).from_buffer_copy(d) for _ in sides]
                for i, dd in enumerate(d2):
                    if d.tile_counters:
                        dd.tile_counters = counters2[i].data_ptr()
            try:
                for _ in range(2):
                    launch(d, d2)
            except RuntimeError:
                return None
            best = 1e30
            for _ in range(3):
                torch.cuda.synchronize()
                ev0.record()
                for _ in range(iters):
                    launch(d, d2)
                if concurrent:
                    for side in sides:
                        main.wait_stream(side)
                ev1.record()
                torch.cuda.synchronize()
                best = min(best, ev0.elapsed_time(ev1) / iters * 1e3)
            return best

        def split_ok(c):
            return mma in (3, 6) and not c.stem and c.desc.Cin % 32 == 0 and c.desc.nlevels == 0

        seen = {}
        for c in self.convs:
            if mma and not split_ok(c):
                continue
            if c.sig + (f'_mma{mma}' if mma else '') in skip:
                continue
            if c.sig in seen:
                c.tile, c.ksplit, c.kwaves, c.stages, c.tail, c.grid_wgs = seen[c.sig]
                c.mma = mma if (mma and c.kwaves == 0) else 0
                continue
            d = c.desc
            M, nkt = d.B * d.Ho * d.Wo, d.k_pad // 32
            if d.nlevels:
                M = sum(d.B * d.level_h[l] * d.level_w[l] for l in range(d.nlevels))
            base = time_cfg(c, (0, 0), 0, 0)
            cands, wave_cands = [], []
            tiles = [(128, 64)] if c.stem else [(128, 128), (128, 64), (64, 128), (64, 64)]
            for tm, tn in tiles:
                wgs = -(-M // tm) * -(-d.Cout // tn)
                for ks in (1, 2, 3, 4, 6, 8, 12, 16, 24):
                    if ks > 1 and (wgs >= 1024 or ks * 2 > nkt or wgs * ks > 8192):
                        continue
                    cands.append(((tm, tn), ks, 0, 2, (0, 0)))
                    if (tm, tn) != (128, 128) and nkt // ks >= 3:
                        cands.append(((tm, tn), ks, 0, 3, (0, 0)))
                    if not c.stem:                       # direct-to-LDS staging, ring of 2 / 3 (/ 4)
                        cands.append(((tm, tn), ks, 0, 22, (0, 0)))
                        if nkt // ks >= 3:
                            cands.append(((tm, tn), ks, 0, 23, (0, 0)))
                        if (tm, tn) == (64, 64) and nkt // ks >= 4:
                            cands.append(((tm, tn), ks, 0, 24, (0, 0)))
                        if (tm, tn) == (64, 64) and nkt // ks >= 2:    # + software-pipelined fragments
                            cands.append(((tm, tn), ks, 0, 33, (0, 0)))
                            cands.append(((tm, tn), ks, 0, 34, (0, 0)))
                        if (tm, tn) == (64, 64) and d.nseg == 1 and d.tile_counters and c.act in (ACT_NONE, ACT_RELU):
                            cands.append(((tm, tn), ks, 0, 43, (0, 0)))    # persistent kernel (conv_persist.hip), ring of 3 / 6
                            cands.append(((tm, tn), ks, 0, 46, (0, 0)))
                # workgroup-quantisation fix: split the tiles of the last partial round (over 256 CUs x 1 or 2 workgroups)
                if not c.stem and d.nseg == 1 and d.tile_counters and cus < wgs <= hip.TILE_COUNTERS:
                    for r in sorted({wgs % cus, wgs % (2 * cus)} - {0}):
                        for ts in (2, 3, 4, 6, 8):
                            if ts * 2 > nkt or r * ts > 2048:
                                continue
                            for stg in ((2, 3, 22, 23) + ((33, 34) if (tm, tn) == (64, 64) else ()) if (tm, tn) != (128, 128) and nkt // ts >= 3 else (2, 22)):
                                cands.append(((tm, tn), 1, 0, stg, (r, ts)))
            if not c.stem:
                for tm, tn in ((32, 32), (64, 32), (32, 64), (64, 64)):
                    waves = -(-M // tm) * -(-d.Cout // tn)
                    for kwv in (1, 2, 4, 8):
                        if kwv > 1 and (waves >= 4096 or kwv > nkt):
                            continue
                        if kwv == 8 and tm * tn == 4096:
                            continue
                        if waves * kwv > 65536:
                            continue
                        cands.append(((tm, tn), 1, kwv, 0, (0, 0)))
                # the wave kernel with private DMA rings (conv_wdma_f32): K split inside the workgroup, no K-slice exchange; one- and
                # two-wave workgroups where the waves share nothing; the tail split of a 32x32 / four-K-wave plan.  (A launch repeated
                # back to back undervalues it against the kernels with a cross-workgroup exchange: tools/tune_forward.py judges the
                # candidates by the plan's forward time.)
                if d.Cin % 32 == 0 and d.nlevels == 0:
                    for tm, tn in ((32, 32), (64, 32), (32, 64)):
                        for kwv in (1, 2, 4):
                            if kwv <= nkt:
                                wave_cands.append(((tm, tn), 1, kwv, 22, (0, 0), 0))
                                wave_cands += [((tm, tn), 1, kwv, 22, (0, 0), wpb) for wpb in (1, 2) if wpb >= kwv and kwv < 4]
                    tiles32 = -(-M // 32) * -(-d.Cout // 32)
                    if cus < tiles32 <= hip.TILE_COUNTERS and d.nseg == 1 and d.tile_counters:
                        wave_cands += [((32, 32), 1, 4, 22, (tiles32 % cus or cus, ts), 0) for ts in (4, 6, 8) if ts * 2 <= nkt]
            if mma:                                          # split-bf16: register staging with one (0) or two (3) register sets
                cands = sorted({(tile, ks, kwv, st, tail) for tile, ks, kwv, stg, tail in cands for st in ((0, 3) if kwv == 0 else (0,))})
            best = (base, (0, 0), 0, 0, 0, (0, 0), 0)
            for cand in [cd + (0,) for cd in cands] + ([] if mma else wave_cands):
                tile, ks, kwv, stg, tail, gw = cand
                t = time_cfg(c, tile, ks, kwv, stg, tail, gw)
                if t is not None and t < best[0] * 0.98:
                    best = (t, tile, ks, kwv, stg, tail, gw)
            c.tile, c.ksplit, c.kwaves, c.stages, c.tail, c.grid_wgs = best[1], best[2], best[3], best[4], best[5], best[6]
            c.mma = mma if (mma and c.kwaves == 0) else 0
            seen[c.sig] = (c.tile, c.ksplit, c.kwaves, c.stages, c.tail, c.grid_wgs)
            results[c.sig + (f'_mma{mma}' if mma else '')] = [best[1][0], best[1][1], best[2], best[3], best[4], best[5][0], best[5][1], best[6]]
            self.autotune_detail[c.sig + (f'_mma{mma}' if mma else '')] = (round(best[0], 2), round(base, 2))
            if verbose:
                print(f'{c.sig:44s} default {base:8.1f} us -> {best[1]} ks={best[2]} kw={best[3]} st={best[4]} tail={best[5]} {best[0]:8.1f} us', flush=True)
        del big_ws
        self.retune()
        return results

    def _launch_one(self, kind, arg, ws):
        if kind == 'conv':
            hip.conv2d_fwd(arg.desc, ws)
        elif kind == 'maxpool':
            hip.maxpool3x3s2(arg[0], arg[1])
        elif kind == 'stem_pool':
            hip.stem_conv_bn_relu_maxpool(self._img, arg[0].weight, arg[0].scale, arg[0].shift, arg[1])
        elif kind == 'bilinear':
            hip.bilinear2x(arg[0], arg[1], arg[2])
        elif kind == 'softmax':
            hip.softmax_rows(arg[0], arg[1])
        elif kind == 'layernorm':
            hip.layernorm(arg[0], arg[1].weight.detach(), arg[1].bias.detach(), arg[1].eps, arg[2])
        elif kind == 'merge_ln':
            hip.patch_merge_layernorm(arg[0], arg[1].weight.detach(), arg[1].bias.detach(), arg[1].eps, arg[2])
        elif kind == 'attn':
            qkv, attn, out, (b, h, w, c, heads, win, shift) = arg
            hip.swin_window_attention(qkv, attn.qkv.bias.detach(), attn.relative_position_bias_table.detach(), b, h, w, c,
                                      heads, win, shift, out)

    def _launch_all(self, img):
        """Replay the plan.  Ops tagged with a branch stream run on side streams; cross-stream producer->consumer edges
        become event waits (inside hipGraph capture they become graph edges, so independent branches overlap)."""
        if self.fused_stem is None:
            hip.nchw_to_nhwc4(img, self.x_in)
        self._img = img                                    # (the fused stem reads the NCHW image itself)
        main = torch.cuda.current_stream()
        needs_event = {j for deps in self.op_deps.values() for j in deps}
        used = set()
        for i, (kind, arg) in enumerate(self.ops):
            sid = self.op_stream.get(i, 0)
            if sid == 0:
                st = main
            else:
                st = self._side_streams.get(sid)
                if st is None:
                    st = self._side_streams[sid] = torch.cuda.Stream(device=self.device)
                used.add(sid)
            for j in self.op_deps.get(i, ()):
                st.wait_event(self._events[j])
            if sid == 0:
                self._launch_one(kind, arg, self.workspaces[0])
            else:
                with torch.cuda.stream(st):
                    self._launch_one(kind, arg, self.workspaces[sid])
            if i in needs_event:
                ev = self._events.get(i)
                if ev is None:
                    ev = self._events[i] = torch.cuda.Event()
                ev.record(st)
        for sid in used:
            main.wait_stream(self._side_streams[sid])

    def run(self, img):
        """Launch the plan; results land in the engine-owned buffers (no allocation, no sync)."""
        if img.dtype != torch.float32 or not img.is_contiguous():
            img = img.float().contiguous()
        if tuple(img.shape) != (self.B, 3, self.H, self.W):
            raise RuntimeError(f'engine built for {(self.B, 3, self.H, self.W)}, got {tuple(img.shape)}')
        if self._weights_epoch != self.net._weights_epoch:
            self.refresh_weights()
        if not self.use_graph:
            self._launch_all(img)
            return
        if self.graph is None:
            self.static_img = torch.empty_like(img)
            self.static_img.copy_(img)
            torch.cuda.synchronize()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                self._launch_all(self.static_img)          # warm-up outside capture
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            try:
                g = torch.cuda.CUDAGraph()
                # thread_local: only THIS thread's calls are checked against the capture.  In the default global mode an event query
                # from any other thread while the capture is open is an error -- and the process group's watchdog thread polls the
                # events of its collectives exactly like that (seen once as an abort of `bench.py` under torchrun: the barrier's
                # work object was still being polled when the first engine captured its plan).
                with torch.cuda.graph(g, capture_error_mode='thread_local'):
                    self._launch_all(self.static_img)
                self.graph = g
            except Exception as exc:                       # still the HIP kernels, just launched one by one
                import warnings
                warnings.warn(f'hipGraph capture failed ({exc!r}); replaying the plan eagerly')
                self.use_graph = False
                torch.cuda.synchronize()
                self._launch_all(img)
                return
        self.static_img.copy_(img)
        self.graph.replay()

    def outputs(self):
        return self.class_pred, self.box_pred, self.coef_pred, self.proto_out

    def forward(self, img):
        self.run(img)
        return tuple(t.clone() for t in self.outputs())
