"""Train-mode Swin-T backbone on the HIP kernels (SURVEY.md §8 row a18 under `loss.backward()`).

Reference: `SwinTransformer.forward` `/root/reference/modules/swin_transformer.py:500-518`, `SwinTransformerBlock.forward
:234-289`, `WindowAttention.forward :172-200`, `Mlp.forward :92-96`, `PatchMerging.forward :299-325`, `DropPath :62-82`.

`torch.autograd.Function` is only the tape.  Every Linear is the 1x1 case of `train_engine.ConvBias` (forward, data gradient
and weight gradient on the f32 MFMA conv kernels; the residual adds of the block are fused into the proj / fc2 epilogues when
DropPath is inactive); LayerNorm, the patch-merge gather + LayerNorm, GELU and the shifted-window attention have their own
forward / backward kernels (`csrc/swin_ops.hip`, `csrc/swin_train.hip`).  DropPath draws its per-sample numbers with `torch.rand` on
the device exactly like the reference (`:76-79`) — a host-level RNG call, kept for parity of the random stream; mask, scaling and the
residual add are one kernel (`DropPathAddFn`).
"""
import ctypes

import torch

from . import hip
from .hip import ACT_NONE
from .train_engine import ConvBias, _grad_slot, scratch


def _vp(t):
    return ctypes.c_void_p(t.data_ptr())


def _ln_ws(device, c):
    return scratch(device, hip.lib().ym_layernorm_bwd_workspace_bytes(c))


class LayerNormFn(torch.autograd.Function):
    """F.layer_norm over the last dim of an NHWC / token tensor."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        x = x.contiguous()
        out = torch.empty_like(x)
        hip.layernorm(x, gamma.detach(), beta.detach(), eps, out)
        ctx.save_for_backward(x, gamma)
        ctx.eps, ctx.beta = eps, beta
        return out

    @staticmethod
    def backward(ctx, dy):
        x, gamma = ctx.saved_tensors
        c = x.shape[-1]
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        dgamma, dbeta = _grad_slot(gamma, (c,)), _grad_slot(ctx.beta, (c,))
        ws = _ln_ws(x.device, c)
        hip.check(hip.lib().ym_layernorm_bwd(hip.ptr(dy), hip.ptr(x), hip.ptr(gamma.detach()), float(ctx.eps), x.numel() // c, c,
                                             hip.ptr(dx), hip.ptr(dgamma), hip.ptr(dbeta), _vp(ws), ws.numel(), hip.stream_ptr()),
                  'ym_layernorm_bwd')
        return dx, dgamma, dbeta, None


class PatchMergeLNFn(torch.autograd.Function):
    """PatchMerging up to its LayerNorm: 2x2 gather (zero padded to even H, W) -> [B, H/2, W/2, 4C] -> LayerNorm(4C)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        x = x.contiguous()
        b, h, w, c = x.shape
        out = torch.empty(b, (h + 1) // 2, (w + 1) // 2, 4 * c, device=x.device, dtype=torch.float32)
        hip.patch_merge_layernorm(x, gamma.detach(), beta.detach(), eps, out)
        ctx.save_for_backward(x, gamma)
        ctx.eps, ctx.beta = eps, beta
        return out

    @staticmethod
    def backward(ctx, dy):
        x, gamma = ctx.saved_tensors
        b, h, w, c = x.shape
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        dgamma, dbeta = _grad_slot(gamma, (4 * c,)), _grad_slot(ctx.beta, (4 * c,))
        ws = _ln_ws(x.device, 4 * c)
        hip.check(hip.lib().ym_patch_merge_layernorm_bwd(hip.ptr(dy), hip.ptr(x), b, h, w, c, hip.ptr(gamma.detach()), float(ctx.eps),
                                                         hip.ptr(dx), hip.ptr(dgamma), hip.ptr(dbeta), _vp(ws), ws.numel(),
                                                         hip.stream_ptr()), 'ym_patch_merge_layernorm_bwd')
        return dx, dgamma, dbeta, None


class GeluFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z):
        z = z.contiguous()
        out = torch.empty_like(z)
        hip.check(hip.lib().ym_gelu_fwd(hip.ptr(z), hip.ptr(out), z.numel(), hip.stream_ptr()), 'ym_gelu_fwd')
        ctx.save_for_backward(z)
        return out

    @staticmethod
    def backward(ctx, dy):
        (z,) = ctx.saved_tensors
        dz = torch.empty_like(z)
        hip.check(hip.lib().ym_gelu_bwd(hip.ptr(dy.contiguous()), hip.ptr(z), hip.ptr(dz), z.numel(), hip.stream_ptr()), 'ym_gelu_bwd')
        return dz


class WindowAttentionFn(torch.autograd.Function):
    """softmax(q k^T / sqrt(d) + relative-position bias (+ shift mask)) v per 7x7 window and head, including the pad / roll /
    window partition and their inverses.  `qkv_bias` is an input because padded tokens carry it (the reference pads after norm1),
    so part of its gradient flows through here."""

    @staticmethod
    def forward(ctx, qkv, qkv_bias, table, heads, window, shift):
        qkv = qkv.contiguous()
        b, h, w, c3 = qkv.shape
        c = c3 // 3
        # qkv.bias gets a second gradient from this node (the Linear's ConvBias is the other producer): autograd sums the two on
        # the main stream, so neither may be written from train_engine's side stream (`_side_ok`)
        qkv_bias._ym_multi_producer = True
        out = torch.empty(b, h, w, c, device=qkv.device, dtype=torch.float32)
        hip.swin_window_attention(qkv, qkv_bias.detach(), table.detach(), b, h, w, c, heads, window, shift, out)
        ctx.save_for_backward(qkv, qkv_bias, table)
        ctx.meta = (heads, window, shift)
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, qkv_bias, table = ctx.saved_tensors
        heads, window, shift = ctx.meta
        b, h, w, c3 = qkv.shape
        c = c3 // 3
        dqkv = torch.empty_like(qkv)
        dbias = torch.zeros(c3, device=qkv.device, dtype=torch.float32)
        dtable = _grad_slot(table, tuple(table.shape))
        dtable.zero_()
        hip.check(hip.lib().ym_swin_window_attention_bwd(
            hip.ptr(qkv), hip.ptr(qkv_bias.detach()), hip.ptr(table.detach()), hip.ptr(dout.contiguous()), b, h, w, c, heads, window,
            shift, hip.ptr(dqkv), hip.ptr(dbias), hip.ptr(dtable), hip.stream_ptr()), 'ym_swin_window_attention_bwd')
        return dqkv, dbias, dtable, None, None, None


def _w4(lin):
    """[out, in] Linear weight as the OIHW weight of a 1x1 conv (a view; the optimizer's gradient slot follows it)."""
    w = lin.weight.view(lin.out_features, lin.in_features, 1, 1)
    slot = getattr(lin.weight, '_ym_grad_slot', None)
    if slot is not None and getattr(lin.weight, '_ym_slot_free', False):
        w._ym_grad_slot = slot.view(lin.out_features, lin.in_features, 1, 1)
        w._ym_slot_free = True
        lin.weight._ym_slot_free = False
    w._ym_owner = lin.weight                 # (side-stream bookkeeping lands on the parameter, not on this temporary view)
    return w


def _linear(x, lin, residual=None):
    return ConvBias.apply(x, _w4(lin), lin.bias, 1, 0, ACT_NONE, lin.out_features, residual)


class DropPathAddFn(torch.autograd.Function):
    """shortcut + DropPath(y) (:71-82, :285, :288) as one kernel each way; `rnd` [B] is the raw torch.rand draw."""

    @staticmethod
    def forward(ctx, res, y, rnd, keep):
        res, y = res.contiguous(), y.contiguous()
        out = torch.empty_like(y)
        b = y.shape[0]
        hip.check(hip.lib().ym_drop_path_add(hip.ptr(res), hip.ptr(y), hip.ptr(rnd), keep, hip.ptr(out), b, y.numel() // b,
                                             hip.stream_ptr()), 'ym_drop_path_add')
        ctx.save_for_backward(rnd)
        ctx.keep = keep
        return out

    @staticmethod
    def backward(ctx, dout):
        rnd, = ctx.saved_tensors
        dout = dout.contiguous()
        dy = torch.empty_like(dout)
        b = dout.shape[0]
        hip.check(hip.lib().ym_drop_path_bwd(hip.ptr(dout), hip.ptr(rnd), ctx.keep, hip.ptr(dy), b, dout.numel() // b,
                                             hip.stream_ptr()), 'ym_drop_path_bwd')
        return dout, dy, None, None


def _residual_drop_path(res, y, drop_prob):
    """res + DropPath(y) in train mode with drop_prob > 0: the reference's torch.rand draw (same shape, dtype and device: same random
    stream), everything else fused."""
    rnd = torch.rand((y.shape[0],) + (1,) * (y.ndim - 1), dtype=y.dtype, device=y.device)
    return DropPathAddFn.apply(res, y, rnd.reshape(-1), 1 - drop_prob)


def swin_block(x, blk, heads, window, training=True):
    """SwinTransformerBlock.forward (:234-289) on NHWC tokens [B, h, w, C]."""
    eps = blk.norm1.eps
    dp = float(getattr(blk, 'drop_prob', 0.0))
    fuse = dp == 0. or not training
    n1 = LayerNormFn.apply(x, blk.norm1.weight, blk.norm1.bias, eps)
    qkv = _linear(n1, blk.attn.qkv)
    att = WindowAttentionFn.apply(qkv, blk.attn.qkv.bias, blk.attn.relative_position_bias_table, heads, window, blk.shift_size)
    if fuse:
        x = _linear(att, blk.attn.proj, residual=x)
    else:
        x = _residual_drop_path(x, _linear(att, blk.attn.proj), dp)
    n2 = LayerNormFn.apply(x, blk.norm2.weight, blk.norm2.bias, blk.norm2.eps)
    hid = GeluFn.apply(_linear(n2, blk.mlp.fc1))
    if fuse:
        return _linear(hid, blk.mlp.fc2, residual=x)
    return _residual_drop_path(x, _linear(hid, blk.mlp.fc2), dp)


def swin_backbone_train(bb, x_nhwc4, training=True):
    """SwinTransformer.forward (:500-518): NHWC image (channels padded 3 -> 4) -> the three layer-normed stage maps (NHWC)."""
    pe = bb.patch_embed
    assert x_nhwc4.shape[1] % 4 == 0 and x_nhwc4.shape[2] % 4 == 0
    x = ConvBias.apply(x_nhwc4, pe.proj.weight, pe.proj.bias, 4, 0, ACT_NONE, pe.proj.out_channels, None)
    x = LayerNormFn.apply(x, pe.norm.weight, pe.norm.bias, pe.norm.eps)
    feats = []
    for li, layer in enumerate(bb.layers):
        for blk in layer.blocks:
            x = swin_block(x, blk, bb.heads[li], bb.window_size, training)
        if li in bb.out_norm_indices:
            norm = getattr(bb, f'norm{li}')
            feats.append(LayerNormFn.apply(x, norm.weight, norm.bias, norm.eps))
        if layer.downsample is not None:
            ds = layer.downsample
            merged = PatchMergeLNFn.apply(x, ds.norm.weight, ds.norm.bias, ds.norm.eps)
            x = _linear(merged, ds.reduction)
    return feats
