"""Anchor generation (host side).  Reference: `/root/reference/utils/box_utils.py:86-101`.

The reference builds a flat python list in float64 and lets `torch.tensor` round it to fp32; the
same arithmetic order is kept here (`(i + 0.5) / conv_w`, `scale * sqrt(ar) / img_size`,
`scale / sqrt(ar) / img_size`) so the fp32 anchors are bit-identical.
"""
from math import sqrt


def make_anchors(cfg, conv_h, conv_w, scale):
    roots = [sqrt(ar) for ar in cfg.aspect_ratios]
    out = []
    for j in range(conv_h):
        cy = (j + 0.5) / conv_h
        for i in range(conv_w):
            cx = (i + 0.5) / conv_w
            for r in roots:
                out += [cx, cy, scale * r / cfg.img_size, scale / r / cfg.img_size]
    return out


def _f32(t):
    import torch
    return t.to(torch.float32).contiguous()


def box_iou(box_a, box_b):
    """[n,4] x [g,4] corner boxes -> IoU [n,g] on the device (`ym_box_iou`); reference `utils/box_utils.py:8-37`
    (2-D inputs only: the evaluation call site, `utils/common_utils.py:184`)."""
    import torch
    from .. import hip
    a, b = _f32(box_a), _f32(box_b)
    if not a.is_cuda:
        raise RuntimeError('yolact_minimal_amd has no CPU path: box_iou needs CUDA/HIP tensors')
    out = torch.empty(a.shape[0], b.shape[0], device=a.device, dtype=torch.float32)
    if out.numel():
        hip.check(hip.lib().ym_box_iou(hip.ptr(a), a.shape[0], hip.ptr(b), b.shape[0], hip.ptr(out), hip.stream_ptr()), 'ym_box_iou')
    return out


def mask_iou(mask1, mask2, to_cpu=True):
    """IoU of binary masks [n, H*W] x [g, H*W] -> [n, g]; reference `utils/box_utils.py:189-200` (which returns `.cpu()`).
    The {0,1} fp32 matmul is evaluated as popcounts of bit rows (`ym_mask_iou`): exact, every mask read once."""
    import ctypes
    import torch
    from .. import hip
    a, b = _f32(mask1), _f32(mask2)
    if not a.is_cuda:
        raise RuntimeError('yolact_minimal_amd has no CPU path: mask_iou needs CUDA/HIP tensors')
    n, g, p = a.shape[0], b.shape[0], a.shape[1]
    out = torch.empty(n, g, device=a.device, dtype=torch.float32)
    if n and g:
        nb = hip.lib().ym_mask_iou_workspace_bytes(n, g, p)
        ws = torch.empty(nb, device=a.device, dtype=torch.uint8)
        hip.check(hip.lib().ym_mask_iou(hip.ptr(a), n, hip.ptr(b), g, p, hip.ptr(out), ctypes.c_void_p(ws.data_ptr()), nb,
                                        hip.stream_ptr()), 'ym_mask_iou')
    return out.cpu() if to_cpu else out
