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
