"""CPU restatement of the reference's `train_aug` pixel pipeline (utils/augmentations.py:9-252).  TEST INFRASTRUCTURE ONLY.

The reference's chain calls cv2 (absent from this image) in two places: `cv2.cvtColor` BGR<->HSV on float32 images (:68,71) and
`cv2.resize` (bilinear) in `multi_scale_resize` (:175,181).  Both are restated from OpenCV's documented float32 behaviour
(HSV: V = max, S = (V - min) / V, H in degrees [0, 360); resize: half-pixel centres, the same equivalence the reference notes at
utils/output_utils.py:225) — PARITY UNPINNED by cv2 for those two steps.  Everything else (mirror, crop, pads, box bookkeeping,
random-call order) is numpy and is pinned bit-for-bit against the real reference functions by oracle/make_golden_augment.py.
`apply_plan` executes a plan (the decisions drawn by the product's host code or by the reference) stage by stage, exactly in
the reference's order, on numpy / torch CPU arrays.
"""
import numpy as np
import torch
import torch.nn.functional as F

NORM_MEAN = np.array([103.94, 116.78, 123.68], dtype=np.float32)     # config.py:66-67 (BGR)
NORM_STD = np.array([57.38, 57.12, 58.40], dtype=np.float32)


def bgr_to_hsv(img):
    """cv2.cvtColor(float32 BGR in 0..255, COLOR_BGR2HSV): H in [0,360), S in [0,1], V in 0..255."""
    b, g, r = img[..., 0], img[..., 1], img[..., 2]
    v = np.maximum(np.maximum(b, g), r)
    mn = np.minimum(np.minimum(b, g), r)
    diff = v - mn
    s = np.where(v > 0, diff / np.where(v > 0, v, 1), 0).astype(np.float32)
    safe = np.where(diff > 0, diff, 1)
    h = np.where(v == r, (g - b) / safe, np.where(v == g, 2.0 + (b - r) / safe, 4.0 + (r - g) / safe)) * 60.0
    h = np.where(diff > 0, h, 0.0)
    h = np.where(h < 0, h + 360.0, h).astype(np.float32)
    return np.stack([h, s, v], -1).astype(np.float32)


def hsv_to_bgr(hsv):
    h, s, v = hsv[..., 0], hsv[..., 1], hsv[..., 2]
    h = np.where(h < 0, h + 360.0, h)
    h = np.where(h >= 360.0, h - 360.0, h) / 60.0
    i = np.floor(h)
    f = h - i
    i = i.astype(np.int32) % 6
    p, q, t = v * (1 - s), v * (1 - s * f), v * (1 - s * (1 - f))
    r = np.choose(i, [v, q, p, p, t, v])
    g = np.choose(i, [t, v, v, q, p, p])
    b = np.choose(i, [p, p, t, v, v, q])
    return np.stack([b, g, r], -1).astype(np.float32)


def photometric(img, plan):
    img = img.astype(np.float32).copy()
    if plan.brightness is not None:
        img = np.clip(img + np.float32(plan.brightness), 0., 255.)
    if plan.contrast is not None:
        img = np.clip(img * np.float32(plan.contrast), 0., 255.)
    hsv = bgr_to_hsv(img)
    hsv[..., 1] *= np.float32(plan.saturation)
    hsv[..., 0] += np.float32(plan.hue)
    hsv[..., 0][hsv[..., 0] > 360.0] -= 360.0
    hsv[..., 0][hsv[..., 0] < 0.0] += 360.0
    return np.clip(hsv_to_bgr(hsv), 0., 255.)


def resize_bilinear(x_hwc, size):
    t = torch.from_numpy(np.ascontiguousarray(x_hwc)).permute(2, 0, 1)[None].float()
    return F.interpolate(t, (size, size), mode='bilinear', align_corners=False)[0].permute(1, 2, 0).numpy()


def resize_bilinear_u8(x_hwc, size):
    """cv2.resize(uint8 HxWxC, (size, size)) with the default INTER_LINEAR: OpenCV's 8-bit path is FIXED POINT (imgproc/resize.cpp:
    horizontal pass with coefficients saturate_cast<short>(w * 2048) into int32, vertical pass
    `(((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2`), so a {0,1} instance mask comes out {0,1}.  The reference
    hits it when the cropped sample is already square: pad_to_square (:138-141) then returns the uint8 masks of annToMask untouched
    and multi_scale_resize (:180-181) resizes them as uint8; every other sample goes through the float32 pad buffer (:147).
    Restated from OpenCV's published algorithm — cv2 is absent here: PARITY UNPINNED for this primitive."""
    x = np.ascontiguousarray(x_hwc)
    assert x.dtype == np.uint8 and x.ndim == 3
    h, w, _ = x.shape

    def taps(n_src, n_dst, clamp_weights):
        d = np.arange(n_dst, dtype=np.float64)
        f = ((d + 0.5) * (n_src / n_dst) - 0.5).astype(np.float32)
        i0 = np.floor(f).astype(np.int64)
        fr = (f - i0.astype(np.float32)).astype(np.float32)
        if clamp_weights:                       # x: weights forced to (1, 0) at the borders, taps inside the row
            lo, hi = i0 < 0, i0 >= n_src - 1
            fr = np.where(lo | hi, np.float32(0), fr)
            i0 = np.where(lo, 0, np.where(hi, n_src - 1, i0))
            i1 = np.minimum(i0 + 1, n_src - 1)
        else:                                   # y: weights kept, row indices clipped
            i1 = np.clip(i0 + 1, 0, n_src - 1)
            i0 = np.clip(i0, 0, n_src - 1)
        w0 = np.rint((np.float32(1) - fr) * np.float32(2048)).astype(np.int64)      # saturate_cast<short> = round half to even
        w1 = np.rint(fr * np.float32(2048)).astype(np.int64)
        return i0, i1, w0, w1

    x0, x1, a0, a1 = taps(w, size, True)
    y0, y1, b0, b1 = taps(h, size, False)
    xi = x.astype(np.int64)
    hor = xi[:, x0, :] * a0[None, :, None] + xi[:, x1, :] * a1[None, :, None]             # [h][size][c], scale 2048
    s0, s1 = hor[y0] >> 4, hor[y1] >> 4
    out = (((b0[:, None, None] * s0) >> 16) + ((b1[:, None, None] * s1) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


def apply_plan(img, masks, plan):
    """img [H,W,3] BGR (uint8 / float), masks [n,H,W] -> (image [3,S,S] float32 normalised RGB, masks [k,S,S] float32)."""
    img = photometric(img, plan)
    u8_square = masks.dtype == np.uint8 and plan.crop[2] == plan.crop[3]      # the reference keeps uint8 masks here (resize_bilinear_u8)
    masks_u8 = masks
    masks = masks.astype(np.float32)
    if plan.mirror:
        img, masks = img[:, ::-1], masks[:, :, ::-1]
    cx, cy, cw, ch = plan.crop
    img, masks = img[cy:cy + ch, cx:cx + cw], masks[:, cy:cy + ch, cx:cx + cw]
    q, (px, py) = plan.square, plan.pad
    sq = np.zeros((q, q, 3), np.float32)
    sq[:, :, :] = NORM_MEAN
    sm = np.zeros((masks.shape[0], q, q), np.float32)
    sq[py:py + ch, px:px + cw] = img
    sm[:, py:py + ch, px:px + cw] = masks
    r = plan.resize
    img = resize_bilinear(sq, r)
    masks = resize_bilinear(sm.transpose(1, 2, 0), r).transpose(2, 0, 1) if sm.shape[0] else sm[:, :r, :r]
    if u8_square and sm.shape[0]:
        mu = masks_u8[:, :, ::-1] if plan.mirror else masks_u8
        mu = mu[:, cy:cy + ch, cx:cx + cw]                                       # (q == cw == ch, no pad)
        masks = resize_bilinear_u8(mu.transpose(1, 2, 0), r).transpose(2, 0, 1).astype(np.float32)
    s = plan.size
    if plan.final_pad is not None:
        fx, fy = plan.final_pad
        out = np.zeros((s, s, 3), np.float32)
        out[:, :, :] = NORM_MEAN
        om = np.zeros((masks.shape[0], s, s), np.float32)
        out[fy:fy + r, fx:fx + r] = img
        om[:, fy:fy + r, fx:fx + r] = masks
        img, masks = out, om
    elif plan.final_crop is not None:
        fx, fy = plan.final_crop
        img, masks = img[fy:fy + s, fx:fx + s], masks[:, fy:fy + s, fx:fx + s]
    masks = masks[plan.keep]
    img = (img - NORM_MEAN) / NORM_STD
    img = img[:, :, (2, 1, 0)].transpose(2, 0, 1)
    return np.ascontiguousarray(img, dtype=np.float32), np.ascontiguousarray(masks, dtype=np.float32)
