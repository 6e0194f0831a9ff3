"""`val_aug` on the GPU (SURVEY.md §8f "next" row 1).

Reference: `/root/reference/utils/augmentations.py:219-227` (`pad_to_square :138-165`, `multi_scale_resize :168-189`
= `cv2.resize`, `normalize_and_toRGB :212-216`).  Same call shape — `val_aug(img, val_size)` with an HWC BGR image —
but `img` is a CUDA tensor (uint8 or float32) and the result is a CUDA float32 tensor `[3, S, S]`; one HIP launch, the
padded square is never materialised.  cv2 is not present in the build image, so this row is pinned against a torch
restatement (pad + `F.interpolate(align_corners=False)`, the equivalence the reference itself notes at
`utils/output_utils.py:225`), not against cv2 — "parity unpinned by the reference".
"""
import ctypes

import torch

from .. import hip
from ..config import norm_mean, norm_std

_MEAN = (ctypes.c_float * 3)(*[float(v) for v in norm_mean])
_STD = (ctypes.c_float * 3)(*[float(v) for v in norm_std])


def val_aug(img, val_size):
    if not (torch.is_tensor(img) and img.is_cuda):
        raise RuntimeError('yolact_minimal_amd.utils.augmentations.val_aug expects a CUDA tensor (HWC, BGR)')
    if img.dtype not in (torch.uint8, torch.float32) or img.dim() != 3 or img.shape[2] != 3:
        raise RuntimeError('val_aug: image must be [H, W, 3] uint8 or float32')
    img = img.contiguous()
    h, w, _ = img.shape
    out = torch.empty(3, val_size, val_size, dtype=torch.float32, device=img.device)
    with torch.cuda.device(img.device):
        hip.check(hip.lib().ym_val_preprocess(ctypes.c_void_p(img.data_ptr()), int(img.dtype == torch.uint8), h, w, val_size, _MEAN,
                                              _STD, hip.ptr(out), hip.stream_ptr()), 'ym_val_preprocess')
    return out


# ------------------------------------------------------------------------------------------------------------------------
# train_aug.  Reference: utils/augmentations.py:9-252 — photometric_distort :60-77, random_mirror :9-16,
# random_crop / crop :80-136, pad_to_square(during_training) :138-165, multi_scale_resize :168-189, to_train_size :192-209,
# clip_box / remove_small_box / to_01_box :19-36, normalize_and_toRGB :212-216, train_aug :230-252.
#
# The reference runs this chain on the CPU in DataLoader workers: ten numpy / cv2 passes over the image and over every mask.
# Here the RANDOM DECISIONS and the box bookkeeping stay on the host (`sample_train_aug`, a few hundred scalar operations, same
# `random` call order as the reference so that a seeded run makes the same choices), and the pixel work is ONE HIP launch for
# the image and one for the masks: every output pixel walks the chain backwards (train-size pad/crop -> bilinear resize ->
# pad-to-square -> crop -> mirror) to its four source texels, applies the photometric distortion to them, blends, normalises.
# cv2 is absent from the image: resize / HSV follow OpenCV's documented float32 formulas and are pinned against a torch / numpy
# restatement (oracle/augment_ref.py), not against cv2 — "parity unpinned by the reference" for those two steps; the random
# decisions and the geometry are pinned against the reference's own numpy functions (oracle/make_golden_augment.py).
# ------------------------------------------------------------------------------------------------------------------------
import random as _random


class AugPlan:
    """Everything `train_aug` decided for one sample.  Geometry is a chain of integer offsets around ONE bilinear resize:
    original --mirror--> --crop (cx, cy, cw, ch)--> --pad to square q (px, py)--> --resize q -> r--> --final: pad (fx, fy)
    into s or crop (fx, fy) of s--> [s, s]."""
    __slots__ = ('brightness', 'contrast', 'saturation', 'hue', 'mirror', 'crop', 'square', 'pad', 'resize', 'final_pad',
                 'final_crop', 'size', 'boxes', 'labels', 'keep')


def _crop_boxes(rng, ori_h, crop_h, ori_w, crop_w, boxes, labels, keep_idx, keep_ratio=0.3):
    """The rejection loop of `crop` (:80-125) on boxes only -> (x1, y1, boxes, labels, keep_idx) or None after 1000 tries."""
    import numpy as np
    areas = (boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1])
    for _ in range(1000):
        x1 = rng.randint(0, ori_w - crop_w)
        y1 = rng.randint(0, ori_h - crop_h)
        mnx = np.maximum(np.float64(x1), boxes[:, 0])
        mny = np.maximum(np.float64(y1), boxes[:, 1])
        mxx = np.minimum(np.float64(x1 + crop_w), boxes[:, 2])
        mxy = np.minimum(np.float64(y1 + crop_h), boxes[:, 3])
        inter = np.clip(mxx - mnx, 0, 10000) * np.clip(mxy - mny, 0, 10000)
        keep = (inter / areas) > keep_ratio
        if keep.any():
            nb = np.stack([mnx, mny, mxx, mxy], 1)[keep]
            nb[:, [0, 2]] -= x1
            nb[:, [1, 3]] -= y1
            return x1, y1, nb, labels[keep], keep_idx[keep]
    return None


def sample_train_aug(img_h, img_w, boxes, labels, train_size, rng=_random):
    """Draw the random decisions of `train_aug` in the reference's order and carry the boxes through them.
    boxes [n,4] pixels (x1,y1,x2,y2) — float64 like the dataset's `np.array(box_list)` (utils/coco.py:97), labels [n].
    Returns an AugPlan, or None where the reference returns Nones."""
    import numpy as np
    p = AugPlan()
    boxes = np.array(boxes, dtype=np.float64).reshape(-1, 4)
    labels = np.array(labels)
    keep_idx = np.arange(boxes.shape[0])
    # photometric_distort (:60-77)
    p.brightness = rng.uniform(-32, 32) if rng.randint(0, 1) else None
    p.contrast = rng.uniform(0.7, 1.3) if rng.randint(0, 1) else None
    p.saturation = rng.uniform(0.7, 1.3)
    p.hue = rng.uniform(-15., 15.)
    # random_mirror (:9-16)
    p.mirror = bool(rng.randint(0, 1))
    if p.mirror:
        boxes[:, 0::2] = img_w - boxes[:, 2::-2]
    # random_crop (:128-136)
    h, w = img_h, img_w
    p.crop = (0, 0, w, h)
    if not rng.randint(0, 1):
        crop_h = int(rng.uniform(0.6, 1) * h)
        crop_w = int(rng.uniform(0.6, 1) * w)
        r = _crop_boxes(rng, h, crop_h, w, crop_w, boxes, labels, keep_idx)
        if r is None:
            return None
        x1, y1, boxes, labels, keep_idx = r
        p.crop = (x1, y1, crop_w, crop_h)
        h, w = crop_h, crop_w
    # pad_to_square, during_training (:138-163)
    q = max(h, w)
    px = py = 0
    if h < w:
        py = rng.randint(0, w - h)
        boxes[:, [1, 3]] += py
    if h > w:
        px = rng.randint(0, h - w)
        boxes[:, [0, 2]] += px
    p.square, p.pad = q, (px, py)
    # multi_scale_resize (:168-186)
    r_size = rng.randint(8, 24) * 32
    boxes *= r_size / q
    p.resize = r_size
    # to_train_size (:192-209)
    p.final_pad = p.final_crop = None
    if r_size < train_size:
        fy = rng.randint(0, train_size - r_size)
        fx = rng.randint(0, train_size - r_size)
        boxes[:, [1, 3]] += fy
        boxes[:, [0, 2]] += fx
        p.final_pad = (fx, fy)
    elif r_size > train_size:
        r = _crop_boxes(rng, r_size, train_size, r_size, train_size, boxes, labels, keep_idx)
        if r is None:
            return None
        fx, fy, boxes, labels, keep_idx = r
        p.final_crop = (fx, fy)
    p.size = train_size
    # clip_box, remove_small_box, to_01_box (:19-36)
    boxes[:, [0, 2]] = np.clip(boxes[:, [0, 2]], 0, train_size - 1)
    boxes[:, [1, 3]] = np.clip(boxes[:, [1, 3]], 0, train_size - 1)
    big = (boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1]) > 20
    boxes, labels, keep_idx = boxes[big], labels[big], keep_idx[big]
    if boxes.shape[0] == 0:
        return None
    boxes[:, [0, 2]] /= train_size
    boxes[:, [1, 3]] /= train_size
    p.boxes, p.labels, p.keep = boxes, labels, keep_idx
    return p


def plan_to_c(plan, img_h, img_w):
    """AugPlan -> the C-ABI `ym_aug_plan`."""
    c = hip.AugPlanC()
    c.H, c.W, c.mirror = img_h, img_w, int(plan.mirror)
    c.cx, c.cy, c.cw, c.ch = plan.crop
    c.q, (c.px, c.py), c.r, c.S = plan.square, plan.pad, plan.resize, plan.size
    c.final_mode, c.fx, c.fy = 0, 0, 0
    if plan.final_pad is not None:
        c.final_mode, (c.fx, c.fy) = 1, plan.final_pad
    elif plan.final_crop is not None:
        c.final_mode, (c.fx, c.fy) = 2, plan.final_crop
    c.has_brightness, c.brightness = int(plan.brightness is not None), float(plan.brightness or 0.0)
    c.has_contrast, c.contrast = int(plan.contrast is not None), float(plan.contrast if plan.contrast is not None else 1.0)
    c.saturation, c.hue = float(plan.saturation), float(plan.hue)
    for i in range(3):
        c.mean[i], c.std[i] = float(norm_mean[i]), float(norm_std[i])
    return c


def apply_train_aug(img, masks, plan):
    """The pixel half of `train_aug` on the device: img [H,W,3] BGR (uint8 / float32) and masks [n,H,W] (uint8 / float32) CUDA
    tensors + an AugPlan -> (image [3,S,S] float32 normalised RGB, masks [k,S,S] float32), two HIP launches."""
    if not (torch.is_tensor(img) and img.is_cuda and masks.is_cuda):
        raise RuntimeError('yolact_minimal_amd.utils.augmentations.train_aug expects CUDA tensors (there is no CPU path)')
    if img.dtype not in (torch.uint8, torch.float32) or masks.dtype not in (torch.uint8, torch.float32):
        raise RuntimeError('train_aug: image / masks must be uint8 or float32')
    img, masks = img.contiguous(), masks.contiguous()
    h, w, _ = img.shape
    c = plan_to_c(plan, h, w)
    s, k = plan.size, len(plan.keep)
    out = torch.empty(3, s, s, dtype=torch.float32, device=img.device)
    mout = torch.empty(k, s, s, dtype=torch.float32, device=img.device)
    keep = torch.as_tensor(plan.keep, dtype=torch.int32).to(img.device)
    with torch.cuda.device(img.device):
        hip.check(hip.lib().ym_train_aug_image(ctypes.c_void_p(img.data_ptr()), int(img.dtype == torch.uint8), ctypes.byref(c),
                                               hip.ptr(out), hip.stream_ptr()), 'ym_train_aug_image')
        hip.check(hip.lib().ym_train_aug_masks(ctypes.c_void_p(masks.data_ptr()), int(masks.dtype == torch.uint8),
                                               hip.ptr(keep, torch.int32), k, ctypes.byref(c), hip.ptr(mout), hip.stream_ptr()),
                  'ym_train_aug_masks')
    return out, mout


def train_aug(img, masks, boxes, labels, train_size, rng=_random):
    """Reference call shape (`train_aug(img, masks, boxes, labels, train_size)`, utils/augmentations.py:230-252) with CUDA
    tensors for `img` / `masks` and host arrays for `boxes` (pixels) / `labels`: returns (img [3,S,S], masks [k,S,S], boxes [k,4]
    in 0..1, labels [k]) — the arrays as numpy float64 / like the reference — or four Nones when the sample is rejected."""
    plan = sample_train_aug(img.shape[0], img.shape[1], boxes, labels, train_size, rng)
    if plan is None:
        return None, None, None, None
    out, mout = apply_train_aug(img, masks, plan)
    return out, mout, plan.boxes, plan.labels
