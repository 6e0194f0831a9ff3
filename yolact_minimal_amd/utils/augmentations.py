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
