"""`nms` / `after_nms` with the reference's signatures, executed by HIP kernels.

Reference: `/root/reference/utils/output_utils.py` — nms `:126-163`, fast_nms `:11-43`, traditional_nms
`:84-123` (+ `cython_nms.pyx:24-74`), after_nms `:200-233`; box math in `utils/box_utils.py:8-37,117-168`.

Same call shapes and return conventions:
  nms(class_pred, box_pred, coef_pred, proto_out, anchors, cfg) -> (class_ids int64[n], scores f32[n],
      boxes f32[n,4] in 0..1, coefs f32[n,32], proto[Hp,Wp,32])  or five Nones when nothing passes the
      score threshold;  batch size 1 only, like the reference (`.squeeze()` at :127-130).
  after_nms(ids_p, class_p, box_p, coef_p, proto_p, img_h, img_w, cfg=None, img_name=None) ->
      (ids, scores, boxes int32[n,4] pixels, masks f32[n,img_h,img_w] in {0,1}) or four Nones;
      `box_p` is scaled IN PLACE like the reference (:230).
The only host<->device synchronisation is one 4-byte read of the detection count at the end of `nms`
(the reference's boolean-mask gathers synchronise several times per call).
"""
import ctypes

import torch

from .. import hip

_anchor_cache = {}
_ws_cache = {}


def _anchors_on(anchors, device):
    if torch.is_tensor(anchors):
        t = anchors.reshape(-1, 4)
        if t.device != device or t.dtype != torch.float32 or not t.is_contiguous():
            t = t.to(device=device, dtype=torch.float32).contiguous()
        return t
    key = (id(anchors), len(anchors), str(device))
    hit = _anchor_cache.get(key)
    if hit is None or hit[0] is not anchors:
        # the reference rebuilds this tensor from a 74k-float python list on EVERY image (:132-133, ~5 ms)
        t = torch.tensor(anchors, dtype=torch.float32).reshape(-1, 4).to(device)
        _anchor_cache[key] = (anchors, t)
        return t
    return hit[1]


def _nms_buffers(device, ncfg):
    key = (str(device), ncfg.num_anchors, ncfg.num_classes, ncfg.coef_dim, ncfg.max_det)
    b = _ws_cache.get(key)
    if b is None:
        nbytes = hip.lib().ym_nms_workspace_bytes(ctypes.byref(ncfg))
        if nbytes == 0:
            raise RuntimeError('ym_nms_workspace_bytes: ' + hip.lib().ym_last_error().decode())
        b = dict(ws=torch.empty(nbytes, dtype=torch.uint8, device=device),
                 count=torch.zeros(1, dtype=torch.int32, device=device))
        _ws_cache[key] = b
    return b


def nms(class_pred, box_pred, coef_pred, proto_out, anchors, cfg):
    if not class_pred.is_cuda:
        raise RuntimeError('yolact_minimal_amd.utils.output_utils.nms needs CUDA (HIP) tensors; there is no CPU path.')
    class_p = class_pred.squeeze()
    box_p = box_pred.squeeze()
    coef_p = coef_pred.squeeze()
    proto_p = proto_out.squeeze()
    if class_p.dim() != 2:
        raise RuntimeError('nms() handles one image at a time (batch size 1), like the reference.')
    device = class_p.device
    n_anchors, n_classes = class_p.shape
    anchors_t = _anchors_on(anchors, device)
    if anchors_t.shape[0] != n_anchors:
        raise RuntimeError(f'{anchors_t.shape[0]} anchors for {n_anchors} predictions')

    ncfg = hip.NmsCfg(n_anchors, n_classes, coef_p.shape[1], int(cfg.top_k), int(cfg.max_detections),
                      float(cfg.nms_score_thre), float(cfg.nms_iou_thre), float(getattr(cfg, 'img_size', 544)))
    bufs = _nms_buffers(device, ncfg)
    md = ncfg.max_det
    ids = torch.empty(md, dtype=torch.int64, device=device)
    scores = torch.empty(md, dtype=torch.float32, device=device)
    boxes = torch.empty(md, 4, dtype=torch.float32, device=device)
    coefs = torch.empty(md, ncfg.coef_dim, dtype=torch.float32, device=device)
    fn = hip.lib().ym_detect_greedy_nms if getattr(cfg, 'traditional_nms', False) else hip.lib().ym_detect_fast_nms
    ws = bufs['ws']
    with torch.cuda.device(device):
        rc = fn(hip.ptr(class_p.contiguous()), hip.ptr(box_p.contiguous()), hip.ptr(coef_p.contiguous()),
                hip.ptr(anchors_t), ctypes.byref(ncfg), hip.ptr(bufs['count'], torch.int32),
                hip.ptr(ids, torch.int64), hip.ptr(scores), hip.ptr(boxes), hip.ptr(coefs),
                ctypes.c_void_p(ws.data_ptr()), ws.numel(), hip.stream_ptr())
    hip.check(rc, 'ym_detect_nms')
    n = int(bufs['count'].item())       # the single sync of the post-processing path
    if n == 0:
        return None, None, None, None, None
    return ids[:n], scores[:n], boxes[:n], coefs[:n], proto_p


def after_nms(ids_p, class_p, box_p, coef_p, proto_p, img_h, img_w, cfg=None, img_name=None):
    if ids_p is None:
        return None, None, None, None

    if cfg and getattr(cfg, 'visual_thre', 0) > 0:
        keep = class_p >= cfg.visual_thre
        if not bool(keep.any()):
            return None, None, None, None
        ids_p, class_p, box_p, coef_p = ids_p[keep], class_p[keep], box_p[keep], coef_p[keep]

    if cfg and getattr(cfg, 'save_lincomb', False):
        raise NotImplementedError('draw_lincomb (visualisation, reference output_utils.py:276-324) is out of scope')

    device = proto_p.device
    n = coef_p.shape[0]
    hp, wp, _ = proto_p.shape
    do_crop = not (cfg and getattr(cfg, 'no_crop', False))
    box_c = box_p if box_p.is_contiguous() else box_p.contiguous()
    with torch.cuda.device(device):
        soft = torch.empty(n, hp, wp, dtype=torch.float32, device=device)
        hip.mask_assemble(proto_p.contiguous(), coef_p.contiguous(), box_c, soft, do_crop)
        masks = torch.empty(n, img_h, img_w, dtype=torch.float32, device=device)
        hip.mask_resize_binarize(soft, img_h, img_w, masks)
        box_px = torch.empty(n, 4, dtype=torch.int32, device=device)
        hip.boxes_to_pixels(box_c, box_px, max(img_h, img_w))
    if box_c is not box_p:
        box_p.copy_(box_c)              # keep the reference's in-place scaling visible to the caller
    return ids_p, class_p, box_px, masks
