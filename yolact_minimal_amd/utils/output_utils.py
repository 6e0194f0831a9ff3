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
import os

import torch

from .. import hip

_anchor_cache = {}
_ws_cache = {}
_plan_cache = {}


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
    # (scratch is per STREAM: requests in flight on different streams must not share it)
    key = (str(device), torch.cuda.current_stream(device).cuda_stream, ncfg.num_anchors, ncfg.num_classes, ncfg.coef_dim, ncfg.max_det)
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
    hp, wp, k = proto_p.shape
    do_crop = not (cfg and getattr(cfg, 'no_crop', False))
    box_c = box_p if box_p.is_contiguous() else box_p.contiguous()
    with torch.cuda.device(device):
        masks = torch.empty(n, img_h, img_w, dtype=torch.float32, device=device)
        box_px = torch.empty(n, 4, dtype=torch.int32, device=device)
        _after_nms_launch(proto_p.contiguous(), coef_p.contiguous(), box_c, None, 1, n, hp, wp, k, img_h, img_w, do_crop, masks, box_px)
    if box_c is not box_p:
        box_p.copy_(box_c)              # keep the reference's in-place scaling visible to the caller
    return ids_p, class_p, box_px, masks


def _after_nms_launch(proto, coefs, boxes, counts, batch, max_det, hp, wp, k, img_h, img_w, do_crop, masks, box_px):
    L = hip.lib()
    nbytes = L.ym_after_nms_batch_workspace_bytes(max_det, hp, wp, img_h, img_w)
    ws = _scratch(proto.device, nbytes) if nbytes else None
    hip.check(L.ym_after_nms_batch(hip.ptr(proto), hip.ptr(coefs), hip.ptr(boxes), hip.ptr(counts, torch.int32) if counts is not None else None,
                                   batch, max_det, hp, wp, k, img_h, img_w, int(do_crop), hip.ptr(masks), hip.ptr(box_px, torch.int32),
                                   ctypes.c_void_p(ws.data_ptr()) if ws is not None else None, nbytes, hip.stream_ptr()),
              'ym_after_nms_batch')


_scratch_bufs = {}


def _scratch(device, nbytes):
    key = (str(device), torch.cuda.current_stream(device).cuda_stream)
    b = _scratch_bufs.get(key)
    if b is None or b.numel() < nbytes:
        b = _scratch_bufs[key] = torch.empty(nbytes, dtype=torch.uint8, device=device)
    return b


# ---- batched post-processing (the reference's nms / after_nms are batch-1 only; eval.py loops) ---------------
class BatchDetections:
    """Device-resident result of `nms_batch` for B images, padded to max_detections rows per image.  `counts` stays on the device;
    `after_nms_batch` consumes it there, and the ONE host read of the batch happens in `after_nms_batch` (or `.split()`)."""

    def __init__(self, counts, ids, scores, boxes, coefs, proto):
        self.counts, self.ids, self.scores, self.boxes, self.coefs, self.proto = counts, ids, scores, boxes, coefs, proto

    def split(self):
        """Per-image 5-tuples exactly as `nms` returns them (five Nones for an image without detections)."""
        out = []
        for b, n in enumerate(self.counts.tolist()):             # the one host read
            out.append((None,) * 5 if n == 0 else
                       (self.ids[b, :n], self.scores[b, :n], self.boxes[b, :n], self.coefs[b, :n], self.proto[b]))
        return out


def nms_batch(class_pred, box_pred, coef_pred, proto_out, anchors, cfg):
    """`nms` (fast_nms) for a whole batch [B, N, *] in one launch set, no host synchronisation.  Per image the result equals
    `nms(class_pred[b:b+1], ...)` (tests/test_gpu_postproc.py::test_batched_postprocessing_equals_per_image)."""
    if not class_pred.is_cuda:
        raise RuntimeError('yolact_minimal_amd.utils.output_utils.nms_batch needs CUDA (HIP) tensors; there is no CPU path.')
    if getattr(cfg, 'traditional_nms', False):
        raise NotImplementedError('nms_batch implements fast_nms; use nms() per image for --traditional_nms')
    if class_pred.dim() != 3:
        raise RuntimeError('nms_batch expects [B, N, C] predictions')
    device = class_pred.device
    batch, n_anchors, n_classes = class_pred.shape
    anchors_t = _anchors_on(anchors, device)
    if anchors_t.shape[0] != n_anchors:
        raise RuntimeError(f'{anchors_t.shape[0]} anchors for {n_anchors} predictions')
    # everything that depends only on the shapes and the thresholds is built once: an eager call spends its time between the
    # caller's line and the first launch HERE (~20 us of Python against ~68 us of device time: bench.post_bench's two columns)
    pkey = (device.index, batch, n_anchors, n_classes, coef_pred.shape[2], cfg.top_k, cfg.max_detections, cfg.nms_score_thre,
            cfg.nms_iou_thre, getattr(cfg, 'img_size', 544))
    plan = _plan_cache.get(pkey)
    L = hip.lib()
    if plan is None:
        ncfg = hip.NmsCfg(n_anchors, n_classes, coef_pred.shape[2], int(cfg.top_k), int(cfg.max_detections),
                          float(cfg.nms_score_thre), float(cfg.nms_iou_thre), float(getattr(cfg, 'img_size', 544)))
        nbytes = L.ym_nms_batch_workspace_bytes(ctypes.byref(ncfg), batch)
        if nbytes == 0:
            raise RuntimeError('ym_nms_batch_workspace_bytes: ' + L.ym_last_error().decode())
        plan = _plan_cache[pkey] = (ncfg, ctypes.byref(ncfg), nbytes)
    ncfg, ncfg_ref, nbytes = plan
    md = ncfg.max_det
    other_device = torch.cuda.current_device() != device.index
    if other_device:
        prev = torch.cuda.current_device()
        torch.cuda.set_device(device)
    try:
        stream = torch.cuda.current_stream().cuda_stream
        key = ('batch', device.index, stream, batch, n_anchors, n_classes)      # (scratch is per STREAM, like _nms_buffers)
        ws = _ws_cache.get(key)
        if ws is None or ws.numel() < nbytes:
            ws = _ws_cache[key] = torch.empty(nbytes, dtype=torch.uint8, device=device)
        counts = torch.empty(batch, dtype=torch.int32, device=device)
        ids = torch.empty(batch, md, dtype=torch.int64, device=device)
        scores = torch.empty(batch, md, dtype=torch.float32, device=device)
        boxes = torch.empty(batch, md, 4, dtype=torch.float32, device=device)
        coefs = torch.empty(batch, md, ncfg.coef_dim, dtype=torch.float32, device=device)
        hip.check(L.ym_detect_fast_nms_batch(hip.ptr(class_pred.contiguous()), hip.ptr(box_pred.contiguous()), hip.ptr(coef_pred.contiguous()),
                                             hip.ptr(anchors_t), ncfg_ref, batch, hip.ptr(counts, torch.int32),
                                             hip.ptr(ids, torch.int64), hip.ptr(scores), hip.ptr(boxes), hip.ptr(coefs),
                                             ctypes.c_void_p(ws.data_ptr()), ws.numel(), ctypes.c_void_p(stream)), 'ym_detect_fast_nms_batch')
    finally:
        if other_device:
            torch.cuda.set_device(prev)
    return BatchDetections(counts, ids, scores, boxes, coefs, proto_out)


def after_nms_batch(dets, img_h, img_w, cfg=None, sync=True):
    """`after_nms` for every image of a `BatchDetections` in one launch set (all images resized to the same img_h x img_w, as in a
    bench / fixed-size serving batch).  Returns a list of per-image 4-tuples like `after_nms` — or, with `sync=False`, the padded
    device tensors (ids, scores, boxes_px, masks, counts) without any host read."""
    device = dets.proto.device
    batch, md = dets.ids.shape
    _, hp, wp, k = dets.proto.shape
    do_crop = not (cfg and getattr(cfg, 'no_crop', False))
    if cfg and getattr(cfg, 'save_lincomb', False):
        raise NotImplementedError('draw_lincomb (visualisation, reference output_utils.py:276-324) is out of scope')
    with torch.cuda.device(device):
        masks = torch.empty(batch, md, img_h, img_w, dtype=torch.float32, device=device)
        box_px = torch.empty(batch, md, 4, dtype=torch.int32, device=device)
        _after_nms_launch(dets.proto.contiguous(), dets.coefs, dets.boxes, dets.counts, batch, md, hp, wp, k, img_h, img_w, do_crop,
                          masks, box_px)
    if not sync:
        return dets.ids, dets.scores, box_px, masks, dets.counts
    out = []
    vt = float(getattr(cfg, 'visual_thre', 0) or 0) if cfg else 0.0
    for b, n in enumerate(dets.counts.tolist()):                  # the ONE host read of the whole batch
        if n == 0:
            out.append((None, None, None, None))
            continue
        r = (dets.ids[b, :n], dets.scores[b, :n], box_px[b, :n], masks[b, :n])
        if vt > 0:                                               # detect.py's score filter (per detection, so it commutes)
            keep = r[1] >= vt
            r = tuple(t[keep] for t in r) if bool(keep.any()) else (None, None, None, None)
        out.append(r)
    return out
