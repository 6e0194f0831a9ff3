"""YOLACT training loss on the device, every term a HIP kernel behind the C-ABI.

Reference: `compute_loss` `/root/reference/modules/yolact.py:166-203`, `category_loss :205-232`, `box_loss :234-239`,
`lincomb_mask_loss :241-291`, `semantic_seg_loss :293-313`, `match`/`encode` `utils/box_utils.py:57-114`.

  match            -> `ym_match_anchors_batch` (one launch, workgroup = image: labels + encoded offsets + matched boxes)
  category + box   -> `ym_class_box_loss` (OHEM ranking by radix select, softmax CE, smooth-L1; gradients in the same pass)
  mask             -> `ym_mask_loss_batch` (f32 MFMA: coefficient x prototype GEMM, sigmoid, crop, BCE, both gradient GEMMs;
                      one launch pair for the batch)
  semantic seg     -> `ym_semantic_loss_batch` (one launch; target built on the fly from the down-sampled gt masks)

The losses are terminal nodes of the graph, so each kernel also writes d(loss)/d(input); the autograd Functions below only
scale those by the incoming gradient.  Same arithmetic and normalisations as the reference (each rank normalises by its LOCAL
positive count).  No host synchronisation anywhere in the loss: the positive counts are produced and consumed on the device,
so the host keeps enqueueing ahead of the GPU through forward, loss and backward (the reference reads them back per image).
"""
import ctypes

import torch

from . import hip


def _vp(t):
    return ctypes.c_void_p(t.data_ptr())


# The loss kernels write d(loss_i)/d(input) in their forward pass; backward only has to scale them by the incoming gradient of
# loss_i.  `Trainer.step` sums the four terms itself (train.py:124: loss_total = loss_c + loss_b + loss_m + loss_s), so that
# gradient is exactly 1 there: inside `unit_loss_grads()` the stored gradients are handed on as they are (no 4 elementwise
# passes over the [B, N, 81] / [B, Hp, Wp, 32] tensors).  Anywhere else (a caller that weights the terms) they are scaled.
_UNIT = [False]


class unit_loss_grads:
    def __enter__(self):
        self.prev, _UNIT[0] = _UNIT[0], True

    def __exit__(self, *exc):
        _UNIT[0] = self.prev
        return False


def _ptr_array(tensors):
    return (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])


def _int_array(values):
    return (ctypes.c_int32 * len(values))(*[int(v) for v in values])


def match(cfg, box_class, anchors, out_offsets, out_conf, out_anchor_box, out_anchor_gt, ws):
    """`match()` for every image of the batch in one launch (workgroup = image), straight into the [B, N, ...] batch buffers."""
    gts = list(box_class)
    for bc in gts:
        hip.ptr(bc)                                                     # contiguous fp32 CUDA tensors [g_i, 5], or raise
    hip.check(hip.lib().ym_match_anchors_batch(
        _ptr_array(gts), _int_array([bc.shape[0] for bc in gts]), len(gts), hip.ptr(anchors), anchors.shape[0],
        float(cfg.pos_iou_thre), float(cfg.neg_iou_thre), hip.ptr(out_offsets), hip.ptr(out_conf, torch.int64),
        hip.ptr(out_anchor_box), hip.ptr(out_anchor_gt, torch.int64), _vp(ws), ws.numel(), hip.stream_ptr()), 'ym_match_anchors_batch')


class _ClassBoxLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, class_p, box_p, offsets, conf, num_pos, conf_alpha, bbox_alpha, ratio):
        b, n, c = class_p.shape
        dev = class_p.device
        cp, bp = class_p.detach().contiguous(), box_p.detach().contiguous()
        dclass, dbox = torch.empty_like(cp), torch.empty_like(bp)
        acc = torch.empty(2, dtype=torch.float64, device=dev)
        ws = torch.empty(hip.lib().ym_loss_workspace_bytes(b, n), dtype=torch.uint8, device=dev)
        hip.check(hip.lib().ym_class_box_loss(
            hip.ptr(cp), hip.ptr(bp), hip.ptr(offsets), hip.ptr(conf, torch.int64), b, n, c, float(conf_alpha),
            float(bbox_alpha), int(ratio), hip.ptr(dclass), hip.ptr(dbox), hip.ptr(num_pos, torch.int32), _vp(acc[0:]),
            _vp(acc[1:]), _vp(ws), ws.numel(), hip.stream_ptr()), 'ym_class_box_loss')
        ctx.save_for_backward(dclass, dbox)
        out = acc.float()
        return out[0], out[1]

    @staticmethod
    def backward(ctx, g_c, g_b):
        dclass, dbox = ctx.saved_tensors
        if _UNIT[0]:
            return dclass, dbox, None, None, None, None, None, None
        return dclass * g_c, dbox * g_b, None, None, None, None, None, None


class _MaskLossFn(torch.autograd.Function):
    """Forward + backward of the mask term for the whole batch in one HIP launch pair (`ym_mask_loss_batch`, workgroup row =
    image): the coefficient x prototype GEMM, sigmoid, crop, BCE and both gradient GEMMs run on the f32 MFMA; the positives'
    coefficients / matched boxes / gt indices are read in place through the anchor indices, and the per-image / total positive
    counts are read from device memory (no host synchronisation).  Autograd only scales the stored gradients."""

    @staticmethod
    def forward(ctx, proto_p, coef_p, anchor_box, anchor_gt, idx, ds_masks, num_pos, alpha_hw):
        b, hp, wp, _ = proto_p.shape
        dev = proto_p.device
        proto_c, coef_c = proto_p.detach().contiguous(), coef_p.detach().contiguous()
        dproto = torch.zeros_like(proto_c)
        dcoef = torch.zeros_like(coef_c)
        acc = torch.zeros(1, dtype=torch.float64, device=dev)
        ws = torch.empty(hip.lib().ym_mask_loss_batch_workspace_bytes(b), dtype=torch.uint8, device=dev)
        hip.ptr(idx, torch.int64), hip.ptr(anchor_box), hip.ptr(anchor_gt, torch.int64), hip.ptr(num_pos, torch.int32)
        cap = idx.shape[1]
        items = (hip.MaskLossItem * b)()
        for i in range(b):
            it = items[i]
            it.proto, it.coef_full = proto_c[i].data_ptr(), coef_c[i].data_ptr()
            it.anchor_box, it.anchor_gt = anchor_box[i].data_ptr(), anchor_gt[i].data_ptr()
            it.gt_masks_ds, it.anchor_idx = hip.ptr(ds_masks[i]).value, idx[i].data_ptr()
            it.n, it.wscale, it.n_dev = cap, 1.0, num_pos.data_ptr() + 4 * i
            it.dproto, it.dcoef_full = dproto[i].data_ptr(), dcoef[i].data_ptr()
        hip.check(hip.lib().ym_mask_loss_batch(items, b, hp, wp, float(alpha_hw), ctypes.c_void_p(num_pos.data_ptr() + 4 * b),
                                               _vp(acc), _vp(ws), ws.numel(), hip.stream_ptr()), 'ym_mask_loss_batch')
        ctx.save_for_backward(dproto, dcoef)
        return (acc * alpha_hw / num_pos[b]).float().reshape(())

    @staticmethod
    def backward(ctx, grad_out):
        dproto, dcoef = ctx.saved_tensors
        if _UNIT[0]:
            return dproto, dcoef, None, None, None, None, None, None
        return dproto * grad_out, dcoef * grad_out, None, None, None, None, None, None


MAX_MASKS_PER_IMAGE = 128        # positives per image the mask-loss kernel holds (cfg.masks_to_train = 100)

_mask_generators = {}


def mask_generator(device):
    """The mask-loss sub-sampling has its OWN device generator (seeded from torch's seed at first use, part of the trainer's
    checkpoint): the reference draws its `randperm` from the CPU generator and only for images over the cap, so a sub-sampling
    draw must not advance the stream Swin's DropPath uses (`torch.rand` on the default device generator)."""
    device = torch.device(device)
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
    g = _mask_generators.get(key)
    if g is None:
        g = _mask_generators[key] = torch.Generator(device=device)
        g.manual_seed((torch.initial_seed() + 0x5EED) % (2 ** 63))
    return g



def lincomb_mask_loss(cfg, pos, anchor_gt, coef_p, proto_p, mask_gt, anchor_box, num_pos=None):
    """`pos`: [B,N] bool, or the matched labels conf_gt (int64, > 0 = positive).  `num_pos`: int32 device tensor [n_0..n_{B-1}, total] of positive counts (`ym_class_box_loss` produces it); computed here
    when absent.  Nothing is read back to the host: the kernel takes the counts from device memory, and the reference's
    `randperm` sub-sampling of > masks_to_train positives (:261-267, a CPU generator there) is a device-side random top-k —
    a uniformly random subset like the reference's, from the device generator instead of the host one."""
    ph, pw = proto_p.shape[1:3]
    b, n = pos.shape
    dev = proto_p.device
    if num_pos is None:
        per = (pos > 0).sum(1)
        num_pos = torch.cat([per, per.sum(0, keepdim=True)]).to(torch.int32)
    if int(cfg.masks_to_train) > MAX_MASKS_PER_IMAGE:
        raise RuntimeError(f'cfg.masks_to_train = {cfg.masks_to_train}: the mask-loss kernel holds at most {MAX_MASKS_PER_IMAGE} '
                           f'positives per image (the reference default is 100)')
    cap = min(int(cfg.masks_to_train), n)
    # positives in anchor order, or — over the cap — the `cap` largest of iid uniform keys among them (ym_select_positives): one
    # launch, no host read; the draw comes from the dedicated generator
    conf = pos if pos.dtype == torch.int64 else pos.to(torch.int64)
    keys = torch.rand(b, n, device=dev, generator=mask_generator(dev))
    idx = torch.empty(b, cap, dtype=torch.int64, device=dev)
    hip.check(hip.lib().ym_select_positives(_vp(conf.contiguous()), _vp(keys), b, n, cap, _vp(num_pos.contiguous()), _vp(idx),
                                            hip.stream_ptr()), 'ym_select_positives')
    ds_masks = []
    for i in range(b):
        g = mask_gt[i].shape[0]
        ds = torch.empty(g, ph, pw, device=dev, dtype=torch.float32)                 # bilinear(align_corners=False) then > 0.5
        hip.mask_resize_binarize(mask_gt[i].contiguous().float(), ph, pw, ds)
        ds_masks.append(ds.reshape(g, ph * pw))
    return _MaskLossFn.apply(proto_p, coef_p, anchor_box.contiguous().float(), anchor_gt.contiguous(), idx, ds_masks,
                             num_pos.contiguous(), cfg.mask_alpha / ph / pw)


class _SemanticLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, seg_p, mask_gt, box_class, coeff):
        b, nc, mh, mw = seg_p.shape
        dev = seg_p.device
        x = seg_p.detach().permute(0, 2, 3, 1)                                        # NHWC view; the conv output is NHWC
        pitch = x.stride(2)
        if not (x.stride(3) == 1 and pitch >= nc and x.stride(1) == mw * pitch and x.stride(0) == mh * mw * pitch):
            x, pitch = x.contiguous(), nc
        dseg = torch.empty(b, mh, mw, pitch, device=dev, dtype=torch.float32)
        acc = torch.zeros(1, dtype=torch.float64, device=dev)
        gs = [int(m.shape[0]) for m in mask_gt]
        ds = torch.empty(max(sum(gs), 1), mh, mw, device=dev, dtype=torch.float32)        # every image's down-sampled gt masks
        cls = torch.cat([bc[:, -1] for bc in box_class]).long() if sum(gs) else torch.zeros(1, dtype=torch.int64, device=dev)
        ds_i, cls_i, at = [], [], 0
        for i in range(b):
            if gs[i]:
                hip.mask_resize_binarize(mask_gt[i].contiguous().float(), mh, mw, ds[at:at + gs[i]])
            ds_i.append(ds[at:at + gs[i]] if gs[i] else ds[:0])
            cls_i.append(cls[at:at + gs[i]] if gs[i] else cls[:0])
            at += gs[i]
        hip.check(hip.lib().ym_semantic_loss_batch(
            _vp(x), b, mh * mw, pitch, nc, _ptr_array(ds_i), _ptr_array(cls_i), 1, _int_array(gs), float(coeff), _vp(dseg), _vp(acc),
            hip.stream_ptr()), 'ym_semantic_loss_batch')
        ctx.save_for_backward(dseg)
        ctx.nc = nc
        return acc.float().reshape(())

    @staticmethod
    def backward(ctx, grad_out):
        (dseg,) = ctx.saved_tensors
        return (dseg[..., :ctx.nc] * grad_out).permute(0, 3, 1, 2), None, None, None


def semantic_seg_loss(cfg, seg_p, mask_gt, box_class):
    b, _, mh, mw = seg_p.shape
    return _SemanticLossFn.apply(seg_p, mask_gt, box_class, cfg.semantic_alpha / mh / mw / b)


def compute_loss(cfg, anchors, class_p, box_p, coef_p, proto_p, seg_p, box_class, mask_gt):
    device = class_p.device
    b, n = box_p.shape[:2]
    offsets = torch.empty(b, n, 4, device=device)
    conf_gt = torch.empty(b, n, dtype=torch.int64, device=device)
    anchor_box = torch.empty(b, n, 4, device=device)
    anchor_gt = torch.empty(b, n, dtype=torch.int64, device=device)
    num_pos = torch.empty(b + 1, dtype=torch.int32, device=device)
    ws = torch.empty(b * n * 4, dtype=torch.uint8, device=device)
    anchors = anchors.contiguous().float()
    match(cfg, [bc.contiguous().float() for bc in box_class], anchors, offsets, conf_gt, anchor_box, anchor_gt, ws)
    loss_c, loss_b = _ClassBoxLossFn.apply(class_p, box_p, offsets, conf_gt, num_pos, cfg.conf_alpha, cfg.bbox_alpha, 3)
    loss_m = lincomb_mask_loss(cfg, conf_gt, anchor_gt, coef_p, proto_p, mask_gt, anchor_box, num_pos)   # counts stay on the device
    loss_s = semantic_seg_loss(cfg, seg_p, mask_gt, box_class)
    return loss_c, loss_b, loss_m, loss_s
