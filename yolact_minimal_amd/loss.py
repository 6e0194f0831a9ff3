"""YOLACT training loss on device tensors.

Reference: `compute_loss` `/root/reference/modules/yolact.py:166-203`, `category_loss :205-232`, `box_loss :234-239`,
`lincomb_mask_loss :241-291`, `semantic_seg_loss :293-313`, `match`/`encode` `utils/box_utils.py:57-114`.

These are the small, data-dependent bookkeeping steps on the `[B, 18525, *]` head outputs (a few MFLOP); they are
expressed with torch tensor ops on the GPU so that autograd can carry the gradient into the HIP backward
kernels.  Same arithmetic and normalisations as the reference (loss per rank normalised by its LOCAL positive count);
the python loop over ground-truth boxes in `match` and the CPU `randperm` are kept on purpose for parity.
"""
import ctypes

import torch
import torch.nn.functional as F

from . import hip


def box_iou(box_a, box_b):
    """[A,4] x [B,4] corner boxes -> [A,B]; inter / (area_a + area_b - inter) (utils/box_utils.py:8-37)."""
    lo = torch.max(box_a[:, None, :2], box_b[None, :, :2])
    hi = torch.min(box_a[:, None, 2:], box_b[None, :, 2:])
    wh = torch.clamp(hi - lo, min=0)
    inter = wh[..., 0] * wh[..., 1]
    area_a = ((box_a[:, 2] - box_a[:, 0]) * (box_a[:, 3] - box_a[:, 1]))[:, None]
    area_b = ((box_b[:, 2] - box_b[:, 0]) * (box_b[:, 3] - box_b[:, 1]))[None, :]
    return inter / (area_a + area_b - inter)


def encode(matched, priors):
    cxcy = ((matched[:, :2] + matched[:, 2:]) / 2 - priors[:, :2]) / (0.1 * priors[:, 2:])
    wh = torch.log((matched[:, 2:] - matched[:, :2]) / priors[:, 2:]) / 0.2
    return torch.cat([cxcy, wh], 1)


def match(cfg, box_gt, anchors, class_gt):
    corners = torch.cat((anchors[:, :2] - anchors[:, 2:] / 2, anchors[:, :2] + anchors[:, 2:] / 2), 1)
    overlaps = box_iou(box_gt, corners)                       # [g, N]
    gt_best_anchor = overlaps.max(1)[1]
    anchor_best, anchor_gt = overlaps.max(0)
    anchor_best.index_fill_(0, gt_best_anchor, 2)
    for j in range(gt_best_anchor.shape[0]):                  # sequential on purpose: the LAST gt wins a shared anchor
        anchor_gt[gt_best_anchor[j]] = j
    matched = box_gt[anchor_gt]
    conf = class_gt[anchor_gt] + 1
    conf[anchor_best < cfg.pos_iou_thre] = -1
    conf[anchor_best < cfg.neg_iou_thre] = 0
    return encode(matched, anchors), conf, matched, anchor_gt


def crop(masks, boxes, padding=1):
    """masks [h,w,n] zeroed outside each (padded) box window — utils/box_utils.py:117-168."""
    h, w, n = masks.shape

    def span(a, b, size):
        a, b = a * size, b * size
        return torch.clamp(torch.min(a, b) - padding, min=0), torch.clamp(torch.max(a, b) + padding, max=size)
    x1, x2 = span(boxes[:, 0], boxes[:, 2], w)
    y1, y2 = span(boxes[:, 1], boxes[:, 3], h)
    xs = torch.arange(w, device=masks.device, dtype=x1.dtype).view(1, -1, 1)
    ys = torch.arange(h, device=masks.device, dtype=x1.dtype).view(-1, 1, 1)
    inside = (xs >= x1.view(1, 1, -1)) & (xs < x2.view(1, 1, -1)) & (ys >= y1.view(1, 1, -1)) & (ys < y2.view(1, 1, -1))
    return masks * inside.float()


def category_loss(cfg, class_p, conf_gt, pos, ratio=3):
    nc = cfg.num_classes
    flat = class_p.reshape(-1, nc)
    mx = flat.max()
    mark = torch.log(torch.sum(torch.exp(flat - mx), 1)) + mx - flat[:, 0]
    mark = mark.reshape(class_p.shape[0], -1)
    mark[pos] = 0
    mark[conf_gt < 0] = 0
    _, idx = mark.sort(1, descending=True)
    _, rank = idx.sort(1)
    num_pos = pos.long().sum(1, keepdim=True)
    num_neg = torch.clamp(ratio * num_pos, max=pos.shape[1] - 1)
    neg = rank < num_neg.expand_as(rank)
    neg[pos] = 0
    neg[conf_gt < 0] = 0
    sel = pos | neg
    return cfg.conf_alpha * F.cross_entropy(class_p[sel].reshape(-1, nc), conf_gt[sel], reduction='sum') / num_pos.sum()


def box_loss(cfg, box_p, offsets, pos):
    return cfg.bbox_alpha * F.smooth_l1_loss(box_p[pos, :], offsets[pos, :], reduction='sum') / pos.sum()


class _MaskLossFn(torch.autograd.Function):
    """Forward + backward of the mask term in one HIP pass per image (`ym_mask_loss_fwd_bwd`): the coefficient x prototype
    GEMM, sigmoid, crop, BCE and both gradient GEMMs run on the f32 MFMA; autograd only scales the stored gradients."""

    @staticmethod
    def forward(ctx, proto_p, coef_p, per_image, coeff):
        b, hp, wp, _ = proto_p.shape
        dev = proto_p.device
        proto_c, coef_c = proto_p.detach().contiguous(), coef_p.detach().contiguous()
        dproto = torch.zeros_like(proto_c)
        dcoef = torch.zeros_like(coef_c)
        acc = torch.zeros(1, dtype=torch.float64, device=dev)
        ws_bytes = hip.lib().ym_mask_loss_workspace_bytes()
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        for i, item in enumerate(per_image):
            if item is None:
                continue
            idx, gt_idx, boxes, dsmask, wscale = item
            cpos = coef_c[i][idx].contiguous()
            hip.check(hip.lib().ym_mask_loss_fwd_bwd(
                hip.ptr(proto_c[i]), hip.ptr(cpos), hip.ptr(boxes), hip.ptr(gt_idx, torch.int32), hip.ptr(dsmask),
                hip.ptr(idx, torch.int64), idx.shape[0], hp, wp, float(wscale), float(coeff), ctypes.c_void_p(acc.data_ptr()),
                hip.ptr(dproto[i]), hip.ptr(dcoef[i]), ctypes.c_void_p(ws.data_ptr()), ws.numel(), hip.stream_ptr()),
                'ym_mask_loss_fwd_bwd')
        ctx.save_for_backward(dproto, dcoef)
        return (acc * coeff).float().reshape(())

    @staticmethod
    def backward(ctx, grad_out):
        dproto, dcoef = ctx.saved_tensors
        return dproto * grad_out, dcoef * grad_out, None, None


def lincomb_mask_loss(cfg, pos, anchor_gt, coef_p, proto_p, mask_gt, anchor_box):
    ph, pw = proto_p.shape[1:3]
    total_pos = int(pos.sum())
    per_image = []
    for i in range(coef_p.shape[0]):
        idx = torch.nonzero(pos[i]).flatten()
        if idx.shape[0] == 0:
            per_image.append(None)
            continue
        g = mask_gt[i].shape[0]
        ds = torch.empty(g, ph, pw, device=proto_p.device, dtype=torch.float32)     # bilinear(align_corners=False) then > 0.5
        hip.mask_resize_binarize(mask_gt[i].contiguous().float(), ph, pw, ds)
        gt_i, bx = anchor_gt[i][idx], anchor_box[i][idx]
        old = idx.shape[0]
        if old > cfg.masks_to_train:
            sel = torch.randperm(old)[:cfg.masks_to_train].to(idx.device)            # CPU generator, like the reference (:263)
            idx, gt_i, bx = idx[sel], gt_i[sel], bx[sel]
        per_image.append((idx.contiguous(), gt_i.to(torch.int32).contiguous(), bx.contiguous(), ds.reshape(g, ph * pw),
                          old / idx.shape[0]))
    coeff = cfg.mask_alpha / ph / pw / total_pos
    return _MaskLossFn.apply(proto_p, coef_p, per_image, coeff)


def semantic_seg_loss(cfg, seg_p, mask_gt, class_gt):
    b, nc, mh, mw = seg_p.shape
    total = 0
    for i in range(b):
        ds = F.interpolate(mask_gt[i].unsqueeze(0), (mh, mw), mode='bilinear', align_corners=False).squeeze(0).gt(0.5).float()
        tgt = torch.zeros_like(seg_p[i])
        for j in range(ds.shape[0]):
            tgt[class_gt[i][j]] = torch.max(tgt[class_gt[i][j]], ds[j])
        total = total + F.binary_cross_entropy_with_logits(seg_p[i], tgt, reduction='sum')
    return cfg.semantic_alpha * total / mh / mw / b


def compute_loss(cfg, anchors, class_p, box_p, coef_p, proto_p, seg_p, box_class, mask_gt):
    device = class_p.device
    b, n = box_p.shape[:2]
    offsets = torch.zeros(b, n, 4, device=device)
    conf_gt = torch.zeros(b, n, dtype=torch.int64, device=device)
    anchor_box = torch.zeros(b, n, 4, device=device)
    anchor_gt = torch.zeros(b, n, dtype=torch.int64, device=device)
    class_gt = []
    with torch.no_grad():
        for i in range(b):
            class_gt.append(box_class[i][:, -1].long())
            offsets[i], conf_gt[i], anchor_box[i], anchor_gt[i] = match(cfg, box_class[i][:, :-1], anchors, class_gt[i])
    pos = conf_gt > 0
    return (category_loss(cfg, class_p, conf_gt, pos), box_loss(cfg, box_p, offsets, pos),
            lincomb_mask_loss(cfg, pos, anchor_gt, coef_p, proto_p, mask_gt, anchor_box),
            semantic_seg_loss(cfg, seg_p, mask_gt, class_gt))
