"""CPU restatement of the evaluation step after `after_nms`.  TEST INFRASTRUCTURE ONLY: imported by
tests/ (and oracle/make_golden_metrics.py, which pins it bit-for-bit against the real reference), never by the product.

Follows /root/reference: `mask_iou` utils/box_utils.py:189-200, `box_iou` :8-37, `APDataObject` utils/common_utils.py:107-169,
`prep_metrics` :174-216, `calc_map` :219-262 (without the AsciiTable formatting).
"""
import numpy as np
import torch


def mask_iou(mask1, mask2):
    inter = torch.matmul(mask1, mask2.t())
    area1 = torch.sum(mask1, dim=1).reshape(1, -1)
    area2 = torch.sum(mask2, dim=1).reshape(1, -1)
    union = (area1.t() + area2) - inter
    return inter / union


def box_iou(a, b):
    hi = torch.min(a[:, None, 2:], b[None, :, 2:])
    lo = torch.max(a[:, None, :2], b[None, :, :2])
    wh = torch.clamp(hi - lo, min=0)
    inter = wh[..., 0] * wh[..., 1]
    area_a = ((a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1]))[:, None]
    area_b = ((b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1]))[None, :]
    return inter / (area_a + area_b - inter)


class APData:
    def __init__(self):
        self.data_points, self.num_gt_positives = [], 0

    def is_empty(self):
        return not self.data_points and self.num_gt_positives == 0

    def get_ap(self):
        if self.num_gt_positives == 0:
            return 0
        pts = sorted(self.data_points, key=lambda x: -x[0])        # list.sort is stable, like the reference's in-place sort
        tp = np.cumsum([1 if p[1] else 0 for p in pts])
        fp = np.cumsum([0 if p[1] else 1 for p in pts])
        prec = [float(t) / float(t + f) for t, f in zip(tp, fp)]
        rec = [float(t) / self.num_gt_positives for t in tp]
        for i in range(len(prec) - 1, 0, -1):
            if prec[i] > prec[i - 1]:
                prec[i - 1] = prec[i]
        idx = np.searchsorted(np.array(rec), np.array([x / 100 for x in range(101)]), side='left')
        ys = [prec[k] if k < len(prec) else 0 for k in idx]
        return sum(ys) / len(ys)


def new_ap_data(num_classes, num_thres):
    return {t: [[APData() for _ in range(num_classes)] for _ in range(num_thres)] for t in ('box', 'mask')}


def prep_metrics(ap_data, ids_p, classes_p, boxes_p, masks_p, gt, gt_masks, height, width, iou_thres):
    gt = gt.clone()
    gt[:, [0, 2]] *= width
    gt[:, [1, 3]] *= height
    gt_classes = gt[:, 4].int().tolist()
    caches = {'mask': mask_iou(masks_p.reshape(-1, height * width), gt_masks.reshape(-1, height * width)),
              'box': box_iou(boxes_p.float(), gt[:, :4].float())}
    for c in sorted(set(list(ids_p) + gt_classes)):
        for k, thr in enumerate(iou_thres):
            for kind in ('box', 'mask'):
                obj = ap_data[kind][k][c]
                obj.num_gt_positives += gt_classes.count(c)
                used = [False] * len(gt_classes)
                for i, pc in enumerate(ids_p):
                    if pc != c:
                        continue
                    best, bj = thr, -1
                    for j, gc in enumerate(gt_classes):
                        if used[j] or gc != c:
                            continue
                        v = caches[kind][i, j].item()
                        if v > best:
                            best, bj = v, j
                    if bj >= 0:
                        used[bj] = True
                    obj.data_points.append((classes_p[i], bj >= 0))


def calc_map(ap_data, iou_thres, num_classes):
    """-> {'box': [all, mAP@50, ...], 'mask': [...]} (the numbers of the reference's table rows, unrounded)."""
    out = {}
    for kind in ('box', 'mask'):
        per_thr = []
        for k in range(len(iou_thres)):
            aps = [ap_data[kind][k][c].get_ap() for c in range(num_classes) if not ap_data[kind][k][c].is_empty()]
            per_thr.append(sum(aps) / len(aps) * 100 if aps else 0)
        out[kind] = [sum(per_thr) / len(per_thr)] + per_thr
    return out


from yolact_minimal_amd.utils.synthetic import synth_eval_case  # noqa: E402,F401  (input generator)
