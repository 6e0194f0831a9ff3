"""Evaluation step right after `after_nms`: `prep_metrics` with its IoU caches and greedy matching on
the device.  Reference: `/root/reference/utils/common_utils.py:107-262` (`APDataObject`, `prep_metrics`, `calc_map`), called
from `eval.py:69,106`.

The reference computes `mask_iou` / `box_iou` on the device, copies both [n, g] matrices to the host and walks a triple python
loop (class x threshold x iou type) with a `.item()` per candidate pair.  Here: `ym_mask_iou` (bit-row popcounts, each mask read
once), `ym_box_iou`, then `ym_match_detections` resolves all 2 x T matchings in one launch; the host reads back ONE uint8
tensor [2, T, n] and only does the AP bookkeeping (lists of (score, is_true)), which stays Python like the reference.
"""
import ctypes

import numpy as np
import torch

from .. import hip
from .box_utils import box_iou, mask_iou


class APDataObject:
    """Accumulator of one (IoU type, IoU threshold, class) cell of the mAP table: the (score, hit?) pairs of every prediction and
    the number of ground-truth instances.  Same public surface as the reference's bookkeeping class (`utils/common_utils.py:107-171`:
    `push`, `add_gt_positives`, `is_empty`, `get_ap`, `data_points`, `num_gt_positives`); `get_ap` is evaluated on arrays and returns
    the reference's value bit for bit (pinned by tests/golden/metrics.npz: the divisions are the same IEEE operations and the
    101 samples are added in the same order)."""
    RECALL_GRID = np.arange(101) / 100                      # COCO's 101 recall thresholds 0.00 ... 1.00

    def __init__(self):
        self.data_points = []                               # [(score, is_true_positive)]
        self.num_gt_positives = 0

    def push(self, score, is_true):
        self.data_points.append((score, is_true))

    def add_gt_positives(self, num_positives):
        self.num_gt_positives += num_positives

    def is_empty(self):
        return not self.data_points and self.num_gt_positives == 0

    def get_ap(self):
        """Area under the monotone precision envelope sampled at the 101 recall thresholds (a threshold beyond the reached recall
        contributes 0)."""
        if self.num_gt_positives == 0:
            return 0
        self.data_points.sort(key=lambda p: -p[0])          # (kept in place and stable, like the reference: callers look at it)
        n = len(self.data_points)
        if n == 0:
            return 0.0
        hits = np.fromiter((bool(p[1]) for p in self.data_points), dtype=np.int64, count=n)
        tp = np.cumsum(hits)
        precision = tp / np.arange(1, n + 1)                # tp / (tp + fp) after each prediction
        recall = tp / self.num_gt_positives
        envelope = np.maximum.accumulate(precision[::-1])[::-1]
        first = np.searchsorted(recall, self.RECALL_GRID, side='left')      # first prediction that reaches each threshold
        samples = np.where(first < n, envelope[np.minimum(first, n - 1)], 0.0)
        return sum(samples.tolist()) / 101                  # left-to-right sum of python floats: the reference's rounding


_thr_cache = {}


def match_detections(iou_box, iou_mask, ids_p, gt_classes, iou_thres, num_classes):
    """[2, T, n] uint8 on the host: does prediction i find an unused same-class gt above threshold t (box / mask IoU)?"""
    dev = iou_box.device
    n, g, t = iou_box.shape[0], iou_box.shape[1], len(iou_thres)
    both = torch.tensor(list(ids_p) + list(gt_classes), dtype=torch.int32).to(dev)        # one H2D copy for the two class lists
    pred, gtc = both[:n], both[n:]
    key = (dev, tuple(iou_thres))
    thr = _thr_cache.get(key)
    if thr is None:
        if len(_thr_cache) > 16:
            _thr_cache.clear()
        thr = _thr_cache[key] = torch.tensor(iou_thres, dtype=torch.float64).to(dev)
    matched = torch.empty(2, t, n, dtype=torch.uint8, device=dev)       # (every prediction belongs to one class < num_classes: all written)
    hip.check(hip.lib().ym_match_detections(hip.ptr(iou_box), hip.ptr(iou_mask), hip.ptr(pred, torch.int32),
                                            hip.ptr(gtc, torch.int32), n, g, hip.ptr(thr, torch.float64), t, num_classes,
                                            ctypes.c_void_p(matched.data_ptr()), hip.stream_ptr()), 'ym_match_detections')
    return matched.cpu().numpy()


def prep_metrics(ap_data, ids_p, classes_p, boxes_p, masks_p, gt, gt_masks, height, width, iou_thres):
    """Same arguments, mutations and `ap_data` updates as the reference (common_utils.py:174-216): `gt` boxes are scaled to
    pixels IN PLACE; classes are visited as `set(ids_p + gt_classes)`; per class / threshold / iou type the gt positives are
    added and one (score, is_true) is pushed per prediction of that class, in prediction order."""
    gt_boxes = gt[:, :4]
    gt_boxes[:, 0::2] *= width                 # (columns 0, 2 / 1, 3 as strided views: no index tensors, one launch each)
    gt_boxes[:, 1::2] *= height
    gt_classes = gt[:, 4].int().tolist()
    gt_masks = gt_masks.reshape(-1, height * width)
    masks_p = masks_p.reshape(-1, height * width)
    ids_p = [int(i) for i in ids_p]

    iou_mask = mask_iou(masks_p, gt_masks, to_cpu=False)
    iou_box = box_iou(boxes_p.float(), gt_boxes.float())
    num_classes = max(ids_p + gt_classes) + 1
    matched = match_detections(iou_box, iou_mask, ids_p, gt_classes, iou_thres, num_classes)

    # host bookkeeping (a third of the loop's metric stage at 100 detections: 2 x 10 x classes cells per image): the flags become
    # nested python lists ONCE, predictions are grouped by class once; per cell only the reference's two updates remain
    flags = matched.astype(bool).tolist()                    # [iou type][threshold][prediction]
    by_class = {}
    for i, pred_class in enumerate(ids_p):
        by_class.setdefault(pred_class, []).append(i)        # prediction order is kept
    for _class in set(ids_p + gt_classes):
        num_gt_per_class = gt_classes.count(_class)
        mine = by_class.get(_class, ())
        scores = [classes_p[i] for i in mine]
        for iou_idx in range(len(iou_thres)):
            for type_idx, iou_type in enumerate(('box', 'mask')):
                ap_obj = ap_data[iou_type][iou_idx][_class]
                ap_obj.add_gt_positives(num_gt_per_class)
                if mine:
                    row = flags[type_idx][iou_idx]
                    ap_obj.data_points.extend(zip(scores, [row[i] for i in mine]))


def calc_map(ap_data, iou_thres, num_classes, step):
    """mAP table of the reference's `calc_map` (common_utils.py:219-255) from the AP grid [iou type][threshold][class]: per
    (type, threshold) the mean AP (x 100) over the classes that hold data, then the mean over the thresholds as 'all'.  Python float
    sums in class / threshold order, so the rounded rows equal the reference's (tests/golden/metrics_*.npz).  Returns
    (table text, box row, mask row); the table is plain ' | '-joined rows (terminaltables is a formatting dependency)."""
    def mean_ap(kind, t):
        cells = ap_data[kind][t]
        vals = [cells[c].get_ap() for c in range(num_classes) if not cells[c].is_empty()]
        return sum(vals) / len(vals) * 100 if vals else 0

    rows = [[f'{step // 1000}k' if step else '', 'all'] + [int(t * 100) for t in iou_thres]]
    for kind in ('box', 'mask'):
        per_thr = [mean_ap(kind, t) for t in range(len(iou_thres))]
        rows.append([kind] + [round(v, 2) for v in [sum(per_thr) / len(per_thr)] + per_thr])
    return '\n'.join(' | '.join(str(c) for c in row) for row in rows), rows[1], rows[2]


def rle_encode(masks, cap_runs=4096):
    """COCO RLE of binary masks [n, h, w] on the device (`ym_rle_encode`) -> list of {'size': [h, w], 'counts': str}, i.e.
    `pycocotools.mask.encode(np.asfortranarray(m.astype(np.uint8)))` with `counts` decoded to ascii (common_utils.py:90-91).
    Only the compressed strings (a few hundred bytes per mask) cross PCIe."""
    if not masks.is_cuda:
        raise RuntimeError('yolact_minimal_amd has no CPU path: rle_encode needs CUDA/HIP tensors')
    m = masks.to(torch.float32).contiguous()
    n, h, w = m.shape
    dev = m.device
    while True:
        cap_str = cap_runs * 4
        counts = torch.empty(n, cap_runs, dtype=torch.int32, device=dev)
        ws = torch.empty(n, cap_runs, dtype=torch.int32, device=dev)
        meta = torch.empty(2, n, dtype=torch.int32, device=dev)
        out = torch.empty(n, cap_str, dtype=torch.uint8, device=dev)
        hip.check(hip.lib().ym_rle_encode(hip.ptr(m), n, h, w, ctypes.c_void_p(counts.data_ptr()), cap_runs,
                                          ctypes.c_void_p(meta[0].data_ptr()), ctypes.c_void_p(out.data_ptr()), cap_str,
                                          ctypes.c_void_p(meta[1].data_ptr()), ctypes.c_void_p(ws.data_ptr()), ws.numel() * 4,
                                          hip.stream_ptr()), 'ym_rle_encode')
        nruns, slen = meta.cpu().tolist()
        if min(slen) >= 0:
            break
        cap_runs = max(cap_runs * 2, max(nruns) + 1)            # a mask was busier than the buffers: grow and redo
    longest = max(slen)
    host = out[:, :max(longest, 1)].cpu().numpy()
    return [{'size': [h, w], 'counts': host[i, :slen[i]].tobytes().decode('ascii')} for i in range(n)]


class MakeJson:
    """Detection dumps for the COCO API (reference common_utils.py:66-104, eval.py:59-67).  `add_mask` accepts the dense
    [h, w] mask like the reference (host or device) or an RLE dict from `rle_encode` (batch the image's masks there)."""

    def __init__(self, coco_label_map=None):
        from ..config import COCO_LABEL_MAP
        self.bbox_data, self.mask_data = [], []
        self.coco_cats = {}
        for coco_id, real_id in (COCO_LABEL_MAP if coco_label_map is None else coco_label_map).items():
            self.coco_cats[real_id - 1] = coco_id

    def _cat(self, category_id):
        return self.coco_cats[int(category_id)]

    def add_bbox(self, image_id, category_id, bbox, score):
        bbox = [bbox[0], bbox[1], bbox[2] - bbox[0], bbox[3] - bbox[1]]
        bbox = [round(float(x) * 10) / 10 for x in bbox]        # nearest 10th, as COCO suggests
        self.bbox_data.append({'image_id': int(image_id), 'category_id': self._cat(category_id), 'bbox': bbox,
                               'score': float(score)})

    def add_mask(self, image_id, category_id, segmentation, score):
        if not isinstance(segmentation, dict):
            seg = torch.as_tensor(segmentation)
            if not seg.is_cuda:
                seg = seg.cuda()
            segmentation = rle_encode(seg[None])[0]
        self.mask_data.append({'image_id': int(image_id), 'category_id': self._cat(category_id),
                               'segmentation': segmentation, 'score': float(score)})

    def dump(self, bbox_path='results/bbox_detections.json', mask_path='results/mask_detections.json'):
        import json
        for data, path in ((self.bbox_data, bbox_path), (self.mask_data, mask_path)):
            with open(path, 'w') as f:
                json.dump(data, f)


# ---- harness helpers the reference scripts import from this module (host only) -----------------------------------------
class ProgressBar:
    """`ProgressBar(length, max_val).get_bar(i)` -> a `length`-character bar (reference common_utils.py:15-38; eval.py:32,81)."""

    def __init__(self, length, max_val):
        self.length, self.max_val = int(length), max_val
        self.string = self.get_bar(0)

    def get_bar(self, new_val):
        filled = int(self.length * (min(new_val, self.max_val) / self.max_val)) if self.max_val else self.length
        self.string = '█' * filled + '░' * (self.length - filled)
        return self.string


def _replace_checkpoint(prefix, cfg_name, new_name, net):
    import glob
    import os
    old = [p for p in glob.glob(f'weights/{prefix}*') if cfg_name in p]
    assert len(old) <= 1, f'Error, multiple {prefix} weight found.'
    for p in old:
        os.remove(p)
    print(f"\nSaving the {prefix} model as '{new_name}'.\n")
    # (clones: under a Trainer / the module's own training state parameters and BatchNorm buffers are views of flat buffers, and
    #  torch.save would write each flat storage whole and share it between the entries; the reference's files hold one storage per key)
    torch.save({k: v.detach().clone() for k, v in net.state_dict().items()}, f'weights/{new_name}')


def save_best(net, mask_map, cfg_name, step):
    """Keep `weights/best_<mask mAP>_<cfg>_<step>.pth` if `mask_map` is at least the stored best (reference common_utils.py:41-53,
    train.py:174; the file-name convention is what eval.py / train.py --resume parse)."""
    import glob
    old = [p for p in glob.glob('weights/best*') if cfg_name in p]
    assert len(old) <= 1, 'Error, multiple best weight found.'
    best = float(old[0].split('/')[-1].split('_')[1]) if old else 0.
    if mask_map >= best:
        _replace_checkpoint('best', cfg_name, f'best_{mask_map}_{cfg_name}_{step}.pth', net)


def save_latest(net, cfg_name, step):
    """`weights/latest_<cfg>_<step>.pth`, replacing the previous one (reference common_utils.py:56-63, train.py:184,196)."""
    _replace_checkpoint('latest', cfg_name, f'latest_{cfg_name}_{step}.pth', net)
