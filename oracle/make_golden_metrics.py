"""Pins oracle/metrics_ref.py against the REAL reference's mask_iou / prep_metrics / APDataObject / calc_map and writes
tests/golden/metrics.npz.  TEST INFRASTRUCTURE ONLY.  Run from the repo root:  python oracle/make_golden_metrics.py

`utils/common_utils.py` imports pycocotools and terminaltables at module top (absent in this image; only MakeJson and the table
formatting use them): empty stand-in modules are registered first, `AsciiTable` is a pass-through holder.
"""
import importlib
import os
import sys
import tempfile
import types

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from oracle import metrics_ref as M  # noqa: E402


def import_reference():
    os.chdir(tempfile.mkdtemp(prefix='yolact_ref_cwd_'))
    sys.path.insert(0, '/root/reference')
    for name in ('cv2', 'pycocotools', 'terminaltables'):
        sys.modules[name] = types.ModuleType(name)

    class AsciiTable:
        def __init__(self, rows):
            self.table = rows
    sys.modules['terminaltables'].AsciiTable = AsciiTable
    return importlib.import_module('utils.common_utils'), importlib.import_module('utils.box_utils')


def main():
    cu, bu = import_reference()
    thres = [x / 100 for x in range(50, 100, 5)]
    out = {}
    for case, (seed, n, g, h, w, nc) in enumerate([(0, 40, 7, 48, 64, 6), (1, 100, 15, 120, 160, 10), (2, 9, 3, 33, 47, 4)]):
        ids, scores, boxes, masks, gt, gt_masks, h, w = M.synth_eval_case(seed, n, g, h, w, nc)
        # mask_iou
        ref_iou = bu.mask_iou(masks.reshape(n, -1), gt_masks.reshape(g, -1))
        mine = M.mask_iou(masks.reshape(n, -1), gt_masks.reshape(g, -1))
        assert torch.equal(torch.nan_to_num(ref_iou, nan=-1.0), torch.nan_to_num(mine, nan=-1.0))
        # prep_metrics on the reference's own APDataObject grid
        ref_ap = {k: [[cu.APDataObject() for _ in range(nc)] for _ in thres] for k in ('box', 'mask')}
        cu.prep_metrics(ref_ap, list(ids), list(scores), boxes.clone(), masks.clone(), gt.clone(), gt_masks.clone(), h, w, thres)
        my_ap = M.new_ap_data(nc, len(thres))
        M.prep_metrics(my_ap, ids, scores, boxes, masks, gt, gt_masks, h, w, thres)
        flat = []
        for kind in ('box', 'mask'):
            for k in range(len(thres)):
                for c in range(nc):
                    a, b = ref_ap[kind][k][c], my_ap[kind][k][c]
                    assert a.num_gt_positives == b.num_gt_positives and list(a.data_points) == list(b.data_points), (kind, k, c)
                    assert a.get_ap() == b.get_ap()
                    flat.append([a.num_gt_positives, len(a.data_points), sum(1 for p in a.data_points if p[1]), a.get_ap()])
        _, row2, row3 = cu.calc_map(ref_ap, thres, nc, step=0)
        mine_map = M.calc_map(my_ap, thres, nc)
        assert [round(v, 2) for v in mine_map['box']] == row2[1:] and [round(v, 2) for v in mine_map['mask']] == row3[1:]
        out[f'c{case}_shape'] = np.array([n, g, h, w, nc])
        out[f'c{case}_seed'] = np.array(seed)
        out[f'c{case}_mask_iou'] = ref_iou.numpy()
        out[f'c{case}_box_iou'] = bu.box_iou(boxes.float(), gt[:, :4] * torch.tensor([w, h, w, h])).numpy()
        out[f'c{case}_ap_grid'] = np.array(flat, dtype=np.float64)
        out[f'c{case}_map_box'] = np.array(mine_map['box'])
        out[f'c{case}_map_mask'] = np.array(mine_map['mask'])
        print('case', case, 'box mAP', row2[1], 'mask mAP', row3[1], 'matched', int(out[f'c{case}_ap_grid'][:, 2].sum()))
    np.savez_compressed(os.path.join(REPO, 'tests', 'golden', 'metrics.npz'), **out)
    print('wrote tests/golden/metrics.npz')


if __name__ == '__main__':
    main()
