"""CPU: the oracle's evaluation restatement (oracle/metrics_ref.py) against the golden vectors produced by the real reference
(oracle/make_golden_metrics.py): mask IoU bit-exact, AP grid (gt positives, pushes, true positives, AP) and mAP rows equal."""
import os

import numpy as np
import torch

from oracle import metrics_ref as M

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'metrics.npz'))
THRES = [x / 100 for x in range(50, 100, 5)]


def grid_of(ap, nc):
    rows = []
    for kind in ('box', 'mask'):
        for k in range(len(THRES)):
            for c in range(nc):
                a = ap[kind][k][c]
                rows.append([a.num_gt_positives, len(a.data_points), sum(1 for p in a.data_points if p[1]), a.get_ap()])
    return np.array(rows, dtype=np.float64)


def test_oracle_metrics_match_reference_golden():
    for case in range(3):
        n, g, h, w, nc = (int(v) for v in GOLD[f'c{case}_shape'])
        ids, scores, boxes, masks, gt, gt_masks, h, w = M.synth_eval_case(int(GOLD[f'c{case}_seed']), n, g, h, w, nc)
        iou = M.mask_iou(masks.reshape(n, -1), gt_masks.reshape(g, -1)).numpy()
        np.testing.assert_array_equal(iou, GOLD[f'c{case}_mask_iou'])
        ap = M.new_ap_data(nc, len(THRES))
        M.prep_metrics(ap, ids, scores, boxes, masks, gt, gt_masks, h, w, THRES)
        np.testing.assert_array_equal(grid_of(ap, nc), GOLD[f'c{case}_ap_grid'])
        m = M.calc_map(ap, THRES, nc)
        np.testing.assert_array_equal(np.array(m['box']), GOLD[f'c{case}_map_box'])
        np.testing.assert_array_equal(np.array(m['mask']), GOLD[f'c{case}_map_mask'])


def test_known_answer_ap():
    """Hand-derived: 2 gt, detections (0.9 TP), (0.8 FP), (0.7 TP) -> precision envelope [1, 2/3, 2/3], recalls [.5, .5, 1]:
    51 recall bars at 1.0 and 50 at 2/3."""
    a = M.APData()
    a.num_gt_positives = 2
    a.data_points = [(0.8, False), (0.7, True), (0.9, True)]
    assert abs(a.get_ap() - (51 * 1.0 + 50 * 2 / 3) / 101) < 1e-12
    iou = M.mask_iou(torch.tensor([[1., 1, 0, 0], [0, 0, 0, 0]]), torch.tensor([[0., 1, 1, 0], [0, 0, 0, 0]]))
    assert iou[0, 0] == torch.tensor(1 / 3, dtype=torch.float32) and iou[1, 0] == 0 and torch.isnan(iou[1, 1])
