"""GPU parity of the evaluation step after after_nms: `ym_mask_iou` / `ym_box_iou` / `ym_match_detections`
behind the reference's `mask_iou` / `prep_metrics` / `calc_map` surface, against the oracle and the reference goldens."""
import os

import numpy as np
import pytest
import torch

from oracle import metrics_ref as M

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'metrics.npz'))
THRES = [x / 100 for x in range(50, 100, 5)]


def _eq_nan(a, b):
    return np.array_equal(np.nan_to_num(a, nan=-1.0), np.nan_to_num(b, nan=-1.0))


@pytest.mark.parametrize('case', [0, 1, 2])
def test_prep_metrics_matches_reference_golden(case):
    from yolact_minimal_amd.utils import common_utils as C
    from yolact_minimal_amd.utils.box_utils import mask_iou, box_iou
    n, g, h, w, nc = (int(v) for v in GOLD[f'c{case}_shape'])
    ids, scores, boxes, masks, gt, gt_masks, h, w = M.synth_eval_case(int(GOLD[f'c{case}_seed']), n, g, h, w, nc)
    iou = mask_iou(masks.reshape(n, -1).to(DEV), gt_masks.reshape(g, -1).to(DEV))
    assert not iou.is_cuda and _eq_nan(iou.numpy(), GOLD[f'c{case}_mask_iou'])          # bit-exact, returned on the host
    gt_px = gt[:, :4] * torch.tensor([w, h, w, h])
    assert _eq_nan(box_iou(boxes.float().to(DEV), gt_px.to(DEV)).cpu().numpy(), GOLD[f'c{case}_box_iou'])
    ap = {k: [[C.APDataObject() for _ in range(nc)] for _ in THRES] for k in ('box', 'mask')}
    gt_dev = gt.clone().to(DEV)
    C.prep_metrics(ap, ids, scores, boxes.to(DEV), masks.to(DEV), gt_dev, gt_masks.to(DEV), h, w, THRES)
    torch.testing.assert_close(gt_dev[:, :4].cpu(), gt_px)                              # scaled in place like the reference
    ref = M.new_ap_data(nc, len(THRES))
    M.prep_metrics(ref, ids, scores, boxes, masks, gt, gt_masks, h, w, THRES)
    rows = []
    for kind in ('box', 'mask'):
        for k in range(len(THRES)):
            for c in range(nc):
                a, b = ap[kind][k][c], ref[kind][k][c]
                assert a.num_gt_positives == b.num_gt_positives and list(a.data_points) == list(b.data_points), (kind, k, c)
                rows.append([a.num_gt_positives, len(a.data_points), sum(1 for p in a.data_points if p[1]), a.get_ap()])
    np.testing.assert_array_equal(np.array(rows, dtype=np.float64), GOLD[f'c{case}_ap_grid'])
    _, row2, row3 = C.calc_map(ap, THRES, nc, step=0)
    assert row2[1:] == [round(v, 2) for v in GOLD[f'c{case}_map_box']] and row3[1:] == [round(v, 2) for v in GOLD[f'c{case}_map_mask']]


@pytest.mark.parametrize('n,g,h,w', [(100, 20, 480, 640), (130, 150, 61, 67), (1, 1, 5, 3), (3, 2, 544, 544), (7, 5, 100, 100), (9, 4, 768, 1024)])
def test_mask_iou_full_size_and_edges(n, g, h, w):
    """BASELINE-size masks (100 x 480 x 640), more than one 128-row group on both sides, odd P (unaligned rows), empty masks
    (0/0 -> NaN): bit-identical to the fp32 matmul of the oracle."""
    from yolact_minimal_amd.utils.box_utils import mask_iou
    gen = torch.Generator().manual_seed(n * 7 + g)
    a = (torch.rand(n, h * w, generator=gen) > 0.6).float()
    b = (torch.rand(g, h * w, generator=gen) > 0.3).float()
    a[0] = 0
    b[-1] = 0
    got = mask_iou(a.to(DEV), b.to(DEV)).numpy()
    want = M.mask_iou(a.double(), b.double())
    inter = torch.matmul(a.double(), b.double().t())
    uni = (a.sum(1, keepdim=True).double() + b.sum(1).double()[None]) - inter
    want32 = (inter.float() / uni.float()).numpy()
    assert _eq_nan(got, want32)
    assert np.isnan(got[0, -1]) and want is not None


def test_match_detections_ties_and_thresholds():
    """Strictly-greater rule: a gt at exactly the threshold is not matched; among equal IoUs the first gt wins; a used gt is
    not matched twice; classes do not mix."""
    from yolact_minimal_amd.utils.common_utils import match_detections
    iou = torch.tensor([[0.5, 0.5, 0.9], [0.5000001, 0.5000001, 0.2], [0.7, 0.1, 0.9]], dtype=torch.float32, device=DEV)
    ids, gtc = [0, 0, 1], [0, 0, 1]
    m = match_detections(iou, iou, ids, gtc, [0.5, 0.75], 2)
    assert m.shape == (2, 2, 3)
    assert m[0, 0].tolist() == [0, 1, 1] and m[0, 1].tolist() == [0, 0, 1]


def _blob_masks(n, h, w, seed):
    g = torch.Generator().manual_seed(seed)
    m = torch.zeros(n, h, w)
    for i in range(n):
        for _ in range(1 + i % 3 if min(h, w) > 2 else 0):
            x1, y1 = int(torch.randint(0, w - 2, (1,), generator=g)), int(torch.randint(0, h - 2, (1,), generator=g))
            x2, y2 = int(torch.randint(x1 + 1, w + 1, (1,), generator=g)), int(torch.randint(y1 + 1, h + 1, (1,), generator=g))
            m[i, y1:y2, x1:x2] = 1.0
    return m


@pytest.mark.parametrize('n,h,w', [(100, 480, 640), (7, 33, 47), (3, 544, 544), (2, 1, 1)])
def test_rle_encode_matches_oracle(n, h, w):
    """`ym_rle_encode` vs the oracle's restatement of cocoapi rleEncode + rleToString (strings identical), at the BASELINE
    post-processing size, plus the encode -> decode round trip through the oracle's decoder."""
    from oracle import rle_ref as R
    from yolact_minimal_amd.utils.common_utils import rle_encode
    m = _blob_masks(n, h, w, n + h)
    m[0] = 0                                              # empty mask: one run
    m[-1] = 1                                             # full mask: leading zero-length run
    if n > 2:
        m[1, 0, 0] = 1                                    # foreground at the first pixel
    got = rle_encode(m.to(DEV))
    for i in range(n):
        want = R.encode(m[i].numpy())
        assert got[i] == want, i
    k = min(n - 1, 5)
    np.testing.assert_array_equal(R.rle_decode(R.rle_from_string(got[k]['counts']), h, w), m[k].numpy().astype(np.uint8))


def test_rle_encode_grows_buffers_for_busy_masks():
    from oracle import rle_ref as R
    from yolact_minimal_amd.utils.common_utils import rle_encode, MakeJson
    g = torch.Generator().manual_seed(3)
    m = (torch.rand(2, 120, 160, generator=g) > 0.5).float()          # ~9600 runs each > the default 4096
    got = rle_encode(m.to(DEV), cap_runs=64)
    assert [R.encode(m[i].numpy()) for i in range(2)] == got
    mj = MakeJson()
    mj.add_bbox(7, 0, [10.04, 20.06, 30.0, 50.0], 0.5)
    mj.add_mask(7, 0, m[0].to(DEV), 0.5)
    assert mj.bbox_data[0]['bbox'] == [10.0, 20.1, 20.0, 29.9] and mj.bbox_data[0]['category_id'] == 1
    assert mj.mask_data[0]['segmentation'] == got[0]
