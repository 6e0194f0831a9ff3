"""GPU parity of the COCO reader's annotation -> mask step: `ym_poly_to_mask` / `ym_runs_to_mask` behind
`COCO.annToMask` / `COCODetection`, bit-exact against the oracle's restatement of pycocotools (oracle/coco_ref.py)."""
import ctypes
import types

import numpy as np
import pytest
import torch

from oracle import coco_ref as C
from oracle import rle_ref as R

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _oracle(segs, h, w):
    return np.stack([C.segm_to_mask(s, h, w) for s in segs])


@pytest.mark.parametrize('h,w', [(48, 64), (37, 53), (1, 9), (9, 1), (33, 4), (240, 320), (427, 640)])
def test_polygons_match_oracle_bit_exact(h, w):
    from yolact_minimal_amd.utils.coco import anns_to_masks
    segs = C.synth_polygons(h * 1000 + w, h, w, n=7)
    segs.append([[0, 0, 0, h, w, h, w, 0]])                          # the whole image
    segs.append([[1.0, 1.0]])                                        # degenerate: nothing
    segs.append([[2, 1, 2, 1, 2.4, 3.3, w + 30, h + 20, w - 1, 0.2, w - 1, 0.2]])   # repeated vertices, far outside
    segs.append([])                                                  # an annotation without polygons
    got = anns_to_masks(segs, h, w, DEV)
    assert got.dtype == torch.uint8 and tuple(got.shape) == (len(segs), h, w)
    np.testing.assert_array_equal(got.cpu().numpy(), _oracle(segs, h, w))


def test_large_masks_use_the_workspace_path():
    from yolact_minimal_amd import hip
    from yolact_minimal_amd.utils.coco import anns_to_masks
    h, w = 900, 1100
    assert hip.lib().ym_ann_to_mask_workspace_bytes(3, h, w) > 0 and hip.lib().ym_ann_to_mask_workspace_bytes(3, 640, 640) == 0
    segs = C.synth_polygons(77, h, w, n=3)
    np.testing.assert_array_equal(anns_to_masks(segs, h, w, DEV).cpu().numpy(), _oracle(segs, h, w))


def test_rle_annotations_and_mixed_lists():
    from yolact_minimal_amd.utils.coco import anns_to_masks
    h, w = 61, 47
    rng = np.random.default_rng(2)
    dense = [(rng.random((h, w)) < p).astype(np.uint8) for p in (0.0, 0.02, 0.5, 1.0)]
    segs = [{'size': [h, w], 'counts': R.rle_counts(m)} for m in dense]
    segs += [{'size': [h, w], 'counts': R.rle_to_string(R.rle_counts(m))} for m in dense]
    got = anns_to_masks(segs, h, w, DEV).cpu().numpy()
    np.testing.assert_array_equal(got, np.stack(dense + dense))
    polys = C.synth_polygons(9, h, w, n=3)
    mixed = [polys[0], segs[2], polys[1], segs[5], polys[2]]
    np.testing.assert_array_equal(anns_to_masks(mixed, h, w, DEV).cpu().numpy(), _oracle(mixed, h, w))
    with pytest.raises(ValueError):
        anns_to_masks([{'size': [h + 1, w], 'counts': [h * w]}], h, w, DEV)


def test_long_run_lists_cross_the_scan_chunks():
    from yolact_minimal_amd.utils.coco import anns_to_masks
    h, w = 128, 96                                                   # checkerboard columns: > 512 runs per mask
    m = ((np.arange(h)[:, None] // 2 + np.arange(w)[None, :]) % 2).astype(np.uint8)
    counts = R.rle_counts(m)
    assert len(counts) > 2048
    np.testing.assert_array_equal(anns_to_masks([{'size': [h, w], 'counts': counts}], h, w, DEV)[0].cpu().numpy(), m)


def _cfg(root, ann, img_size):
    from yolact_minimal_amd.config import COCO_LABEL_MAP
    return types.SimpleNamespace(train_imgs=root + '/imgs', val_imgs=root + '/imgs', train_ann=ann, val_ann=ann, img_size=img_size,
                                 continuous_id=COCO_LABEL_MAP, val_num=-1, image=root + '/imgs')


def test_dataset_modes_end_to_end(tmp_path):
    import random
    from yolact_minimal_amd.utils import coco as K
    from yolact_minimal_amd.utils.augmentations import val_aug
    ann = C.write_synth_dataset(str(tmp_path), n_images=6, seed=4)
    cfg = _cfg(str(tmp_path), ann, 64)
    tcfg = _cfg(str(tmp_path), ann, 256)        # multi_scale_resize draws 256..768: a 64 px train crop would reject most samples

    val = K.COCODetection(cfg, 'val', device=DEV)
    idx = K.COCO(ann, device=DEV)
    for i in range(len(val)):
        img, boxes, masks, h, w = val[i]
        rec = val.read(i)
        assert tuple(img.shape) == (3, 64, 64) and img.is_cuda and masks.dtype == torch.uint8
        np.testing.assert_array_equal(masks.cpu().numpy(), _oracle([a['segmentation'] for a in rec['anns']], h, w))
        torch.testing.assert_close(img, val_aug(torch.from_numpy(rec['img']).to(DEV), 64), rtol=0, atol=0)
        assert boxes.shape == (len(rec['anns']), 5) and boxes[:, :4].max() <= 1.0
        np.testing.assert_array_equal(idx.annToMask(rec['anns'][0]).cpu().numpy(), masks[0].cpu().numpy())
    imgs, boxes, masks, h, w = K.val_collate([val[0]])
    assert tuple(imgs.shape) == (1, 3, 64, 64) and masks.dtype == torch.float32 and boxes.dtype == torch.float32

    train = K.COCODetection(tcfg, 'train', device=DEV, rng=random.Random(11))
    loader = K.BatchLoader(train, 3, K.train_collate, shuffle=True, seed=1, threads=2)
    nb = 0
    for imgs, targets, masks in loader:
        nb += 1
        assert tuple(imgs.shape) == (3, 3, 256, 256) and len(targets) == len(masks) == 3
        for t, m in zip(targets, masks):
            assert t.shape[1] == 5 and m.shape[0] == t.shape[0] and tuple(m.shape[1:]) == (256, 256) and m.dtype == torch.float32
            assert t.is_cuda and float(t[:, :4].min()) >= 0 and float(t[:, :4].max()) <= 1
    assert nb == len(loader) == 2

    # the same seed replays the same augmented batches (random draws happen in sample order on the consumer thread)
    def run():
        ds = K.COCODetection(tcfg, 'train', device=DEV, rng=random.Random(5))
        return [b[0].clone() for b in K.BatchLoader(ds, 2, K.train_collate, shuffle=True, seed=2, threads=3)]
    for a, b in zip(run(), run()):
        assert torch.equal(a, b)

    det = K.COCODetection(cfg, 'detect', device=DEV)
    img, origin, name = K.detect_collate([det[2]])
    assert tuple(img.shape) == (1, 3, 64, 64) and origin.dtype == np.uint8 and name == '000002.jpg'


def test_empty_and_rejected_inputs():
    from yolact_minimal_amd import hip
    from yolact_minimal_amd.utils.coco import anns_to_masks
    assert tuple(anns_to_masks([], 10, 12, DEV).shape) == (0, 10, 12)
    with pytest.raises(RuntimeError, match='W <= 4096'):
        anns_to_masks([[[0, 0, 5, 0, 5, 5]]], 4, 5000, DEV)
    out = torch.empty(1, 8, 8, dtype=torch.uint8, device=DEV)
    rc = hip.lib().ym_poly_to_mask(None, None, None, 1, 8, 8, ctypes.c_void_p(out.data_ptr()), None, 0, hip.stream_ptr())
    assert rc != 0 and b'null pointer' in hip.lib().ym_last_error()


def test_frozen_vectors(golden_dir):
    import json
    import os
    from yolact_minimal_amd.utils.coco import anns_to_masks
    gold = json.load(open(os.path.join(golden_dir, 'coco_polys.json')))
    for case in gold['cases']:
        h, w = case['h'], case['w']
        got = anns_to_masks(case['segmentations'], h, w, DEV).cpu().numpy()
        for m, counts in zip(got, case['counts']):
            np.testing.assert_array_equal(m, R.rle_decode(counts, h, w))
