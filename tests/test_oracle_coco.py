"""CPU: the annotation -> mask restatement (oracle/coco_ref.py, cocoapi rleFrPoly) against hand-derived known answers —
pycocotools is absent from the image and the reference holds no fixtures for it: parity unpinned by the reference — plus the host
half of the COCO reader (annotation index, filtering, rank sharding, RLE string parsing), which needs no GPU."""
import types

import numpy as np
import pytest
import torch

from oracle import coco_ref as C
from oracle import rle_ref as R


def test_rectangle_known_answers():
    # hand-walked through rleFrPoly (5x grid): the rectangle (1,1)-(3,2) in a 4 x 5 image gives the toggle points
    # (1,1) (1,2) (2,1) (2,2) -> positions 5 6 9 10 -> runs [5,1,3,1,10] -> the half-open box x in [1,3), y in [1,2)
    assert C.poly_to_counts([1, 1, 1, 2, 3, 2, 3, 1], 4, 5) == [5, 1, 3, 1, 10]
    assert sorted(C.poly_boundary_points([1, 1, 1, 2, 3, 2, 3, 1], 4, 5)) == [(1, 1), (1, 2), (2, 1), (2, 2)]
    # integer rectangles fill exactly [x0, x1) x [y0, y1) (the identity rleFrBbox relies on), any vertex order / start
    rng = np.random.default_rng(0)
    for _ in range(40):
        h, w = int(rng.integers(4, 40)), int(rng.integers(4, 40))
        x0, x1 = sorted(rng.choice(w + 1, 2, replace=False).tolist())
        y0, y1 = sorted(rng.choice(h + 1, 2, replace=False).tolist())
        box = np.zeros((h, w), np.uint8)
        box[y0:y1, x0:x1] = 1
        ring = [(x0, y0), (x0, y1), (x1, y1), (x1, y0)]
        if rng.random() < 0.5:
            ring = ring[::-1]
        r = int(rng.integers(4))
        ring = ring[r:] + ring[:r]
        np.testing.assert_array_equal(C.poly_to_mask([float(c) for p in ring for c in p], h, w), box)
    # the whole image, and a polygon that reaches outside it (clamped to the last row / dropped past the last column)
    np.testing.assert_array_equal(C.poly_to_mask([0, 0, 0, 6, 9, 6, 9, 0], 6, 9), np.ones((6, 9), np.uint8))
    m = C.poly_to_mask([2, 2, 2, 50, 50, 50, 50, 2], 6, 9)
    box = np.zeros((6, 9), np.uint8)
    box[2:, 2:] = 1
    np.testing.assert_array_equal(m, box)


def test_degenerate_polygons():
    assert C.poly_to_mask([3.3, 2.2], 8, 8).sum() == 0                                  # one vertex
    assert C.poly_to_mask([1, 1, 5, 5], 8, 8).sum() == 0                                # a segment walked there and back cancels
    a = C.poly_to_mask([1, 1, 1, 1, 1, 5, 6, 5, 6, 5, 6, 1], 8, 8)                      # repeated vertices change nothing
    np.testing.assert_array_equal(a, C.poly_to_mask([1, 1, 1, 5, 6, 5, 6, 1], 8, 8))


def test_union_and_rle_forms():
    h, w = 40, 56
    for polys in C.synth_polygons(3, h, w, n=8):
        u = np.zeros((h, w), np.uint8)
        for p in polys:
            c = C.poly_to_counts(p, h, w)
            assert sum(c) == h * w and all(v > 0 for v in c[1:])
            u |= R.rle_decode(c, h, w).astype(np.uint8)
        np.testing.assert_array_equal(C.segm_to_mask(polys, h, w), u)
        counts = R.rle_counts(u)
        np.testing.assert_array_equal(C.segm_to_mask({'size': [h, w], 'counts': counts}, h, w), u)
        np.testing.assert_array_equal(C.segm_to_mask({'size': [h, w], 'counts': R.rle_to_string(counts)}, h, w), u)


def test_triangle_is_close_to_the_geometric_fill():
    # sanity of the restatement as a rasteriser: pixel centres inside the triangle, up to the boundary pixels
    h, w = 60, 70
    tri = [10.5, 3.2, 60.1, 30.7, 5.3, 50.9]
    m = C.poly_to_mask(tri, h, w).astype(bool)
    ys, xs = np.mgrid[0:h, 0:w]
    px, py = xs + 0.5, ys + 0.5
    (ax, ay), (bx, by), (cx, cy) = (tri[0], tri[1]), (tri[2], tri[3]), (tri[4], tri[5])
    s1 = (bx - ax) * (py - ay) - (by - ay) * (px - ax)
    s2 = (cx - bx) * (py - by) - (cy - by) * (px - bx)
    s3 = (ax - cx) * (py - cy) - (ay - cy) * (px - cx)
    inside = ((s1 >= 0) & (s2 >= 0) & (s3 >= 0)) | ((s1 <= 0) & (s2 <= 0) & (s3 <= 0))
    assert (m ^ inside).sum() <= 0.04 * inside.sum()


def _cfg(root, ann, img_size=64, val_num=-1):
    from yolact_minimal_amd.config import COCO_LABEL_MAP
    return types.SimpleNamespace(train_imgs=root + '/imgs', val_imgs=root + '/imgs', train_ann=ann, val_ann=ann, img_size=img_size,
                                 continuous_id=COCO_LABEL_MAP, val_num=val_num, image=root + '/imgs')


def test_reader_host_half(tmp_path):
    from yolact_minimal_amd.utils import coco as K
    ann = C.write_synth_dataset(str(tmp_path), n_images=5, seed=1)
    cfg = _cfg(str(tmp_path), ann, val_num=3)
    idx = K.COCO(ann)
    assert len(idx.imgs) == 5 and set(idx.imgToAnns) == set(idx.imgs)
    some = idx.getAnnIds(imgIds=101)
    assert [a['image_id'] for a in idx.loadAnns(some)] == [101] * len(some)
    assert idx.getAnnIds(imgIds=101, iscrowd=1) == [a['id'] for a in idx.imgToAnns[101] if a['iscrowd']]
    assert idx.getAnnIds(imgIds=[101], catIds=[2]) == [a['id'] for a in idx.imgToAnns[101] if a['category_id'] == 2]
    assert idx.loadImgs(101)[0]['file_name'] == '000001.jpg'

    for mode in ('train', 'val'):
        ds = K.COCODetection(cfg, mode, device='cpu')           # read() is host-only
        assert len(ds) == (5 if mode == 'train' else 3)
        rec = ds.read(1)
        every = [a for a in idx.imgToAnns[rec['img_id']] if not a['iscrowd']]
        assert rec['img'].dtype == np.uint8 and rec['img'].shape == (60, 44, 3)
        # train drops the < 4 px box, val keeps it; crowd annotations never pass
        assert len(rec['anns']) == (len(every) - 1 if mode == 'train' else len(every))
        for a, b, l in zip(rec['anns'], rec['boxes'], rec['labels']):
            x, y, bw, bh = a['bbox']
            np.testing.assert_array_equal(b, [x, y, x + bw, y + bh])
            assert l == cfg.continuous_id[a['category_id']] - 1
    det = K.COCODetection(cfg, 'detect', device='cpu')
    assert len(det) == 5 and det.read(0)['name'] == '000000.jpg'

    # BGR like cv2.imread: channel 0 of ours is channel 2 of PIL's RGB
    from PIL import Image
    rgb = np.asarray(Image.open(str(tmp_path / 'imgs' / '000000.jpg')).convert('RGB'))
    np.testing.assert_array_equal(K.imread_bgr(str(tmp_path / 'imgs' / '000000.jpg')), rgb[:, :, ::-1])

    with pytest.raises(RuntimeError, match='HIP'):
        K.anns_to_masks([[[1, 1, 3, 1, 3, 4]]], 8, 8, device='cpu')      # no CPU rasteriser in the product


def test_rle_string_parser_matches_oracle():
    from yolact_minimal_amd.utils.coco import _rle_string_to_counts
    rng = np.random.default_rng(5)
    for _ in range(20):
        c = rng.integers(0, 5000, int(rng.integers(1, 30))).tolist()
        s = R.rle_to_string(c)
        assert _rle_string_to_counts(s) == c == R.rle_from_string(s)
        assert _rle_string_to_counts(s.encode('ascii')) == c


@pytest.mark.parametrize('n,world', [(10, 1), (10, 4), (7, 2), (13, 8)])
def test_batch_loader_shards_like_distributed_sampler(n, world):
    from torch.utils.data import DistributedSampler
    from yolact_minimal_amd.utils.coco import BatchLoader

    class DS:
        def __len__(self):
            return n

        def read(self, i):
            return i

        def finish(self, r):
            return r

    for shuffle in (False, True):
        seen = []
        for rank in range(world):
            ref = DistributedSampler(DS(), num_replicas=world, rank=rank, shuffle=shuffle, seed=3)
            ref.set_epoch(2)
            bl = BatchLoader(DS(), 2, lambda b: b, shuffle=shuffle, rank=rank, world_size=world, seed=3, drop_last=False)
            bl.set_epoch(2)
            assert bl.indices() == list(ref)
            got = [i for batch in bl for i in batch]
            assert got == list(ref) and len(bl) == -(-len(got) // 2)
            seen += got
        assert set(seen) == set(range(n))


def test_oracle_matches_its_frozen_vectors(golden_dir):
    import json
    import os
    gold = json.load(open(os.path.join(golden_dir, 'coco_polys.json')))
    for case in gold['cases']:
        for seg, counts in zip(case['segmentations'], case['counts']):
            assert R.rle_counts(C.segm_to_mask(seg, case['h'], case['w'])) == counts
