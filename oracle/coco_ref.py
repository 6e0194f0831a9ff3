"""CPU restatement of the annotation -> mask step of the COCO reader.  TEST INFRASTRUCTURE ONLY (tests/ and bench.py's cpu leg).

The reference's dataset calls `self.coco.annToMask(aa)` per annotation (utils/coco.py:96).  That is pycocotools
(`coco.py annToRLE / annToMask` -> `_mask.frPyObjects` -> cocoapi `common/maskApi.c rleFrPoly`, `rleMerge`, `rleDecode`);
pycocotools is un-vendored, unpinned by the reference's README and ABSENT from this image, so the published algorithm is
restated here and **parity is unpinned by the reference**: it rests on hand-derived known answers (integer rectangles fill the
half-open box, tests/test_oracle_coco.py) and on structural properties (union over polygons, RLE round trip).

rleFrPoly in words: scale the vertices by 5 and round; walk every edge one step at a time along its longer axis (rounding the
other coordinate); every time the walk crosses to a new integer x it records — if that x sits exactly on a pixel-column centre
after undoing the scale — the boundary point (column, first row at or below the crossing); the points, ordered in the
column-major pixel order, are where the run-length code toggles between 0 and 1.  Two points on the same pixel cancel.
"""
import math

import numpy as np

from .rle_ref import rle_decode, rle_from_string

SCALE = 5.0


def _rounded(v):
    return int(SCALE * v + .5)          # C `(int)(scale*v+.5)`: truncation toward zero


def poly_boundary_points(xy, h, w):
    """The (column, row) toggle points of one polygon `xy = [x0, y0, x1, y1, ...]` (maskApi.c rleFrPoly, first two stages)."""
    k = len(xy) // 2
    px = [_rounded(xy[2 * j]) for j in range(k)]
    py = [_rounded(xy[2 * j + 1]) for j in range(k)]
    px.append(px[0])
    py.append(py[0])
    us, vs = [], []
    for j in range(k):
        xs, xe, ys, ye = px[j], px[j + 1], py[j], py[j + 1]
        dx, dy = abs(xe - xs), abs(ys - ye)
        flip = (dx >= dy and xs > xe) or (dx < dy and ys > ye)
        if flip:
            xs, xe, ys, ye = xe, xs, ye, ys
        if dx >= dy:
            s = (ye - ys) / dx if dx else float('nan')       # 0/0 in C: only ever multiplied by t = 0 ... and cast (see below)
            for d in range(dx + 1):
                t = dx - d if flip else d
                us.append(t + xs)
                vs.append(int(ys + s * t + .5) if dx else ys)  # dx == dy == 0: a repeated vertex; C casts NaN (undefined) — no
                #                                                crossing can come from it because u does not change
        else:
            s = (xe - xs) / dy
            for d in range(dy + 1):
                t = dy - d if flip else d
                vs.append(t + ys)
                us.append(int(xs + s * t + .5))
    pts = []
    for j in range(1, len(us)):
        if us[j] == us[j - 1]:
            continue
        xd = float(us[j] if us[j] < us[j - 1] else us[j] - 1)
        xd = (xd + .5) / SCALE - .5
        if math.floor(xd) != xd or xd < 0 or xd > w - 1:
            continue
        yd = float(vs[j] if vs[j] < vs[j - 1] else vs[j - 1])
        yd = (yd + .5) / SCALE - .5
        yd = 0.0 if yd < 0 else (float(h) if yd > h else yd)
        pts.append((int(xd), int(math.ceil(yd))))
    return pts


def poly_to_counts(xy, h, w):
    """maskApi.c rleFrPoly: the run lengths (column-major, starting with zeros) of one polygon."""
    a = sorted(x * h + y for x, y in poly_boundary_points(xy, h, w)) + [h * w]
    diffs, prev = [], 0
    for t in a:
        diffs.append(t - prev)
        prev = t
    out = [diffs[0]]
    j = 1
    while j < len(diffs):
        if diffs[j] > 0:
            out.append(diffs[j])
            j += 1
        else:                                   # an empty run: the two neighbours join
            j += 1
            if j < len(diffs):
                out[-1] += diffs[j]
                j += 1
    return out


def poly_to_mask(xy, h, w):
    return rle_decode(poly_to_counts(xy, h, w), h, w)


def segm_to_mask(segm, h, w):
    """pycocotools annToMask on one annotation's `segmentation`: a list of polygons (their union), an uncompressed RLE
    ({'counts': [ints], 'size': [h, w]}) or a compressed one ({'counts': str})."""
    if isinstance(segm, list):
        m = np.zeros((h, w), np.uint8)
        for poly in segm:
            m |= poly_to_mask(poly, h, w).astype(np.uint8)
        return m
    counts = segm['counts']
    if isinstance(counts, (str, bytes)):
        counts = rle_from_string(counts.decode('ascii') if isinstance(counts, bytes) else counts)
    return rle_decode(list(counts), h, w).astype(np.uint8)


from yolact_minimal_amd.utils.synthetic import synth_polygons  # noqa: E402,F401  (input generator)


def write_synth_dataset(root, n_images=6, seed=0, sizes=((48, 64), (60, 44), (37, 53))):
    """A tiny COCO-format dataset on disk for the reader tests: JPEG images + instances JSON (polygon annotations with the
    boxes of their masks, one RLE 'iscrowd' annotation per image, one too-small box).  Returns the annotation path."""
    import json
    import os
    from PIL import Image
    rng = np.random.default_rng(seed)
    os.makedirs(os.path.join(root, 'imgs'), exist_ok=True)
    images, anns, aid = [], [], 1
    coco_ids = [1, 2, 3, 17, 18, 44, 90]
    for i in range(n_images):
        h, w = sizes[i % len(sizes)]
        name = f'{i:06d}.jpg'
        Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8)).save(os.path.join(root, 'imgs', name), quality=95)
        images.append({'id': 100 + i, 'file_name': name, 'height': h, 'width': w})
        for polys in synth_polygons(seed * 1000 + i, h, w, n=int(rng.integers(1, 5))):
            m = segm_to_mask(polys, h, w)
            ys, xs = np.nonzero(m)
            if len(xs) == 0:
                continue
            x0, y0, x1, y1 = float(xs.min()), float(ys.min()), float(xs.max() + 1), float(ys.max() + 1)
            anns.append({'id': aid, 'image_id': 100 + i, 'category_id': int(rng.choice(coco_ids)), 'iscrowd': 0,
                         'bbox': [x0, y0, x1 - x0, y1 - y0], 'area': float(m.sum()), 'segmentation': polys})
            aid += 1
        crowd = np.zeros((h, w), np.uint8)
        crowd[2:h // 2, 3:w // 2] = 1
        from .rle_ref import rle_counts
        anns.append({'id': aid, 'image_id': 100 + i, 'category_id': 1, 'iscrowd': 1, 'bbox': [3, 2, w // 2 - 3, h // 2 - 2],
                     'area': float(crowd.sum()), 'segmentation': {'size': [h, w], 'counts': rle_counts(crowd)}})
        aid += 1
        anns.append({'id': aid, 'image_id': 100 + i, 'category_id': 2, 'iscrowd': 0, 'bbox': [1.0, 1.0, 2.0, 3.0], 'area': 6.0,
                     'segmentation': [[1, 1, 3, 1, 3, 4, 1, 4]]})        # narrower than 4 px: dropped in train mode
        aid += 1
    path = os.path.join(root, 'instances.json')
    with open(path, 'w') as f:
        json.dump({'images': images, 'annotations': anns, 'categories': [{'id': c, 'name': str(c)} for c in coco_ids]}, f)
    return path
