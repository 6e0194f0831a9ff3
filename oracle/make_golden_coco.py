#!/usr/bin/env python3
"""Regression vectors for the annotation -> mask step (tests/golden/coco_polys.json).

pycocotools cannot run in the build image, so these vectors are NOT reference outputs: they freeze what oracle/coco_ref.py (the
restatement of cocoapi rleFrPoly / rleMerge / rleDecode) produces today, as run-length counts, so that a later change of the
oracle or of the HIP kernel cannot drift silently.  "Parity unpinned by the reference" stays true for this row."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import coco_ref as C  # noqa: E402
from oracle import rle_ref as R  # noqa: E402

cases = []
for seed, (h, w) in enumerate(((48, 64), (37, 53), (120, 90), (33, 4))):
    segs = C.synth_polygons(100 + seed, h, w, n=5)
    segs.append([[0.5, 0.5, w - 0.5, 0.5, w - 0.5, h - 0.5, 0.5, h - 0.5]])          # half-pixel rectangle
    segs.append([[-3.0, -2.0, w + 4.0, -2.0, w + 4.0, h + 3.0, -3.0, h + 3.0]])      # larger than the image
    cases.append({'h': h, 'w': w, 'segmentations': segs,
                  'counts': [R.rle_counts(C.segm_to_mask(s, h, w)) for s in segs]})
out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'coco_polys.json')
with open(out, 'w') as f:
    json.dump({'generator': 'oracle/make_golden_coco.py (oracle self-snapshot, not a pycocotools output)', 'cases': cases}, f)
print('wrote', out, os.path.getsize(out), 'bytes')
