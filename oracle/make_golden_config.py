"""Pins the configuration surface (`config.py`: cfg classes x modes x argument variants) against the REAL reference and writes
tests/golden/config.json.  TEST INFRASTRUCTURE ONLY.  Run from the repo root: python oracle/make_golden_config.py"""
import argparse
import json
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from oracle.make_golden import import_reference  # noqa: E402

CFGS = ('res101_coco', 'res50_coco', 'swin_tiny_coco', 'res50_pascal', 'res101_custom', 'res50_custom')
VARIANTS = [dict(img_size=544, train_bs=8, bs_per_gpu=8, resume=None, weight='weights/x.pth', traditional_nms=False, val_num=-1,
                 coco_api=False, val_interval=4000),
            dict(img_size=320, train_bs=16, bs_per_gpu=4, resume='weights/latest_res101_coco_1000.pth', weight=None,
                 traditional_nms=True, val_num=200, coco_api=True, val_interval=2000)]


def jsonable(v):
    if isinstance(v, np.ndarray):
        return ['ndarray', v.dtype.name, v.tolist()]
    if isinstance(v, (tuple, list)):
        return [type(v).__name__, [jsonable(x) for x in v]]
    if isinstance(v, dict):
        return ['dict', [[jsonable(k), jsonable(x)] for k, x in v.items()]]
    if isinstance(v, (bool, int, float, str)) or v is None:
        return v
    return ['repr', repr(v)]


def snapshot(cfg):
    return {k: jsonable(v) for k, v in sorted(vars(cfg).items())}


def make_args(name, mode, variant):
    a = argparse.Namespace(cfg=name, **variant)
    a.mode, a.cuda, a.gpu_id = mode, False, None
    if mode == 'detect':                      # detect.py's extra arguments (detect.py:17-32)
        a.image, a.video, a.hide_mask, a.hide_bbox, a.hide_score = 'img_dir', None, False, False, False
        a.cutout, a.save_lincomb, a.no_crop, a.real_time, a.visual_thre = False, False, False, False, 0.3
    return a


def main():
    ref_config, _, _, _ = import_reference()
    out = {}
    for name in CFGS:
        for mode in ('train', 'val', 'detect'):
            for vi, variant in enumerate(VARIANTS):
                cfg = getattr(ref_config, name)(make_args(name, mode, variant))
                out[f'{name}|{mode}|{vi}'] = snapshot(cfg)
    out['__module__'] = {k: jsonable(getattr(ref_config, k)) for k in ('COCO_CLASSES', 'COCO_LABEL_MAP', 'PASCAL_CLASSES', 'norm_mean',
                                                                       'norm_std') if hasattr(ref_config, k)}
    path = os.path.join(REPO, 'tests', 'golden', 'config.json')
    json.dump(out, open(path, 'w'), indent=0, sort_keys=True)
    print('wrote', path, len(out), 'snapshots')


if __name__ == '__main__':
    main()
