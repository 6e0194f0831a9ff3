"""CPU: the train_aug host logic (`sample_train_aug`: random-call order, float64 box bookkeeping) and the oracle's stage chain
against golden vectors produced by the REAL reference's train_aug (oracle/make_golden_augment.py)."""
import os
import random

import numpy as np

from oracle import augment_ref as A
from oracle.make_golden_augment import synth_sample
from yolact_minimal_amd.utils.augmentations import sample_train_aug

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'augment.npz'))


def _plan_vector(plan):
    return np.array([plan.brightness if plan.brightness is not None else np.nan,
                     plan.contrast if plan.contrast is not None else np.nan, plan.saturation, plan.hue, float(plan.mirror),
                     *plan.crop, plan.square, *plan.pad, plan.resize, *(plan.final_pad or (-1, -1)), *(plan.final_crop or (-1, -1))])


def test_sampler_and_chain_match_reference_golden():
    n_cases = len([k for k in GOLD.files if k.endswith('_case')])
    assert n_cases >= 10
    modes = set()
    for k in range(n_cases):
        seed, h, w, n, size = (int(v) for v in GOLD[f'c{k}_case'])
        img, masks, boxes, labels = synth_sample(seed, h, w, n)
        random.seed(1000 + seed)
        plan = sample_train_aug(h, w, boxes, labels, size)
        assert plan is not None
        np.testing.assert_array_equal(plan.boxes, GOLD[f'c{k}_boxes'])                    # float64, bit for bit
        np.testing.assert_array_equal(np.asarray(plan.labels, dtype=np.float64), GOLD[f'c{k}_labels'])
        np.testing.assert_array_equal(_plan_vector(plan), GOLD[f'c{k}_plan'])
        o_img, o_masks = A.apply_plan(img, masks, plan)
        np.testing.assert_allclose(np.array([o_img.astype(np.float64).sum(), np.abs(o_img.astype(np.float64)).sum()]),
                                   GOLD[f'c{k}_img_digest'], rtol=1e-5)
        np.testing.assert_allclose(o_masks.astype(np.float64).sum(axis=(1, 2)), GOLD[f'c{k}_mask_sum'], rtol=1e-6, atol=1e-4)
        modes.add((plan.mirror, plan.crop != (0, 0, w, h), plan.final_pad is not None, plan.final_crop is not None))
    assert len(modes) >= 4                                   # mirror / crop / final pad / final crop branches all occur


def test_hsv_round_trip_and_known_colours():
    rng = np.random.default_rng(0)
    img = rng.uniform(0, 255, (16, 16, 3)).astype(np.float32)
    np.testing.assert_allclose(A.hsv_to_bgr(A.bgr_to_hsv(img)), img, atol=2e-3)
    hsv = A.bgr_to_hsv(np.array([[[0, 0, 255], [0, 255, 0], [255, 0, 0], [128, 128, 128]]], dtype=np.float32))
    np.testing.assert_allclose(hsv[0, :, 0], [0, 120, 240, 0])               # red, green, blue, grey
    np.testing.assert_allclose(hsv[0, :, 1], [1, 1, 1, 0])
