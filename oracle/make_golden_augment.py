"""Pins the train_aug host logic (`yolact_minimal_amd.utils.augmentations.sample_train_aug`) and the oracle's stage chain
(`oracle/augment_ref.apply_plan`) against the REAL reference `utils.augmentations.train_aug`, and writes
tests/golden/augment.npz.  TEST INFRASTRUCTURE ONLY.  Run from the repo root: python oracle/make_golden_augment.py

cv2 is absent from this image.  The reference's train_aug touches it in exactly two primitives — cv2.cvtColor (BGR<->HSV) and
cv2.resize — neither of which influences the random decisions or the box arithmetic.  To let the reference's OWN control flow run
end to end, a `cv2` module object is registered whose `resize` / `cvtColor` call the oracle's restatements of those two
primitives (oracle/augment_ref.py).  What is pinned bit-for-bit against the reference is therefore: the order and arguments of
every `random` call, the float64 box / label bookkeeping, which masks survive, and the ORDER / OFFSETS of the pixel stages
(mirror, crop, pads, final crop) around those two primitives.  The primitives themselves stay "parity unpinned by cv2".
"""
import importlib
import os
import random
import sys
import tempfile
import types

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from oracle import augment_ref as A  # noqa: E402


def import_reference_aug():
    os.chdir(tempfile.mkdtemp(prefix='yolact_ref_cwd_'))
    sys.path.insert(0, '/root/reference')
    cv2 = types.ModuleType('cv2')
    cv2.COLOR_BGR2HSV, cv2.COLOR_HSV2BGR = 40, 54

    def resize(img, size):
        x = img if img.ndim == 3 else img[:, :, None]
        # (uint8 in -> OpenCV's fixed-point 8-bit path, uint8 out: the masks of an already-square sample)
        out = A.resize_bilinear_u8(x, size[0]) if x.dtype == np.uint8 else A.resize_bilinear(x.astype(np.float32), size[0])
        return out if img.ndim == 3 and out.shape[2] > 1 else out[:, :, 0] if out.shape[2] == 1 else out

    def cvt(img, code):
        return A.bgr_to_hsv(img) if code == cv2.COLOR_BGR2HSV else A.hsv_to_bgr(img)
    cv2.resize, cv2.cvtColor = resize, cvt
    sys.modules['cv2'] = cv2
    return importlib.import_module('utils.augmentations')


from yolact_minimal_amd.utils.synthetic import synth_sample  # noqa: E402,F401  (input generator)


def main():
    ref = import_reference_aug()
    from yolact_minimal_amd.utils.augmentations import sample_train_aug
    out, n_none, n_u8 = {}, 0, 0
    cases = [(s, 96 + 8 * (s % 5), 128 - 6 * (s % 4), 1 + s % 4, 160 if s % 3 else 544) for s in range(40)]
    cases[8:12] = [(40 + s, 104, 104, 1 + s % 3, 160) for s in range(4)]          # square images: the uint8-mask branch when no crop is drawn
    for k, (seed, h, w, n, size) in enumerate(cases):
        img, masks, boxes, labels = synth_sample(seed, h, w, n)
        random.seed(1000 + seed)
        r_img, r_masks, r_boxes, r_labels = ref.train_aug(img.copy(), masks.copy(), boxes.copy(), labels.copy(), size)
        random.seed(1000 + seed)
        plan = sample_train_aug(h, w, boxes.copy(), labels.copy(), size)
        if r_img is None:
            assert plan is None, seed
            n_none += 1
            continue
        assert plan is not None, seed
        assert np.array_equal(plan.boxes, r_boxes), (seed, plan.boxes, r_boxes)          # float64, bit for bit
        assert np.array_equal(np.asarray(plan.labels, dtype=np.float64), np.asarray(r_labels, dtype=np.float64)), seed
        o_img, o_masks = A.apply_plan(img, masks, plan)
        n_u8 += int(masks.dtype == np.uint8 and plan.crop[2] == plan.crop[3])
        assert o_masks.shape == r_masks.shape, (seed, o_masks.shape, r_masks.shape)
        np.testing.assert_allclose(o_img, r_img.astype(np.float32), rtol=0, atol=2e-4, err_msg=str(seed))
        np.testing.assert_allclose(o_masks, r_masks.astype(np.float32), rtol=0, atol=1e-6, err_msg=str(seed))
        if k < 12:                                       # keep a dozen as committed vectors
            out[f'c{k}_case'] = np.array([seed, h, w, n, size])
            out[f'c{k}_boxes'] = r_boxes
            out[f'c{k}_labels'] = np.asarray(r_labels, dtype=np.float64)
            out[f'c{k}_plan'] = np.array([plan.brightness if plan.brightness is not None else np.nan,
                                          plan.contrast if plan.contrast is not None else np.nan, plan.saturation, plan.hue,
                                          float(plan.mirror), *plan.crop, plan.square, *plan.pad, plan.resize,
                                          *(plan.final_pad or (-1, -1)), *(plan.final_crop or (-1, -1))])
            out[f'c{k}_img_digest'] = np.array([r_img.astype(np.float64).sum(), np.abs(r_img.astype(np.float64)).sum()])
            out[f'c{k}_mask_sum'] = r_masks.astype(np.float64).sum(axis=(1, 2))
    assert n_u8 >= 1, 'no sample exercised the uint8 (already square) mask branch'
    np.savez_compressed(os.path.join(REPO, 'tests', 'golden', 'augment.npz'), **out)
    print(f'{len(cases)} seeded samples: decisions, boxes, labels and stage chain equal to the reference ({n_none} rejected by both); '
          f'wrote tests/golden/augment.npz')


if __name__ == '__main__':
    main()
