"""CPU: the fixtures behind tests/test_gpu_overfit.py.  The synthetic shapes dataset is a pure function of its seed (the reference's
runs in tests/golden/overfit_reference_*.json were made from it), and runs of one recipe that differ only in length share their
first step."""
import json
import os
import sys

import numpy as np


def test_shapes_dataset_is_seeded_and_well_formed():
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tools'))
    from overfit_demo import make_dataset
    a, b = make_dataset(6, 64, seed=3), make_dataset(6, 64, seed=3)
    other = make_dataset(6, 64, seed=4)
    assert any(not np.array_equal(x[0].numpy(), y[0].numpy()) for x, y in zip(a, other))
    for (img, gt, masks), (img2, gt2, masks2) in zip(a, b):
        assert np.array_equal(img.numpy(), img2.numpy()) and np.array_equal(gt.numpy(), gt2.numpy()) and np.array_equal(masks.numpy(), masks2.numpy())
        assert img.shape == (3, 64, 64) and gt.shape[1] == 5 and masks.shape == (gt.shape[0], 64, 64) and 2 <= gt.shape[0] <= 3
        assert float(masks.sum(0).max()) == 1.0                                   # the shapes do not overlap
        for (x1, y1, x2, y2, c), m in zip(gt.tolist(), masks.numpy()):
            ys, xs = np.nonzero(m)
            assert 0 <= c < 4 and abs(xs.min() / 64 - x1) < 1e-6 and abs((ys.max() + 1) / 64 - y2) < 1e-6      # tight boxes


def test_reference_runs_of_one_recipe_share_their_first_step(golden_dir):
    g = {k: json.load(open(os.path.join(golden_dir, f'overfit_reference_{k}.json'))) for k in ('128', '128_1400', '128_ddp2', '128_seed1')}
    assert g['128']['losses'][0] == g['128_1400']['losses'][0]                     # same seed, same pictures, same first batch
    assert g['128']['losses'][0] != g['128_seed1']['losses'][0]
    assert g['128_ddp2']['world'] == 2 and g['128_ddp2']['losses'][0] != g['128']['losses'][0]      # rank 0's shard of 4 pictures
    for k, v in g.items():
        assert v['images_with_detections'] == v['images'] and all(np.isfinite(x).all() for _, x in v['losses']), k
