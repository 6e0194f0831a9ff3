"""The oracle's correctly rounded exp (oracle/expf_cr.c) against exact arithmetic and against the reference's torch.exp.

The reference's exp in the box decode (utils/output_utils.py:150) is MKL VML on this torch build — not reproducible op for
op (closed, host-ISA dependent); see oracle/make_golden_exp.py for the measurement.  These tests pin the anchor used instead."""
import os
from decimal import Decimal, getcontext

import numpy as np
import torch

from oracle import yolact_ref as R


def _round_to_float32(d):
    """Decimal -> nearest float32 (ties cannot occur for exp of a non-zero float)."""
    f = np.float32(float(d))                     # double rounding is checked below by comparing neighbours exactly
    cands = [np.nextafter(f, np.float32(-np.inf), dtype=np.float32), f, np.nextafter(f, np.float32(np.inf), dtype=np.float32)]
    return min(cands, key=lambda c: abs(Decimal(float(c)) - d))


def test_expf_cr_is_correctly_rounded_against_50_digit_arithmetic():
    getcontext().prec = 50
    g = torch.Generator().manual_seed(5)
    x = torch.cat([torch.randn(1500, generator=g) * 0.5, (torch.rand(500, generator=g) * 2 - 1) * 80.0,
                   torch.tensor([0.0, -0.0, 1.0, -1.0, 88.0, -87.0, 1e-10, -1e-10])])
    y = R.expf_cr(x).numpy()
    for xi, yi in zip(x.numpy(), y):
        want = _round_to_float32(Decimal(float(xi)).exp())
        assert yi == want, (xi, yi, want)


def test_expf_cr_matches_double_exp_rounded_on_a_million_inputs_and_specials():
    g = torch.Generator().manual_seed(6)
    x = torch.cat([(torch.rand(1_000_000, generator=g) * 2 - 1) * 100.0, torch.randn(1_000_000, generator=g)])
    np.testing.assert_array_equal(R.expf_cr(x).numpy(), torch.exp(x.double()).float().numpy())
    sp = torch.tensor([float('nan'), float('inf'), -float('inf'), 88.72, 88.73, 89.5, -103.0, -103.9, -104.5, -200.0])
    got, want = R.expf_cr(sp), torch.exp(sp.double()).float()
    assert torch.isnan(got[0]) and torch.equal(got[1:], want[1:])


def test_reference_torch_exp_is_within_one_ulp_of_the_anchor(golden_dir):
    """Frozen outputs of the reference's torch.exp (build container, MKL AVX-512 path): never more than 1 ulp from the
    correctly rounded value, equal in ~99 % — which is why boxes are pinned to the anchor, not to this host-dependent libm."""
    g = np.load(os.path.join(golden_dir, 'exp_torch_cpu.npz'))
    cr = R.expf_cr(torch.from_numpy(g['x'])).numpy()
    ulp = np.abs(cr.view(np.int32).astype(np.int64) - g['y'].view(np.int32).astype(np.int64))
    assert ulp.max() <= 1
    assert 0.97 < float((ulp == 0).mean()) < 1.0      # it really is a different function, but only just


def _post_cases():
    """The inputs of tests/golden/post_*.npz, regenerated from their seeds (oracle/make_golden.py gen_post)."""
    a544 = R.anchors_for(544, [24, 48, 96, 192, 384])
    a128 = R.anchors_for(128, [int(128 / 544 * s) for s in (24, 48, 96, 192, 384)])
    yield ('dense544',) + tuple(R.synth_head_outputs(18525, seed=1)) + (a544,)
    yield ('sparse544',) + tuple(R.synth_head_outputs(18525, seed=2, bg_bias=9.0)) + (a544,)
    yield ('small128',) + tuple(R.synth_head_outputs(1023, proto_hw=32, seed=3, bg_bias=5.0)) + (a128,)
    cls, box, coef, proto = R.synth_head_outputs(1023, proto_hw=32, seed=5, bg_bias=7.5)
    box[0, ::3, 0] = -40.0
    box[0, ::3, 2] = -8.0
    yield 'degenerate128', cls, box, coef, proto, a128


def test_the_box_deviation_is_the_references_libm_and_nothing_else(golden_dir):
    """Closes the question "are the kernel's boxes the reference's boxes?" (the GPU tests assert kernel == oracle with exp='cr'
    bit for bit; against the reference's frozen boxes ~1 % of the coordinates differ by 1 ulp):
      (i)  decoding with the FROZEN outputs of the reference's own torch.exp (MKL VML on the build host, stored per candidate in
           tests/golden/exp_decode_frozen.npz) instead of calling exp reproduces the reference's boxes BIT FOR BIT, together with
           ids / scores / coefs -- so everything in the decode and the NMS except that one libm call is restated exactly;
      (ii) with the correctly rounded exp the result differs from the reference's ONLY in box coordinates, and a decoded candidate
           box differs from its frozen counterpart ONLY in coordinates fed by a w / h whose frozen exp value is not the correctly
           rounded one (x1, x2 <- exp(b_w * 0.2); y1, y2 <- exp(b_h * 0.2)), never by more than the 1 ulp of that exp."""
    frozen = np.load(os.path.join(golden_dir, 'exp_decode_frozen.npz'))
    n_diff_coords = 0
    for tag, cls, box, coef, proto, anchors in _post_cases():
        gold = np.load(os.path.join(golden_dir, f'post_{tag}.npz'))
        y = torch.from_numpy(frozen[f'{tag}_y'])
        keep = cls[0].t()[1:].max(dim=0)[0] > 0.05
        assert int(keep.sum()) == int(frozen[f'{tag}_n']) == y.shape[0]
        # (i) the reference's result, without its libm
        r = R.nms(cls, box, coef, proto, anchors, exp=y)
        for got, key in zip(r[:4], ('ids', 'scores', 'boxes', 'coefs')):
            np.testing.assert_array_equal(got.numpy(), gold[key], err_msg=f'{tag}: {key} with the frozen exp values')
        # (ii) candidate level: where do the two decodes differ?
        x = box[0][keep][:, 2:] * 0.2
        cr = R.expf_cr(x)
        ulp = (cr.view(torch.int32).long() - y.view(torch.int32).long()).abs()
        assert int(ulp.max()) <= 1
        b_cr, b_fr = R.decode(box[0][keep], anchors[keep], 'cr'), R.decode(box[0][keep], anchors[keep], y)
        differs = b_cr != b_fr                                    # [K, 4] = (x1, y1, x2, y2)
        exp_differs = (ulp != 0)[:, [0, 1, 0, 1]]                  # which exp feeds which coordinate
        assert not bool((differs & ~exp_differs).any()), f'{tag}: a box coordinate differs although its exp value does not'
        n_diff_coords += int(differs.sum())
        # ... and the detections: same ids / scores / coefs, boxes within the 1 ulp of exp (<= 1.2e-7 for boxes in [0, 1])
        c = R.nms(cls, box, coef, proto, anchors, exp='cr')
        assert torch.equal(c[0], r[0]) and torch.equal(c[1], r[1]) and torch.equal(c[3], r[3])
        assert float((c[2] - r[2]).abs().max()) <= 1.2e-7
    assert n_diff_coords > 0          # (the test is not vacuous: the two exps do differ on these inputs)
