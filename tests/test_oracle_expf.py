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
