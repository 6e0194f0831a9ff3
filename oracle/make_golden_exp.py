"""Probe + golden for the one transcendental of the box decode (reference utils/output_utils.py:150, `torch.exp`).

Run in the build container:  python oracle/make_golden_exp.py
  1. measures what torch.exp(float32) on the CPU is NOT: an op-for-op C replica of SLEEF `Sleef_expf*_u10` (the algorithm
     ATen's Vectorized<float>::exp would call) and the correctly rounded value are both compared with it;
  2. freezes 8192 (x, torch.exp(x)) pairs of THIS host's torch build (MKL VML, AVX-512 code path) as tests/golden/exp_torch_cpu.npz,
     so that the distance between the reference's exp and the correctly rounded anchor (<= 1 ulp) stays a tested fact.
"""
import ctypes
import os
import subprocess
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))

SLEEF_C = r'''
#include <math.h>
#include <stdint.h>
#include <string.h>
static float pow2if(int q) { uint32_t u = (uint32_t)(q + 0x7f) << 23; float f; memcpy(&f, &u, 4); return f; }
float expf_u10(float d) {   /* SLEEF 3.x sleefsimdsp.c xexpf, scalar, FMA build */
    int q = (int)rintf(d * 1.442695040888963407359924681001892137426645954152985934135449406931f);
    float s = fmaf((float)q, -0.693145751953125f, d), u;
    s = fmaf((float)q, -1.428606765330187045e-06f, s);
    u = 0.000198527617612853646278381f;
    u = fmaf(u, s, 0.00139304355252534151077271f);
    u = fmaf(u, s, 0.00833336077630519866943359f);
    u = fmaf(u, s, 0.0416664853692054748535156f);
    u = fmaf(u, s, 0.166666671633720397949219f);
    u = fmaf(u, s, 0.5f);
    u = 1.0f + fmaf(s * s, u, s);
    u = u * pow2if(q >> 1) * pow2if(q - (q >> 1));
    if (d < -104.f) u = 0.f;
    if (d > 100.f) u = INFINITY;
    return u;
}
void expf_u10_array(const float* x, float* y, long n) { for (long i = 0; i < n; ++i) y[i] = expf_u10(x[i]); }
'''


def main():
    from oracle import yolact_ref as R
    print(torch.__version__, 'mkl', torch.backends.mkl.is_available(), torch.backends.cpu.get_cpu_capability())
    with tempfile.TemporaryDirectory() as td:
        src, so = os.path.join(td, 's.c'), os.path.join(td, 's.so')
        open(src, 'w').write(SLEEF_C)
        subprocess.check_call(['gcc', '-O2', '-mfma', '-ffp-contract=off', '-shared', '-fPIC', src, '-o', so, '-lm'])
        sl = ctypes.CDLL(so)
        g = torch.Generator().manual_seed(0)
        for scale in (0.25, 1.0, 8.0):
            x = ((torch.rand(10_000_000, generator=g) * 2 - 1) * scale).contiguous()
            y = torch.exp(x)
            z = torch.empty_like(x)
            sl.expf_u10_array(ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(z.data_ptr()), ctypes.c_long(x.numel()))
            cr = R.expf_cr(x)
            d = (y.view(torch.int32) - cr.view(torch.int32)).abs()
            print(f'|x| < {scale}: torch.exp != SLEEF expf_u10 replica in {(y != z).float().mean().item():.4%}, '
                  f'!= correctly rounded in {(d != 0).float().mean().item():.4%} (max {int(d.max())} ulp)')
    g = torch.Generator().manual_seed(1)
    x = torch.cat([torch.randn(6144, generator=g) * 0.5, (torch.rand(2048, generator=g) * 2 - 1) * 20.0]).contiguous()
    np.savez(os.path.join(HERE, '..', 'tests', 'golden', 'exp_torch_cpu.npz'), x=x.numpy(), y=torch.exp(x).numpy())
    print('wrote tests/golden/exp_torch_cpu.npz')


def post_cases():
    """The inputs of tests/golden/post_*.npz (oracle/make_golden.py gen_post), regenerated from their seeds."""
    from oracle import yolact_ref as R
    a544 = R.anchors_for(544, [24, 48, 96, 192, 384])
    a128 = R.anchors_for(128, [int(128 / 544 * s) for s in (24, 48, 96, 192, 384)])
    yield ('dense544',) + tuple(R.synth_head_outputs(18525, seed=1)) + (a544,)
    yield ('sparse544',) + tuple(R.synth_head_outputs(18525, seed=2, bg_bias=9.0)) + (a544,)
    yield ('small128',) + tuple(R.synth_head_outputs(1023, proto_hw=32, seed=3, bg_bias=5.0)) + (a128,)
    cls, box, coef, proto = R.synth_head_outputs(1023, proto_hw=32, seed=5, bg_bias=7.5)
    box[0, ::3, 0] = -40.0
    box[0, ::3, 2] = -8.0
    yield 'degenerate128', cls, box, coef, proto, a128
    cls, box, coef, proto = R.synth_head_outputs(1023, proto_hw=32, seed=6, bg_bias=5.0)
    cls[0, 1::2] = cls[0, 0::2][: cls[0, 1::2].shape[0]]
    yield 'ties128', cls, box, coef, proto, a128


def freeze_decode_exp():
    """For every post-processing golden: the outputs of THIS host's `torch.exp` (the reference's call, utils/output_utils.py:150)
    on exactly the arguments the decode feeds it -- `box_p[keep, 2:] * 0.2` for every candidate over the score threshold.  With
    them the reference's boxes can be re-derived without MKL (tests/test_oracle_expf.py: decode with the frozen values == the
    frozen reference boxes bit for bit; the kernel's boxes differ exactly where the frozen value is not the correctly rounded
    exp)."""
    out = {}
    for tag, cls, box, coef, proto, anchors in post_cases():
        keep = cls[0].t()[1:].max(dim=0)[0] > 0.05
        x = box[0][keep][:, 2:] * 0.2
        out[f'{tag}_y'] = torch.exp(x).numpy()
        out[f'{tag}_n'] = np.array(int(keep.sum()))
        print(tag, int(keep.sum()), 'candidates')
    np.savez_compressed(os.path.join(HERE, '..', 'tests', 'golden', 'exp_decode_frozen.npz'), **out)
    print('wrote tests/golden/exp_decode_frozen.npz')


if __name__ == '__main__':
    import sys
    sys.path.insert(0, os.path.join(HERE, '..'))
    if 'decode' in sys.argv[1:]:
        freeze_decode_exp()
    else:
        main()
        freeze_decode_exp()
