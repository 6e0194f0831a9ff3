/* TEST INFRASTRUCTURE (CPU oracle) — never linked or loaded by the product.
 *
 * exp() of the box decode, reference utils/output_utils.py:150  (`torch.exp(box_p[:, 2:] * 0.2)`).
 *
 * What the reference executes there on this torch build (2.10.0+rocm7.0, USE_MKL=ON) is Intel MKL VML `vsExp` in VML_HA
 * mode (ATen/cpu/vml.h routes exp to MKL whenever MKL is enabled): closed source, ISA-dispatched (its results differ
 * between AVX-512 / AVX2 / non-Intel hosts — Intel's "conditional numerical reproducibility"), licence forbids
 * disassembly.  Measured in the build container (tools/exp_probe.py): torch.exp(float32) differs from the correctly
 * rounded value in 1.10 % of inputs (always by 1 ulp) and from an op-for-op replica of SLEEF `expf_u10` in 4.4-9 % of
 * inputs — so it is neither.  An op-for-op replica of the reference's exp is therefore not obtainable; the parity anchor
 * for this one primitive is the MATHEMATICAL value instead: `oracle_expf_cr` returns exp(x) rounded to nearest float.
 * It is evaluated in IEEE double with a FIXED operation sequence (rint, fma, mul — no libm call), which
 * csrc/postproc.hip reproduces instruction for instruction, so GPU and oracle agree bit for bit on every input, and
 * the double result (relative error < 3e-16) rounds to the correctly rounded float except when exp(x) lies within
 * ~3e-16 of a rounding boundary (probability ~1e-8 per input; tests/test_oracle_expf.py checks 1e6 inputs against
 * 50-digit arithmetic).
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

static const double LOG2E = 1.44269504088896338700e+00;
static const double LN2_HI = 6.93147180369123816490e-01;   /* 0x3fe62e42fee00000: 21 trailing zero bits, k*LN2_HI exact */
static const double LN2_LO = 1.90821492927058770002e-10;

float oracle_expf_cr(float xf) {
    if (xf != xf) return xf;
    double x = (double)xf;
    if (x > 89.0) return INFINITY;            /* exp(88.73) > FLT_MAX */
    if (x < -104.0) return 0.0f;              /* exp(-103.98) < 2^-150 */
    const double k = rint(x * LOG2E);
    double r = fma(k, -LN2_HI, x);
    r = fma(k, -LN2_LO, r);
    /* Taylor polynomial of degree 13, |r| <= 0.3466: remainder < 4e-18 */
    double p = 1.0 / 6227020800.0;
    p = fma(p, r, 1.0 / 479001600.0);
    p = fma(p, r, 1.0 / 39916800.0);
    p = fma(p, r, 1.0 / 3628800.0);
    p = fma(p, r, 1.0 / 362880.0);
    p = fma(p, r, 1.0 / 40320.0);
    p = fma(p, r, 1.0 / 5040.0);
    p = fma(p, r, 1.0 / 720.0);
    p = fma(p, r, 1.0 / 120.0);
    p = fma(p, r, 1.0 / 24.0);
    p = fma(p, r, 1.0 / 6.0);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    const int64_t bits = ((int64_t)k + 1023) << 52;   /* 2^k, k in [-151, 129]: a normal double */
    double s;
    memcpy(&s, &bits, 8);
    return (float)(p * s);                    /* one rounding double -> float (denormal floats included) */
}

void oracle_expf_cr_array(const float* x, float* y, long n) {
    for (long i = 0; i < n; ++i) y[i] = oracle_expf_cr(x[i]);
}
