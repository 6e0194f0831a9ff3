// Training-side NHWC helpers for gfx950: batch-norm (batch statistics) forward/backward, activation backward,
// per-channel column sums (bias gradients), max-pool and bilinear x2 backward, dgrad weight packing, SGD.
// All tensors are [M][C] fp32 with C contiguous (M = B*H*W); every kernel is HBM-bound and uses 16-byte accesses.
// Channel reductions accumulate in fp64 (MI355X has a full-rate fp64 vector pipe) so that var = E[x^2]-E[x]^2 is
// safe and the result does not depend on the reduction tree shape beyond fp64 rounding.
#include "ym_common.h"

namespace {

typedef double f64x4 __attribute__((ext_vector_type(4)));

inline int rows_grid(long long M, int C4, int rows_per_block_hint = 64) {
    // one block handles 256 threads = (256 / C4cols) row lanes; grid sized to cover M with ~64 rows per thread
    (void)C4;
    long long blocks = (M + rows_per_block_hint - 1) / rows_per_block_hint;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}

// ---- column reductions ------------------------------------------------------------------------------------
// Thread (cq, rl): channel quad cq = tid % CQ, row lane rl = tid / CQ, CQ = min(C/4, 256).  Each block walks rows
// rl + k*RL of its row range; partials are combined through LDS and then one fp64 atomicAdd per (block, channel).
// (bn_affine, the normalise + affine step in its one fixed operation order, lives in ym_common.h)
// MODE 0: sum x, sum x^2           (bn forward statistics)
// MODE 1: sum dz, sum dz*xhat      (bn backward; dz = dout masked by relu(out); out == nullptr: the mask is re-derived from y)
// MODE 2: sum dz                   (bias gradient; dz = dy * act'(y))
template <int MODE>
__global__ __launch_bounds__(256) void k_col_reduce(const float* __restrict__ a, const float* __restrict__ b,
                                                     const float* __restrict__ c, const float* __restrict__ mean,
                                                     const float* __restrict__ invstd, long long M, int C, int relu, int act,
                                                     double* __restrict__ out0, double* __restrict__ out1,
                                                     double* __restrict__ part = nullptr, const float* __restrict__ gamma = nullptr,
                                                     const float* __restrict__ beta = nullptr) {
    // part != nullptr: every workgroup stores its column sums to part[block][2][C] (no atomics: with a large grid the 2*C
    // contended fp64 atomics per workgroup were the bottleneck, which is why the atomic path caps the grid at 512) and
    // k_col_finish adds the blocks in order.
    const int C4 = C >> 2;
    const int CQ = C4 < 256 ? C4 : 256;
    const int RL = 256 / CQ;
    const int tid = threadIdx.x;
    const int cq_l = tid % CQ, rl = tid / CQ;
    __shared__ double red[2][256][4];
    for (int cq0 = 0; cq0 < C4; cq0 += CQ) {
        const int cq = cq0 + cq_l;
        f64x4 s0 = {0, 0, 0, 0}, s1 = {0, 0, 0, 0};
        if (rl < RL && cq < C4) {
            f32x4 mu = {0, 0, 0, 0}, is = {1, 1, 1, 1}, gm = {1, 1, 1, 1}, bt = {0, 0, 0, 0};
            if (MODE == 1) { mu = *reinterpret_cast<const f32x4*>(mean + cq * 4); is = *reinterpret_cast<const f32x4*>(invstd + cq * 4); }
            const bool remask = MODE == 1 && relu && b == nullptr;      // no saved `out`: mask = bn_affine(y) > 0
            if (remask) { gm = *reinterpret_cast<const f32x4*>(gamma + cq * 4); bt = *reinterpret_cast<const f32x4*>(beta + cq * 4); }
            const long long stride = (long long)gridDim.x * RL;
            long long m = (long long)blockIdx.x * RL + rl;
            if (MODE == 0) {
                // 4 independent rows in flight per lane
                for (; m + 3 * stride < M; m += 4 * stride) {
                    f32x4 x[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) x[u] = *reinterpret_cast<const f32x4*>(a + (size_t)(m + u * stride) * C + cq * 4);
#pragma unroll
                    for (int u = 0; u < 4; ++u)
#pragma unroll
                        for (int e = 0; e < 4; ++e) { const double v = x[u][e]; s0[e] += v; s1[e] += v * v; }
                }
            } else if (MODE == 1) {
                for (; m + 1 * stride < M; m += 2 * stride) {
                    f32x4 d[2], o[2], y[2];
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const size_t off = (size_t)(m + u * stride) * C + cq * 4;
                        d[u] = *reinterpret_cast<const f32x4*>(a + off);
                        y[u] = *reinterpret_cast<const f32x4*>(c + off);
                        if (remask) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) o[u][e] = bn_affine(y[u][e], mu[e], is[e], gm[e], bt[e]);
                        } else o[u] = relu ? *reinterpret_cast<const f32x4*>(b + off) : f32x4{1.f, 1.f, 1.f, 1.f};
                    }
#pragma unroll
                    for (int u = 0; u < 2; ++u)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const double dd = o[u][e] > 0.f ? d[u][e] : 0.f;
                            s0[e] += dd;
                            s1[e] += dd * (double)((y[u][e] - mu[e]) * is[e]);
                        }
                }
            }
            if (MODE == 2) {
                // 4 independent rows in flight per lane (this pass runs beside the weight gradients, or on their stream, with about one
                // resident wave per SIMD: one row at a time it waited on memory for 89 % of its cycles); rows are added in order
                for (; m + 3 * stride < M; m += 4 * stride) {
                    f32x4 d[4], y[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const size_t off = (size_t)(m + u * stride) * C + cq * 4;
                        d[u] = *reinterpret_cast<const f32x4*>(a + off);
                        if (act) y[u] = *reinterpret_cast<const f32x4*>(b + off);
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        if (act) {
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                d[u][e] = act == YM_ACT_RELU ? (y[u][e] > 0.f ? d[u][e] : 0.f) : d[u][e] * (1.f - y[u][e] * y[u][e]);
                        }
#pragma unroll
                        for (int e = 0; e < 4; ++e) s0[e] += (double)d[u][e];
                    }
                }
            }
            for (; m < M; m += stride) {
                const size_t off = (size_t)m * C + cq * 4;
                const f32x4 x = *reinterpret_cast<const f32x4*>(a + off);
                if (MODE == 0) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { const double v = x[e]; s0[e] += v; s1[e] += v * v; }
                } else if (MODE == 1) {
                    // a = dout, b = out (post-activation, for the relu mask), c = y_raw
                    f32x4 dz = x;
                    const f32x4 y = *reinterpret_cast<const f32x4*>(c + off);
                    if (remask) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) dz[e] = bn_affine(y[e], mu[e], is[e], gm[e], bt[e]) > 0.f ? dz[e] : 0.f;
                    } else if (relu) {
                        const f32x4 o = *reinterpret_cast<const f32x4*>(b + off);
#pragma unroll
                        for (int e = 0; e < 4; ++e) dz[e] = o[e] > 0.f ? dz[e] : 0.f;
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) { const double d = dz[e]; s0[e] += d; s1[e] += d * (double)((y[e] - mu[e]) * is[e]); }
                } else {
                    // a = dy, b = y (post-activation)
                    f32x4 dz = x;
                    if (act) {
                        const f32x4 y = *reinterpret_cast<const f32x4*>(b + off);
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            dz[e] = act == YM_ACT_RELU ? (y[e] > 0.f ? dz[e] : 0.f) : dz[e] * (1.f - y[e] * y[e]);
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) s0[e] += (double)dz[e];
                }
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) { red[0][tid][e] = s0[e]; red[1][tid][e] = s1[e]; }
        __syncthreads();
        if (rl == 0 && cq < C4) {
            for (int r = 1; r < RL; ++r)
#pragma unroll
                for (int e = 0; e < 4; ++e) { s0[e] += red[0][r * CQ + cq_l][e]; s1[e] += red[1][r * CQ + cq_l][e]; }
            if (part) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    part[((size_t)blockIdx.x * 2 + 0) * C + cq * 4 + e] = s0[e];
                    part[((size_t)blockIdx.x * 2 + 1) * C + cq * 4 + e] = s1[e];
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    atomicAdd(out0 + cq * 4 + e, s0[e]);
                    if (MODE != 2) atomicAdd(out1 + cq * 4 + e, s1[e]);
                }
            }
        }
        __syncthreads();
    }
}

// ordered sum of the per-workgroup partials: 16 channels x 16 block slices per workgroup, slices combined through LDS
__global__ __launch_bounds__(256) void k_col_finish(const double* __restrict__ part, int blocks, int C, double* __restrict__ out0,
                                                     double* __restrict__ out1, float* __restrict__ out0_f = nullptr) {
    __shared__ double red[2][16][17];
    const int cl = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl;
    double a = 0.0, b = 0.0;
    if (c < C) {
        int k = sl;
        for (; k + 48 < blocks; k += 64) {
            double x[4], y[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                x[u] = part[((size_t)(k + 16 * u) * 2) * C + c];
                y[u] = part[((size_t)(k + 16 * u) * 2 + 1) * C + c];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) { a += x[u]; b += y[u]; }
        }
        for (; k < blocks; k += 16) { a += part[((size_t)k * 2) * C + c]; b += part[((size_t)k * 2 + 1) * C + c]; }
    }
    red[0][sl][cl] = a;
    red[1][sl][cl] = b;
    __syncthreads();
    if (sl == 0 && c < C) {
        double sa = 0.0, sb = 0.0;
#pragma unroll
        for (int r = 0; r < 16; ++r) { sa += red[0][r][cl]; sb += red[1][r][cl]; }
        out0[c] = sa;
        out1[c] = sb;
        if (out0_f) out0_f[c] = (float)sa;          // (the bias gradient: saves the separate fp64 -> fp32 launch)
    }
}

__global__ void k_bn_finalize(const double* sum, const double* sumsq, long long M, float eps, float momentum, float* mean,
                              float* invstd, float* run_mean, float* run_var, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const double mu = sum[c] / (double)M;
    double var = sumsq[c] / (double)M - mu * mu;
    if (var < 0) var = 0;
    mean[c] = (float)mu;
    invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (run_mean) {
        const double unbiased = M > 1 ? var * (double)M / (double)(M - 1) : var;
        run_mean[c] = (float)((1.0 - momentum) * (double)run_mean[c] + momentum * mu);
        run_var[c] = (float)((1.0 - momentum) * (double)run_var[c] + momentum * unbiased);
    }
}

__global__ __launch_bounds__(256) void k_bn_apply(const float* __restrict__ y, const float* __restrict__ mean,
                                                   const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                   const float* __restrict__ beta, const float* __restrict__ residual,
                                                   int relu, float* __restrict__ out, long long M, int C) {
    const int C4 = C >> 2;
    const size_t total = (size_t)M * C4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int cq = (int)(i % C4);
        const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + cq * 4), is = *reinterpret_cast<const f32x4*>(invstd + cq * 4);
        const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + cq * 4), b = *reinterpret_cast<const f32x4*>(beta + cq * 4);
        f32x4 v = *reinterpret_cast<const f32x4*>(y + i * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = bn_affine(v[e], mu[e], is[e], g[e], b[e]);
        if (residual) v += *reinterpret_cast<const f32x4*>(residual + i * 4);
        if (relu) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = v[e] < 0.f ? 0.f : v[e];
        }
        *reinterpret_cast<f32x4*>(out + i * 4) = v;
    }
}

// k_bn_finalize + k_bn_apply in one launch (the statistics already sit in `sum` / `sumsq`, accumulated by the conv epilogue).
// The grid stride is a multiple of C/4, so a thread keeps the same four channels for its whole loop: it derives their mean /
// invstd once, in fp64 like k_bn_finalize (same floats), and the first C/4 threads of the grid also publish them and update the
// running statistics.  Needs C/4 to divide 256 * gridDim.x.
__global__ __launch_bounds__(256) void k_bn_finalize_apply(const float* __restrict__ y, const double* __restrict__ sum,
                                                            const double* __restrict__ sumsq, float eps, float momentum,
                                                            float* __restrict__ mean, float* __restrict__ invstd,
                                                            float* __restrict__ run_mean, float* __restrict__ run_var,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            const float* __restrict__ residual, int relu, float* __restrict__ out,
                                                            long long M, int C) {
    const int C4 = C >> 2;
    const size_t gt = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int cq = (int)(gt % C4);
    f32x4 mu, is;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int c = cq * 4 + e;
        const double m = sum[c] / (double)M;
        double var = sumsq[c] / (double)M - m * m;
        if (var < 0) var = 0;
        mu[e] = (float)m;
        is[e] = (float)(1.0 / sqrt(var + (double)eps));
        if (gt < (size_t)C4) {
            mean[c] = mu[e]; invstd[c] = is[e];
            if (run_mean) {
                const double unbiased = M > 1 ? var * (double)M / (double)(M - 1) : var;
                run_mean[c] = (float)((1.0 - momentum) * (double)run_mean[c] + momentum * m);
                run_var[c] = (float)((1.0 - momentum) * (double)run_var[c] + momentum * unbiased);
            }
        }
    }
    const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + cq * 4), b = *reinterpret_cast<const f32x4*>(beta + cq * 4);
    const size_t total = (size_t)M * C4;
    const size_t stride = (size_t)gridDim.x * 256;
    auto one = [&](size_t at, f32x4 v, const f32x4 r) __attribute__((always_inline)) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = bn_affine(v[e], mu[e], is[e], g[e], b[e]);
        if (residual) v += r;
        if (relu) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = v[e] < 0.f ? 0.f : v[e];
        }
        *reinterpret_cast<f32x4*>(out + at * 4) = v;
    };
    size_t i = gt;
    for (; i + 3 * stride < total; i += 4 * stride) {          // four quads per lane, loads first
        f32x4 v[4], r[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            v[u] = *reinterpret_cast<const f32x4*>(y + (i + u * stride) * 4);
            r[u] = residual ? *reinterpret_cast<const f32x4*>(residual + (i + u * stride) * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) one(i + u * stride, v[u], r[u]);
    }
    for (; i < total; i += stride)
        one(i, *reinterpret_cast<const f32x4*>(y + i * 4), residual ? *reinterpret_cast<const f32x4*>(residual + i * 4) : f32x4{0.f, 0.f, 0.f, 0.f});
}

// dy_raw = gamma*invstd * (dz - dbeta/M - xhat * dgamma/M); dres = dz
__global__ __launch_bounds__(256) void k_bn_bwd_apply(const float* __restrict__ dout, const float* __restrict__ out,
                                                       const float* __restrict__ y, const float* __restrict__ mean,
                                                       const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, const double* __restrict__ dbeta,
                                                       const double* __restrict__ dgamma, int relu, float* __restrict__ dy,
                                                       float* __restrict__ dres, long long M, int C, float* __restrict__ dgamma_f,
                                                       float* __restrict__ dbeta_f) {
    const int C4 = C >> 2;
    const size_t total = (size_t)M * C4;
    const float invM = 1.f / (float)M;
    if (blockIdx.x == 0)
        for (int c = threadIdx.x; c < C; c += 256) { dgamma_f[c] = (float)dgamma[c]; dbeta_f[c] = (float)dbeta[c]; }
    // When the grid stride is a multiple of C/4 a thread keeps its channel quad for the whole tensor: the seven per-channel vectors
    // (mean, invstd, gamma, beta, the two fp64 sums) are then loaded and converted ONCE instead of per 16 bytes of dout (they were
    // 8 of the 11 load instructions of an iteration of this HBM-bound pass).  Same arithmetic, same order.
    const size_t stride = (size_t)gridDim.x * 256;
    const bool fixed_c = stride % (size_t)C4 == 0;
    f32x4 mu, is, g, bt = {0.f, 0.f, 0.f, 0.f}, dbf, dgf;
    auto load_channel = [&](int cq) __attribute__((always_inline)) {
        mu = *reinterpret_cast<const f32x4*>(mean + cq * 4);
        is = *reinterpret_cast<const f32x4*>(invstd + cq * 4);
        g = *reinterpret_cast<const f32x4*>(gamma + cq * 4);
        if (relu && !out) bt = *reinterpret_cast<const f32x4*>(beta + cq * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { dbf[e] = (float)dbeta[cq * 4 + e]; dgf[e] = (float)dgamma[cq * 4 + e]; }
    };
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (fixed_c && i < total) load_channel((int)(i % C4));
    auto one = [&](size_t at, f32x4 dz, const f32x4 yy, const f32x4 o) __attribute__((always_inline)) {
        if (relu && out) {
#pragma unroll
            for (int e = 0; e < 4; ++e) dz[e] = o[e] > 0.f ? dz[e] : 0.f;
        } else if (relu) {                                        // no saved `out`: the same affine as the forward pass
#pragma unroll
            for (int e = 0; e < 4; ++e) dz[e] = bn_affine(yy[e], mu[e], is[e], g[e], bt[e]) > 0.f ? dz[e] : 0.f;
        }
        if (dres) *reinterpret_cast<f32x4*>(dres + at * 4) = dz;
        f32x4 r;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float xh = (yy[e] - mu[e]) * is[e];
            r[e] = g[e] * is[e] * (dz[e] - dbf[e] * invM - xh * dgf[e] * invM);
        }
        *reinterpret_cast<f32x4*>(dy + at * 4) = r;
    };
    // U quads per thread with ALL their loads issued first (2-3 x U 16-byte loads in flight per lane): in the training step this pass
    // shares the CUs with the weight-gradient stream and gets few resident waves, so its bandwidth comes from loads per wave, not
    // from the number of waves (one quad per thread ran 2x its stand-alone time there).  Same arithmetic per element.
    constexpr int U = 4;
    if (fixed_c) {
        for (; i + (size_t)(U - 1) * stride < total; i += (size_t)U * stride) {
            f32x4 dz[U], yy[U], o[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const size_t at = i + (size_t)u * stride;
                dz[u] = *reinterpret_cast<const f32x4*>(dout + at * 4);
                yy[u] = *reinterpret_cast<const f32x4*>(y + at * 4);
                if (relu && out) o[u] = *reinterpret_cast<const f32x4*>(out + at * 4);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) one(i + (size_t)u * stride, dz[u], yy[u], o[u]);
        }
    }
    for (; i < total; i += stride) {
        if (!fixed_c) load_channel((int)(i % C4));
        f32x4 o = {0.f, 0.f, 0.f, 0.f};
        if (relu && out) o = *reinterpret_cast<const f32x4*>(out + i * 4);
        one(i, *reinterpret_cast<const f32x4*>(dout + i * 4), *reinterpret_cast<const f32x4*>(y + i * 4), o);
    }
}

// grid of the BatchNorm-backward apply pass: four quads per thread (see the kernel), a multiple of 2 blocks so that the grid stride stays
// a multiple of C / 4 for every channel count of the nets (<= 2048)
inline int bn_apply_grid(size_t total) {
    static int per_thread = 0;                       // YM_BN_APPLY_QUADS=1: one quad per thread, the launch shape of rounds 1-5 (A/B)
    if (per_thread == 0) { const char* e = getenv("YM_BN_APPLY_QUADS"); per_thread = (e && atoi(e) > 0) ? atoi(e) : 4; }
    size_t g = (total + 256 * (size_t)per_thread - 1) / (256 * (size_t)per_thread);
    if (g > 8192) g = 8192;
    if (g < 2) g = 2;
    return (int)((g + 1) & ~(size_t)1);
}

__global__ __launch_bounds__(256) void k_act_bwd(const float* __restrict__ dy, const float* __restrict__ y, int act,
                                                  float* __restrict__ dz, size_t total4) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (size_t)gridDim.x * 256) {
        f32x4 d = *reinterpret_cast<const f32x4*>(dy + i * 4);
        const f32x4 v = *reinterpret_cast<const f32x4*>(y + i * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) d[e] = act == YM_ACT_RELU ? (v[e] > 0.f ? d[e] : 0.f) : d[e] * (1.f - v[e] * v[e]);
        *reinterpret_cast<f32x4*>(dz + i * 4) = d;
    }
}

__global__ void k_f64_to_f32(const double* in, float* out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (float)in[i];
}

// MaxPool2d(3,2,1) backward: every input pixel gathers from the (at most 4) windows that contain it and takes the
// gradient of those whose FIRST maximum (scan order kh, kw — ATen's tie rule) is this pixel.
__global__ __launch_bounds__(256) void k_maxpool_bwd(const float* __restrict__ x, const float* __restrict__ dy,
                                                      float* __restrict__ dx, int B, int H, int W, int C4, int Ho, int Wo) {
    const size_t total = (size_t)B * H * W * C4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % C4);
        size_t t = i / C4;
        const int iw = (int)(t % W); t /= W;
        const int ih = (int)(t % H);
        const int b = (int)(t / H);
        const f32x4 xv = *reinterpret_cast<const f32x4*>(x + i * 4);
        f32x4 g = {0.f, 0.f, 0.f, 0.f};
        // windows containing the pixel: oh*2-1 <= ih <= oh*2+1  <=>  ih/2 <= oh <= (ih+1)/2
        for (int oh = ih / 2; oh <= (ih + 1) / 2 && oh < Ho; ++oh) {
            for (int ow = iw / 2; ow <= (iw + 1) / 2 && ow < Wo; ++ow) {
                // is (ih, iw) the first max of window (oh, ow)?
                bool first[4] = {true, true, true, true};
#pragma unroll
                for (int dyy = 0; dyy < 3; ++dyy) {
                    const int y2 = oh * 2 - 1 + dyy;
                    if ((unsigned)y2 >= (unsigned)H) continue;
#pragma unroll
                    for (int dxx = 0; dxx < 3; ++dxx) {
                        const int x2 = ow * 2 - 1 + dxx;
                        if ((unsigned)x2 >= (unsigned)W) continue;
                        if (y2 == ih && x2 == iw) continue;
                        const f32x4 o = *reinterpret_cast<const f32x4*>(x + ((((size_t)b * H + y2) * W + x2) * C4 + c) * 4);
                        const bool before = (y2 < ih) || (y2 == ih && x2 < iw);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            // another element wins if it is larger, or equal and earlier in scan order (NaN: ATen keeps NaN max)
                            if (o[e] > xv[e] || (o[e] == xv[e] && before) || (o[e] != o[e] && !(xv[e] != xv[e]))) first[e] = false;
                        }
                    }
                }
                const f32x4 d = *reinterpret_cast<const f32x4*>(dy + ((((size_t)b * Ho + oh) * Wo + ow) * C4 + c) * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) g[e] += first[e] ? d[e] : 0.f;
            }
        }
        *reinterpret_cast<f32x4*>(dx + i * 4) = g;
    }
}

// The same gather with the argmax positions saved by the forward (ym_maxpool3x3s2_fwd_idx): per input pixel at most 4 index
// words and 4 gradient vectors instead of re-scanning up to 4 windows of 9 (32 vector loads).
__global__ __launch_bounds__(256) void k_maxpool_bwd_idx(const uint8_t* __restrict__ idx, const float* __restrict__ dy,
                                                          float* __restrict__ dx, int B, int H, int W, int C4, int Ho, int Wo) {
    const size_t total = (size_t)B * H * W * C4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % C4);
        size_t t = i / C4;
        const int iw = (int)(t % W); t /= W;
        const int ih = (int)(t % H);
        const int b = (int)(t / H);
        f32x4 g = {0.f, 0.f, 0.f, 0.f};
        for (int oh = ih / 2; oh <= (ih + 1) / 2 && oh < Ho; ++oh) {
            for (int ow = iw / 2; ow <= (iw + 1) / 2 && ow < Wo; ++ow) {
                const uint32_t me = (uint32_t)((ih - (oh * 2 - 1)) * 3 + (iw - (ow * 2 - 1)));
                const size_t o = (((size_t)b * Ho + oh) * Wo + ow) * C4 + c;
                const uint32_t a = reinterpret_cast<const uint32_t*>(idx)[o];
                const f32x4 d = *reinterpret_cast<const f32x4*>(dy + o * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) g[e] += ((a >> (8 * e)) & 0xFFu) == me ? d[e] : 0.f;
            }
        }
        *reinterpret_cast<f32x4*>(dx + i * 4) = g;
    }
}

__device__ __forceinline__ void bil_src(int dst, int in_sz, int out_sz, int align, int& i0, int& i1, float& l1) {
    float src;
    if (align) {
        const float sc = out_sz > 1 ? (float)(in_sz - 1) / (float)(out_sz - 1) : 0.f;
        src = sc * dst;
    } else {
        src = 0.5f * (dst + 0.5f) - 0.5f;
        if (src < 0.f) src = 0.f;
    }
    i0 = (int)src;
    if (i0 > in_sz - 1) i0 = in_sz - 1;
    i1 = i0 + ((i0 < in_sz - 1) ? 1 : 0);
    l1 = src - (float)i0;
}

// Bilinear x2 backward as a GATHER (deterministic): every input pixel sums the contributions of the <= 4x4 output
// pixels whose interpolation footprint contains it.
__global__ __launch_bounds__(256) void k_bilinear2x_bwd(const float* __restrict__ dy, float* __restrict__ dx, int B, int H,
                                                         int W, int C4, int align) {
    const int Ho = 2 * H, Wo = 2 * W;
    const size_t total = (size_t)B * H * W * C4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % C4);
        size_t t = i / C4;
        const int iw = (int)(t % W); t /= W;
        const int ih = (int)(t % H);
        const int b = (int)(t / H);
        f32x4 g = {0.f, 0.f, 0.f, 0.f};
        const int oy_lo = max(0, 2 * ih - 3), oy_hi = min(Ho - 1, 2 * ih + 3);
        const int ox_lo = max(0, 2 * iw - 3), ox_hi = min(Wo - 1, 2 * iw + 3);
        for (int oy = oy_lo; oy <= oy_hi; ++oy) {
            int y0, y1; float ly;
            bil_src(oy, H, Ho, align, y0, y1, ly);
            const float wy = (y0 == ih ? 1.f - ly : 0.f) + (y1 == ih ? ly : 0.f);
            if (wy == 0.f) continue;
            for (int ox = ox_lo; ox <= ox_hi; ++ox) {
                int x0, x1; float lx;
                bil_src(ox, W, Wo, align, x0, x1, lx);
                const float wx = (x0 == iw ? 1.f - lx : 0.f) + (x1 == iw ? lx : 0.f);
                if (wx == 0.f) continue;
                const f32x4 d = *reinterpret_cast<const f32x4*>(dy + ((((size_t)b * Ho + oy) * Wo + ox) * C4 + c) * 4);
                g += d * (wy * wx);
            }
        }
        *reinterpret_cast<f32x4*>(dx + i * 4) = g;
    }
}

// OIHW -> [Cin][KH][KW][cout_pad] (the "transposed" packing read by the dgrad gather), zero padded
__global__ void k_pack_weight_dgrad(const float* __restrict__ w, float* __restrict__ out, int Cout, int Cin, int KH, int KW,
                                    int cout_pad) {
    const size_t total = (size_t)Cin * KH * KW * cout_pad;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int co = (int)(i % cout_pad);
        size_t t = i / cout_pad;
        const int kw = (int)(t % KW); t /= KW;
        const int kh = (int)(t % KH);
        const int ci = (int)(t / KH);
        out[i] = co < Cout ? w[(((size_t)co * Cin + ci) * KH + kh) * KW + kw] : 0.f;
    }
}

// Every layer's weight packing (forward [Cout_pad][k_pad] and dgrad [Cin][KH][KW][cout_pad] images) in ONE launch after an
// optimizer step: workgroup = one chunk of one item's destination (forward image: 1024 consecutive elements; dgrad image: a
// 32 x 32 tile of the transpose); the item is found by bisection over the chunk prefix in the device-side table.
__global__ __launch_bounds__(256) void k_pack_weights_batch(const ym_pack_item* __restrict__ items, int n_items) {
    int lo = 0, hi = n_items - 1;
    while (lo < hi) {                                            // last item whose first_chunk <= blockIdx.x
        const int mid = (lo + hi + 1) >> 1;
        if (items[mid].first_chunk <= blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const ym_pack_item it = items[lo];
    const size_t base = (size_t)(blockIdx.x - it.first_chunk) * 1024;
    const int KHW = it.kh * it.kw;
    if (it.kind == 0) {                                          // forward image: [rows][k_pad], k = tap * cin_pad + c
        const size_t total = (size_t)it.rows * it.pad_b;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const size_t i = base + u * 256 + threadIdx.x;
            if (i >= total) break;
            const int n = (int)(i / it.pad_b), k = (int)(i - (size_t)n * it.pad_b);
            const int tap = k / it.pad_a, c = k - tap * it.pad_a;
            float v = 0.f;
            if (n < it.cout && tap < KHW && c < it.cin) v = it.src[((size_t)n * it.cin + c) * KHW + tap];
            it.dst[i] = v;
        }
    } else {                                                     // dgrad image: [Cin][KH][KW][cout_pad] = the transpose of
        // src viewed as [cout][R], R = cin*KH*KW, zero padded to cout_pad columns: one 32 x 32 tile per workgroup through LDS,
        // 128-byte rows on both sides (a chunk of this kind is a tile, not 1024 consecutive elements)
        __shared__ float tile[32][33];
        const int R = it.cin * KHW, tiles_co = it.pad_a >> 5;
        const int local = (int)(blockIdx.x - it.first_chunk);
        const int r0 = (local / tiles_co) << 5, co0 = (local % tiles_co) << 5;
        const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int co = co0 + ty + 8 * k, r = r0 + tx;
            tile[ty + 8 * k][tx] = (co < it.cout && r < R) ? it.src[(size_t)co * R + r] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int r = r0 + ty + 8 * k;
            if (r < R) it.dst[(size_t)r * it.pad_a + co0 + tx] = tile[tx][ty + 8 * k];
        }
    }
}

// SGD with momentum and weight decay over a flat parameter buffer (torch.optim.SGD semantics, dampening 0, no nesterov):
//   g = grad + wd * p ; buf = first ? g : mom * buf + g ; p -= lr * buf
__global__ __launch_bounds__(256) void k_sgd(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ buf,
                                              size_t n, float lr, float mom, float wd, int first) {
#pragma clang fp contract(off)          // (HIP's __fmul_rn / __fadd_rn are plain operators: only the pragma keeps the product and the sum apart)
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        // the roundings of torch.optim.SGD's own device kernels, so that the reference loop (train.py:61,130) and this launch give
        // the same bits: grad.add(param, alpha=wd) and param.add_(buf, alpha=-lr) are one fused multiply-add each, while
        // buf.mul_(momentum).add_(grad) rounds the product before the sum (tests/test_gpu_reference_loop.py)
        const float gg = __builtin_fmaf(wd, p[i], g[i]);
        const float mb = mom * buf[i];
        const float b = first ? gg : mb + gg;
        buf[i] = b;
        p[i] = __builtin_fmaf(-lr, b, p[i]);
    }
}

inline int ew_grid(size_t total, int cap = 8192) {
    size_t g = (total + 255) / 256;
    if (g > (size_t)cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

}  // namespace

extern "C" int ym_bn_train_fwd(const float* y, int64_t M, int C, const float* gamma, const float* beta, float eps,
                               float momentum, float* running_mean, float* running_var, const float* residual, int relu,
                               float* out, float* save_mean, float* save_invstd, void* workspace, size_t workspace_bytes,
                               ym_stream_t s) {
    YM_REQUIRE(y && gamma && beta && out && save_mean && save_invstd && workspace, "bn_train_fwd: null pointer");
    YM_REQUIRE(M > 0 && C > 0 && C % 4 == 0, "bn_train_fwd: C %% 4 != 0");
    if (workspace_bytes < (size_t)C * 16) { ym_set_error("bn_train_fwd: workspace < %d B", C * 16); return YM_ENOSPC; }
    hipStream_t st = (hipStream_t)s;
    double* sum = (double*)workspace;
    double* sumsq = sum + C;
    (void)hipMemsetAsync(sum, 0, (size_t)C * 16, st);
    const int CQ = (C / 4) < 256 ? (C / 4) : 256, RL = 256 / CQ;
    int grid = (int)((M + (long long)RL * 16 - 1) / ((long long)RL * 16));
    if (grid > 512) grid = 512;       // every block ends with 2*C contended fp64 atomics: keep the block count low
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(k_col_reduce<0>, dim3(grid), dim3(256), 0, st, y, nullptr, nullptr, nullptr, nullptr, (long long)M, C, 0,
                       0, sum, sumsq);
    hipLaunchKernelGGL(k_bn_finalize, dim3(ym_cdiv(C, 256)), dim3(256), 0, st, sum, sumsq, (long long)M, eps, momentum,
                       save_mean, save_invstd, running_mean, running_var, C);
    hipLaunchKernelGGL(k_bn_apply, dim3(ew_grid((size_t)M * (C / 4))), dim3(256), 0, st, y, save_mean, save_invstd, gamma, beta,
                       residual, relu, out, (long long)M, C);
    return ym_check_launch("bn_train_fwd");
}

extern "C" int ym_bn_train_fwd_stats(const float* y, int64_t M, int C, const float* gamma, const float* beta, float eps,
                                     float momentum, float* running_mean, float* running_var, const float* residual,
                                     int relu, float* out, float* save_mean, float* save_invstd, const void* stats,
                                     ym_stream_t s) {
    YM_REQUIRE(y && gamma && beta && out && save_mean && save_invstd && stats, "bn_train_fwd_stats: null pointer");
    YM_REQUIRE(M > 0 && C > 0 && C % 4 == 0, "bn_train_fwd_stats: C %% 4 != 0");
    hipStream_t st = (hipStream_t)s;
    const double* sum = (const double*)stats;
    {   // one launch when a grid exists whose stride (256 * grid) is a multiple of C/4 and covers C/4 threads
        const int C4 = C / 4;
        const size_t total = (size_t)M * C4;
        long long grid = (long long)((total + 256 * 8 - 1) / (256 * 8));        // >= 8 float4 per thread: the per-thread fp64
        if (grid > 2048) grid = 2048;                                             // statistics stay a small share of the work
        int step = 1;                                                            // smallest g with C4 | 256 * g
        while ((256ll * step) % C4 != 0 && step <= 64) ++step;
        if ((256ll * step) % C4 == 0) {
            grid = (grid + step - 1) / step * step;
            while (grid * 256 < C4) grid += step;
            hipLaunchKernelGGL(k_bn_finalize_apply, dim3((unsigned)grid), dim3(256), 0, st, y, sum, sum + C, eps, momentum, save_mean,
                               save_invstd, running_mean, running_var, gamma, beta, residual, relu, out, (long long)M, C);
            return ym_check_launch("bn_train_fwd_stats");
        }
    }
    hipLaunchKernelGGL(k_bn_finalize, dim3(ym_cdiv(C, 256)), dim3(256), 0, st, sum, sum + C, (long long)M, eps, momentum,
                       save_mean, save_invstd, running_mean, running_var, C);
    hipLaunchKernelGGL(k_bn_apply, dim3(ew_grid((size_t)M * (C / 4))), dim3(256), 0, st, y, save_mean, save_invstd, gamma, beta,
                       residual, relu, out, (long long)M, C);
    return ym_check_launch("bn_train_fwd_stats");
}

extern "C" size_t ym_bn_train_bwd_workspace_bytes(int64_t M, int C) {
    const int CQ = (C / 4) < 256 ? (C / 4) : 256, RL = 256 / CQ;
    long long grid = (M + (long long)RL * 16 - 1) / ((long long)RL * 16);
    if (grid < 1) grid = 1;
    if (grid > 1024) grid = 1024;
    return (size_t)C * 16 + (size_t)grid * 2 * C * 8;
}

extern "C" int ym_bn_train_bwd(const float* dout, const float* out, const float* y, int64_t M, int C, const float* gamma,
                               const float* beta, const float* save_mean, const float* save_invstd, int relu, float* dy,
                               float* dres, float* dgamma, float* dbeta, void* workspace, size_t workspace_bytes, ym_stream_t s) {
    YM_REQUIRE(dout && y && gamma && save_mean && save_invstd && dy && dgamma && dbeta && workspace && (out || !relu || beta),
               "bn_train_bwd: null pointer (relu needs `out`, or `beta` to re-derive the mask from y)");
    YM_REQUIRE(M > 0 && C > 0 && C % 4 == 0, "bn_train_bwd: C %% 4 != 0");
    if (workspace_bytes < (size_t)C * 16) { ym_set_error("bn_train_bwd: workspace < %d B", C * 16); return YM_ENOSPC; }
    hipStream_t st = (hipStream_t)s;
    double* db = (double*)workspace;
    double* dg = db + C;
    const int CQ = (C / 4) < 256 ? (C / 4) : 256, RL = 256 / CQ;
    int grid = (int)((M + (long long)RL * 16 - 1) / ((long long)RL * 16));
    if (grid < 1) grid = 1;
    const int big = grid > 1024 ? 1024 : grid;
    if (workspace_bytes >= (size_t)C * 16 + (size_t)big * 2 * C * 8) {           // two-stage: partials, then an ordered sum
        double* part = dg + C;
        hipLaunchKernelGGL(k_col_reduce<1>, dim3(big), dim3(256), 0, st, dout, out, y, save_mean, save_invstd, (long long)M, C,
                           relu, 0, db, dg, part, gamma, beta);
        hipLaunchKernelGGL(k_col_finish, dim3(ym_cdiv(C, 16)), dim3(256), 0, st, part, big, C, db, dg);
    } else {
        if (grid > 512) grid = 512;
        (void)hipMemsetAsync(db, 0, (size_t)C * 16, st);
        hipLaunchKernelGGL(k_col_reduce<1>, dim3(grid), dim3(256), 0, st, dout, out, y, save_mean, save_invstd, (long long)M, C,
                           relu, 0, db, dg, (double*)nullptr, gamma, beta);
    }
    hipLaunchKernelGGL(k_bn_bwd_apply, dim3(bn_apply_grid((size_t)M * (C / 4))), dim3(256), 0, st, dout, out, y, save_mean,
                       save_invstd, gamma, beta, db, dg, relu, dy, dres, (long long)M, C, dgamma, dbeta);
    return ym_check_launch("bn_train_bwd");
}

extern "C" int ym_bn_train_bwd_apply(const float* dout, const float* out, const float* y, int64_t M, int C, const float* gamma,
                                     const float* beta, const float* save_mean, const float* save_invstd, int relu, float* dy,
                                     float* dres, float* dgamma, float* dbeta, const void* stats, ym_stream_t s) {
    YM_REQUIRE(dout && y && gamma && save_mean && save_invstd && dy && dgamma && dbeta && stats && (out || !relu || beta),
               "bn_train_bwd_apply: null pointer (relu needs `out`, or `beta` to re-derive the mask from y)");
    YM_REQUIRE(M > 0 && C > 0 && C % 4 == 0, "bn_train_bwd_apply: C %% 4 != 0");
    const double* db = (const double*)stats;
    hipLaunchKernelGGL(k_bn_bwd_apply, dim3(bn_apply_grid((size_t)M * (C / 4))), dim3(256), 0, (hipStream_t)s, dout, out, y, save_mean,
                       save_invstd, gamma, beta, db, db + C, relu, dy, dres, (long long)M, C, dgamma, dbeta);
    return ym_check_launch("bn_train_bwd_apply");
}

extern "C" int ym_act_bias_bwd(const float* dy, const float* y, int64_t M, int C, int act, float* dz, float* dbias,
                               void* workspace, size_t workspace_bytes, ym_stream_t s) {
    YM_REQUIRE(dy && (y || act == YM_ACT_NONE) && M > 0 && C > 0 && C % 4 == 0, "act_bias_bwd: bad args");
    hipStream_t st = (hipStream_t)s;
    if (dz && act != YM_ACT_NONE)
        hipLaunchKernelGGL(k_act_bwd, dim3(ew_grid((size_t)M * (C / 4))), dim3(256), 0, st, dy, y, act, dz, (size_t)M * (C / 4));
    if (dbias) {
        YM_REQUIRE(workspace, "act_bias_bwd: workspace");
        if (workspace_bytes < (size_t)C * 8) { ym_set_error("act_bias_bwd: workspace < %d B", C * 8); return YM_ENOSPC; }
        double* acc = (double*)workspace;
        const int CQ = (C / 4) < 256 ? (C / 4) : 256, RL = 256 / CQ;
        int grid = (int)((M + (long long)RL * 32 - 1) / ((long long)RL * 32));
        if (grid < 1) grid = 1;
        const int big = grid > 1024 ? 1024 : grid;
        if (workspace_bytes >= (size_t)C * 16 + (size_t)big * 2 * C * 8) {
            // enough scratch for per-workgroup partials + an ordered finish (as in ym_bn_train_bwd): no contended fp64 atomics
            double* part = acc + 2 * C;
            hipLaunchKernelGGL(k_col_reduce<2>, dim3(big), dim3(256), 0, st, dy, y, nullptr, nullptr, nullptr, (long long)M, C, 0,
                               act, acc, acc + C, part);
            hipLaunchKernelGGL(k_col_finish, dim3(ym_cdiv(C, 16)), dim3(256), 0, st, part, big, C, acc, acc + C, dbias);
            return ym_check_launch("act_bias_bwd");
        } else {
            if (grid > 2048) grid = 2048;
            (void)hipMemsetAsync(acc, 0, (size_t)C * 8, st);
            hipLaunchKernelGGL(k_col_reduce<2>, dim3(grid), dim3(256), 0, st, dy, y, nullptr, nullptr, nullptr, (long long)M, C, 0,
                               act, acc, nullptr);
        }
        hipLaunchKernelGGL(k_f64_to_f32, dim3(ym_cdiv(C, 256)), dim3(256), 0, st, acc, dbias, C);
    }
    return ym_check_launch("act_bias_bwd");
}

extern "C" int ym_maxpool3x3s2_bwd(const float* x, const float* dy, float* dx, int B, int H, int W, int C, ym_stream_t s) {
    YM_REQUIRE(x && dy && dx && B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "maxpool_bwd: bad args");
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    hipLaunchKernelGGL(k_maxpool_bwd, dim3(ew_grid((size_t)B * H * W * (C / 4))), dim3(256), 0, (hipStream_t)s, x, dy, dx, B, H,
                       W, C / 4, Ho, Wo);
    return ym_check_launch("maxpool_bwd");
}

extern "C" int ym_maxpool3x3s2_bwd_idx(const uint8_t* idx, const float* dy, float* dx, int B, int H, int W, int C, ym_stream_t s) {
    YM_REQUIRE(idx && dy && dx && B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "maxpool_bwd_idx: bad args");
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    hipLaunchKernelGGL(k_maxpool_bwd_idx, dim3(ew_grid((size_t)B * H * W * (C / 4))), dim3(256), 0, (hipStream_t)s, idx, dy, dx, B,
                       H, W, C / 4, Ho, Wo);
    return ym_check_launch("maxpool_bwd_idx");
}

extern "C" int ym_bilinear2x_bwd(const float* dy, float* dx, int B, int H, int W, int C, int align_corners, ym_stream_t s) {
    YM_REQUIRE(dy && dx && B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "bilinear2x_bwd: bad args");
    hipLaunchKernelGGL(k_bilinear2x_bwd, dim3(ew_grid((size_t)B * H * W * (C / 4))), dim3(256), 0, (hipStream_t)s, dy, dx, B, H,
                       W, C / 4, align_corners ? 1 : 0);
    return ym_check_launch("bilinear2x_bwd");
}

extern "C" int ym_pack_conv_weight_dgrad(const float* w_oihw, float* w_packed, int Cout, int Cin, int KH, int KW, int cout_pad,
                                         ym_stream_t s) {
    YM_REQUIRE(w_oihw && w_packed && Cout > 0 && Cin > 0 && cout_pad >= Cout && cout_pad % 32 == 0,
               "pack_conv_weight_dgrad: bad args (cout_pad must be a multiple of 32)");
    const size_t total = (size_t)Cin * KH * KW * cout_pad;
    hipLaunchKernelGGL(k_pack_weight_dgrad, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)s, w_oihw, w_packed, Cout, Cin, KH,
                       KW, cout_pad);
    return ym_check_launch("pack_conv_weight_dgrad");
}

extern "C" int ym_pack_conv_weights_batch(const ym_pack_item* items_dev, int n_items, int total_chunks, ym_stream_t s) {
    YM_REQUIRE(items_dev && n_items > 0 && total_chunks > 0, "pack_conv_weights_batch: bad args");
    hipLaunchKernelGGL(k_pack_weights_batch, dim3(total_chunks), dim3(256), 0, (hipStream_t)s, items_dev, n_items);
    return ym_check_launch("pack_conv_weights_batch");
}

extern "C" int ym_sgd_step(float* param, const float* grad, float* momentum_buf, int64_t n, float lr, float momentum,
                           float weight_decay, int first_step, ym_stream_t s) {
    YM_REQUIRE(param && grad && momentum_buf && n > 0, "sgd_step: bad args");
    hipLaunchKernelGGL(k_sgd, dim3(ew_grid((size_t)n)), dim3(256), 0, (hipStream_t)s, param, grad, momentum_buf, (size_t)n, lr,
                       momentum, weight_decay, first_step);
    return ym_check_launch("sgd_step");
}

// ---------------------------------------------------------------------------------------------------------------------------
// Backward glue of the fused prediction head (modules/yolact.py:27-30,149-157): the loss returns gradients of the CONCATENATED
// [B][N][*] tensors; the data / weight gradient convs of the 351-channel head want dz [rows][pitch] per level.
// ---------------------------------------------------------------------------------------------------------------------------
namespace {
struct HeadLevels { int row[6]; int anchor[6]; int nlev; };

__global__ __launch_bounds__(256) void k_head_grad_gather(const float* __restrict__ dclass, const float* __restrict__ dbox,
                                                           const float* __restrict__ dcoef, const float* __restrict__ coef, int B,
                                                           int N, int nc, int cd, int na, const HeadLevels lv, int pitch,
                                                           const float* __restrict__ g_scale, float* __restrict__ dz) {
    const int c_conf = na * nc, c_box = na * 4, c_coef = na * cd;
    const long long total = (long long)lv.row[lv.nlev] * pitch;
    const float gc = g_scale ? g_scale[0] : 1.f, gb = g_scale ? g_scale[1] : 1.f, gm = g_scale ? g_scale[2] : 1.f;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const int r = (int)(e / pitch), c = (int)(e - (long long)r * pitch);
        int l = 0;
#pragma unroll
        for (int q = 1; q < 5; ++q)
            if (q < lv.nlev && r >= lv.row[q]) l = q;
        const int hw = (lv.row[l + 1] - lv.row[l]) / B;
        const int local = r - lv.row[l];
        const int b = local / hw, pix = local - b * hw;
        const long long a0 = (long long)b * N + lv.anchor[l] + (long long)pix * na;      // first anchor of this pixel
        float v = 0.f;
        if (c < c_conf) v = dclass[a0 * nc + c] * gc;
        else if (c < c_conf + c_box) v = dbox[a0 * 4 + (c - c_conf)] * gb;
        else if (c < c_conf + c_box + c_coef) {
            const long long i = a0 * cd + (c - c_conf - c_box);
            const float t = coef[i];
            v = dcoef[i] * gm * (1.f - t * t);                                             // tanh'
        }
        dz[e] = v;
    }
}

__global__ void k_scatter3(const float* __restrict__ src, float* d0, int n0, float* d1, int n1, float* d2, int n2, int accumulate) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n0 + n1 + n2) return;
    float* dst = i < n0 ? d0 + i : (i < n0 + n1 ? d1 + (i - n0) : d2 + (i - n0 - n1));
    *dst = accumulate ? *dst + src[i] : src[i];
}
}  // namespace

extern "C" int ym_head_grad_gather(const float* dclass, const float* dbox, const float* dcoef, const float* coef, int B, int N, int nc,
                                   int cd, int na, int nlev, const int32_t* lev_row, const int32_t* lev_anchor, int pitch,
                                   const float* g_scale, float* dz, ym_stream_t s) {
    YM_REQUIRE(dclass && dbox && dcoef && coef && dz && lev_row && lev_anchor, "head_grad_gather: null pointer");
    YM_REQUIRE(nlev >= 1 && nlev <= 5 && B > 0 && pitch >= na * (nc + 4 + cd), "head_grad_gather: bad shape");
    HeadLevels lv;
    lv.nlev = nlev;
    for (int l = 0; l <= nlev; ++l) { lv.row[l] = lev_row[l]; lv.anchor[l] = lev_anchor[l]; }
    for (int l = 0; l < nlev; ++l)
        YM_REQUIRE((lev_row[l + 1] - lev_row[l]) % B == 0 && (lev_anchor[l + 1] - lev_anchor[l]) == (lev_row[l + 1] - lev_row[l]) / B * na,
                   "head_grad_gather: level %d rows / anchors inconsistent", l);
    const long long total = (long long)lev_row[nlev] * pitch;
    hipLaunchKernelGGL(k_head_grad_gather, dim3(ew_grid((size_t)total)), dim3(256), 0, (hipStream_t)s, dclass, dbox, dcoef, coef, B, N,
                       nc, cd, na, lv, pitch, g_scale, dz);
    return ym_check_launch("head_grad_gather");
}

extern "C" int ym_scatter3(const float* src, float* d0, int n0, float* d1, int n1, float* d2, int n2, int accumulate, ym_stream_t s) {
    YM_REQUIRE(src && n0 >= 0 && n1 >= 0 && n2 >= 0 && (n0 == 0 || d0) && (n1 == 0 || d1) && (n2 == 0 || d2), "scatter3: bad args");
    const int n = n0 + n1 + n2;
    if (n == 0) return YM_OK;
    hipLaunchKernelGGL(k_scatter3, dim3(ym_cdiv(n, 256)), dim3(256), 0, (hipStream_t)s, src, d0, n0, d1, n1, d2, n2, accumulate);
    return ym_check_launch("scatter3");
}
