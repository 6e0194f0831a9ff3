// Backward kernels of the Swin-T backbone (SURVEY.md §8 row a18 under loss.backward(), reference train.py:126 with
// cfg swin_tiny_coco) and the AdamW step the reference selects for it (train.py:62-63).  The Linear layers reuse the conv
// data / weight gradient kernels (a Linear is a 1x1 convolution of the NHWC token tensor).
//   LayerNorm backward (modules/swin_transformer.py:225-228,245,273,310,321 are the forward call sites): one wave per row,
//     statistics recomputed from x; per-workgroup partial dgamma/dbeta + an ordered second pass (deterministic).
//   Patch-merge LayerNorm backward: same, with the 2x2 gather of the forward turned into a scatter of dx.
//   GELU (exact erf form, :92-96) forward / backward on the saved pre-activation.
//   Window attention backward (:172-200 + pad / roll / partition of :249-283): one 64-lane workgroup per (window, head),
//     lane = token; the 49x49 score matrix is rebuilt in registers row by row, dV / dK reduce over queries through LDS.  Attention
//     is 1.5 % of the network's FLOPs, so this is plain FMA code: no MFMA tiling.
#include "ym_common.h"

namespace {

// ---- LayerNorm backward ----------------------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(256) void k_layernorm_bwd(const float* __restrict__ dy, const float* __restrict__ x,
                                                        const float* __restrict__ gamma, float eps, float* __restrict__ dx,
                                                        float* __restrict__ part /*[grid][2][C]*/, long long M, int C, int B, int H,
                                                        int W, int Csrc) {
    __shared__ float s_red[4][2][1536];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const long long wave0 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long long nw = ((long long)gridDim.x * blockDim.x) >> 6;
    const int C4 = C >> 2;
    f32x4 ag[6], ab[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) { ag[i] = f32x4{0.f, 0.f, 0.f, 0.f}; ab[i] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    for (long long m = wave0; m < M; m += nw) {
        f32x4 v[6], g[6];
        size_t src[6];
        bool live[6];
        float sum = 0.f;
        int Wo = 0, b = 0, oy = 0, ox = 0;
        if (MODE == 1) {
            const int Ho = (H + 1) / 2;
            Wo = (W + 1) / 2;
            b = (int)(m / ((long long)Ho * Wo));
            const int rem = (int)(m - (long long)b * Ho * Wo);
            oy = rem / Wo; ox = rem - oy * Wo;
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int c4 = lane + 64 * i;
            v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            live[i] = false;
            src[i] = 0;
            if (c4 < C4) {
                if (MODE == 0) {
                    src[i] = (size_t)m * C + c4 * 4;
                    live[i] = true;
                } else {
                    const int q = (c4 * 4) / Csrc, cc = c4 * 4 - q * Csrc;
                    const int iy = 2 * oy + (q & 1), ix = 2 * ox + (q >> 1);
                    live[i] = iy < H && ix < W;
                    src[i] = (((size_t)b * H + iy) * W + ix) * Csrc + cc;
                }
                if (live[i]) v[i] = *reinterpret_cast<const f32x4*>(x + src[i]);
                sum += v[i][0] + v[i][1] + v[i][2] + v[i][3];
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
        const float mean = sum / (float)C;
        float sq = 0.f;
#pragma unroll
        for (int i = 0; i < 6; ++i)
            if (lane + 64 * i < C4) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float d = v[i][e] - mean; sq += d * d; }
            }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
        const float rstd = 1.f / sqrtf(sq / (float)C + eps);
        float c1 = 0.f, c2 = 0.f;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int c4 = lane + 64 * i;
            g[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (c4 < C4) {
                const f32x4 d = *reinterpret_cast<const f32x4*>(dy + (size_t)m * C + c4 * 4);
                const f32x4 gm = *reinterpret_cast<const f32x4*>(gamma + c4 * 4);
                const f32x4 xh = (v[i] - mean) * rstd;
                v[i] = xh;
                g[i] = d * gm;                               // d loss / d xhat
                ag[i] += d * xh;
                ab[i] += d;
#pragma unroll
                for (int e = 0; e < 4; ++e) { c1 += g[i][e]; c2 += g[i][e] * xh[e]; }
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { c1 += __shfl_xor(c1, o); c2 += __shfl_xor(c2, o); }
        c1 /= (float)C; c2 /= (float)C;
#pragma unroll
        for (int i = 0; i < 6; ++i)
            if (lane + 64 * i < C4 && live[i])
                *reinterpret_cast<f32x4*>(dx + src[i]) = (g[i] - c1 - v[i] * c2) * rstd;
    }
    // per-workgroup partial dgamma / dbeta
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int c4 = lane + 64 * i;
        if (c4 < C4) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { s_red[wv][0][c4 * 4 + e] = ag[i][e]; s_red[wv][1][c4 * 4 + e] = ab[i][e]; }
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        part[((size_t)blockIdx.x * 2 + 0) * C + c] = (s_red[0][0][c] + s_red[1][0][c]) + (s_red[2][0][c] + s_red[3][0][c]);
        part[((size_t)blockIdx.x * 2 + 1) * C + c] = (s_red[0][1][c] + s_red[1][1][c]) + (s_red[2][1][c] + s_red[3][1][c]);
    }
}

__global__ void k_ln_param_grad(const float* __restrict__ part, int blocks, int C, float* __restrict__ dgamma, float* __restrict__ dbeta) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double sg = 0.0, sb = 0.0;
    for (int b = 0; b < blocks; ++b) { sg += part[((size_t)b * 2) * C + c]; sb += part[((size_t)b * 2 + 1) * C + c]; }
    dgamma[c] = (float)sg;
    dbeta[c] = (float)sb;
}

// ---- GELU ------------------------------------------------------------------------------------------------------------------
__global__ void k_gelu_fwd(const float* __restrict__ z, float* __restrict__ out, long long n4) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        f32x4 v = *reinterpret_cast<const f32x4*>(z + i * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = 0.5f * v[e] * (1.f + erff(v[e] * 0.70710678118654752440f));
        *reinterpret_cast<f32x4*>(out + i * 4) = v;
    }
}

__global__ void k_gelu_bwd(const float* __restrict__ dy, const float* __restrict__ z, float* __restrict__ dz, long long n4) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(z + i * 4), d = *reinterpret_cast<const f32x4*>(dy + i * 4);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float cdf = 0.5f * (1.f + erff(v[e] * 0.70710678118654752440f));
            const float pdf = 0.39894228040143267794f * expf(-0.5f * v[e] * v[e]);
            o[e] = d[e] * (cdf + v[e] * pdf);
        }
        *reinterpret_cast<f32x4*>(dz + i * 4) = o;
    }
}

// ---- window attention backward ------------------------------------------------------------------------------------------
constexpr int WS = 7, NTOK = 49, HD = 32, LP = HD + 1, PP = NTOK /* odd pitch: lane-private rows are conflict-free */, NREL = (2 * WS - 1) * (2 * WS - 1);

struct AttnBwdP {
    const float* qkv;        // [B*H*W][3C]
    const float* qkv_bias;   // [3C]
    const float* table;      // [169][heads]
    const float* dout;       // [B*H*W][C]
    float* dqkv;             // [B*H*W][3C]
    float* dbias_pad;        // [3C]  gradient reaching the qkv bias through the padded tokens (accumulated)
    float* dtable;           // [169][heads] (accumulated)
    int B, H, W, C, heads, shift, Hp, Wp, nWh, nWw, nblk;
    float scale;
};

// grid = heads * nblk workgroups of 64 lanes; workgroup (head, k) walks windows k, k + nblk, ... of its head, so the
// relative-position-bias gradient of the head is accumulated in LDS and flushed once.
__global__ __launch_bounds__(64) void k_window_attention_bwd(const AttnBwdP p) {
    __shared__ float s_q[NTOK][LP], s_k[NTOK][LP], s_v[NTOK][LP], s_do[NTOK][LP];
    __shared__ float s_p[NTOK][PP], s_ds[NTOK][PP];
    __shared__ float s_bias[NREL], s_dbias[NREL];
    __shared__ int s_tok[64], s_reg[64];
    const int lane = threadIdx.x;
    const int head = blockIdx.x % p.heads, k0 = blockIdx.x / p.heads;
    const int C3 = 3 * p.C;
    const long long nwin = (long long)p.B * p.nWh * p.nWw;
    for (int i = lane; i < NREL; i += 64) { s_bias[i] = p.table[i * p.heads + head]; s_dbias[i] = 0.f; }
    for (long long win = k0; win < nwin; win += p.nblk) {
        long long t = win;
        const int wx = (int)(t % p.nWw); t /= p.nWw;
        const int wy = (int)(t % p.nWh);
        const int b = (int)(t / p.nWh);
        __syncthreads();
        int tok = -2, reg = 0;
        if (lane < NTOK) {
            const int iy = lane / WS, ix = lane - iy * WS;
            const int py = wy * WS + iy, px = wx * WS + ix;
            int oy = py + p.shift, ox = px + p.shift;
            if (oy >= p.Hp) oy -= p.Hp;
            if (ox >= p.Wp) ox -= p.Wp;
            tok = (oy < p.H && ox < p.W) ? (b * p.H + oy) * p.W + ox : -1;
            if (p.shift > 0) {
                const int hr = py < p.Hp - WS ? 0 : (py < p.Hp - p.shift ? 1 : 2);
                const int wr = px < p.Wp - WS ? 0 : (px < p.Wp - p.shift ? 1 : 2);
                reg = hr * 3 + wr;
            }
        }
        s_tok[lane] = tok;
        s_reg[lane] = reg;
        float q[HD], dO[HD];
        if (lane < NTOK) {
            const float* src = tok >= 0 ? p.qkv + (size_t)tok * C3 : p.qkv_bias;
#pragma unroll
            for (int d4 = 0; d4 < HD / 4; ++d4) {
                const f32x4 qv = *reinterpret_cast<const f32x4*>(src + head * HD + d4 * 4);
                const f32x4 kv = *reinterpret_cast<const f32x4*>(src + p.C + head * HD + d4 * 4);
                const f32x4 vv = *reinterpret_cast<const f32x4*>(src + 2 * p.C + head * HD + d4 * 4);
                f32x4 dv = {0.f, 0.f, 0.f, 0.f};
                if (tok >= 0) dv = *reinterpret_cast<const f32x4*>(p.dout + (size_t)tok * p.C + head * HD + d4 * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    q[d4 * 4 + e] = qv[e] * p.scale;
                    dO[d4 * 4 + e] = dv[e];
                    s_q[lane][d4 * 4 + e] = qv[e] * p.scale;
                    s_k[lane][d4 * 4 + e] = kv[e];
                    s_v[lane][d4 * 4 + e] = vv[e];
                    s_do[lane][d4 * 4 + e] = dv[e];
                }
            }
        }
        __syncthreads();
        if (lane < NTOK) {
            // row `lane` of S = q_s K^T + bias + mask, P = softmax(S), dP = dO V^T, dS = P o (dP - rowsum(P o dP))
            const int qiy = lane / WS, qix = lane - qiy * WS;
            float* pr = s_p[lane];                       // this lane's rows live in LDS (lane-private until the column phase)
            float* dsr = s_ds[lane];
            float mx = -INFINITY;
            for (int j = 0; j < NTOK; ++j) {
                float sc = 0.f;
#pragma unroll
                for (int d = 0; d < HD; ++d) sc += q[d] * s_k[j][d];
                const int kiy = j / WS, kix = j - kiy * WS;
                sc += s_bias[(qiy - kiy + WS - 1) * (2 * WS - 1) + (qix - kix + WS - 1)];
                if (p.shift > 0 && s_reg[j] != reg) sc += -100.f;
                pr[j] = sc;
                mx = fmaxf(mx, sc);
            }
            float sum = 0.f;
            for (int j = 0; j < NTOK; ++j) { const float e = expf(pr[j] - mx); pr[j] = e; sum += e; }
            const float inv = 1.f / sum;
            float rowsum = 0.f;
            for (int j = 0; j < NTOK; ++j) {
                const float pj = pr[j] * inv;
                pr[j] = pj;
                float a = 0.f;
#pragma unroll
                for (int d = 0; d < HD; ++d) a += dO[d] * s_v[j][d];
                dsr[j] = a;
                rowsum += pj * a;
            }
            float dq[HD];
#pragma unroll
            for (int d = 0; d < HD; ++d) dq[d] = 0.f;
            for (int j = 0; j < NTOK; ++j) {
                const float ds = pr[j] * (dsr[j] - rowsum);
                dsr[j] = ds;
                const int kiy = j / WS, kix = j - kiy * WS;
                atomicAdd(&s_dbias[(qiy - kiy + WS - 1) * (2 * WS - 1) + (qix - kix + WS - 1)], ds);   // distinct per lane for a given j
#pragma unroll
                for (int d = 0; d < HD; ++d) dq[d] += ds * s_k[j][d];
            }
            // dq = scale * dq_s
            if (tok >= 0) {
                float* dst = p.dqkv + (size_t)tok * C3 + head * HD;
#pragma unroll
                for (int d4 = 0; d4 < HD / 4; ++d4)
                    *reinterpret_cast<f32x4*>(dst + d4 * 4) = f32x4{dq[d4 * 4] * p.scale, dq[d4 * 4 + 1] * p.scale, dq[d4 * 4 + 2] * p.scale, dq[d4 * 4 + 3] * p.scale};
            } else {
#pragma unroll
                for (int d = 0; d < HD; ++d) atomicAdd(p.dbias_pad + head * HD + d, dq[d] * p.scale);
            }
        }
        __syncthreads();
        if (lane < NTOK) {
            // column `lane`: dK[j] = sum_i dS[i][j] q_s[i], dV[j] = sum_i P[i][j] dO[i]
            float dk[HD], dv[HD];
#pragma unroll
            for (int d = 0; d < HD; ++d) { dk[d] = 0.f; dv[d] = 0.f; }
#pragma unroll 7
            for (int i = 0; i < NTOK; ++i) {
                const float ds = s_ds[i][lane], pp = s_p[i][lane];
#pragma unroll
                for (int d = 0; d < HD; ++d) { dk[d] += ds * s_q[i][d]; dv[d] += pp * s_do[i][d]; }
            }
            if (tok >= 0) {
                float* dst = p.dqkv + (size_t)tok * C3 + head * HD;
#pragma unroll
                for (int d4 = 0; d4 < HD / 4; ++d4) {
                    *reinterpret_cast<f32x4*>(dst + p.C + d4 * 4) = f32x4{dk[d4 * 4], dk[d4 * 4 + 1], dk[d4 * 4 + 2], dk[d4 * 4 + 3]};
                    *reinterpret_cast<f32x4*>(dst + 2 * p.C + d4 * 4) = f32x4{dv[d4 * 4], dv[d4 * 4 + 1], dv[d4 * 4 + 2], dv[d4 * 4 + 3]};
                }
            } else {
#pragma unroll
                for (int d = 0; d < HD; ++d) {
                    atomicAdd(p.dbias_pad + p.C + head * HD + d, dk[d]);
                    atomicAdd(p.dbias_pad + 2 * p.C + head * HD + d, dv[d]);
                }
            }
        }
    }
    __syncthreads();
    for (int i = lane; i < NREL; i += 64)
        if (s_dbias[i] != 0.f) atomicAdd(p.dtable + i * p.heads + head, s_dbias[i]);
}

// ---- AdamW (torch.optim.AdamW single-tensor update order) --------------------------------------------------------------
__global__ void k_adamw(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, long long n,
                        float decay, float beta1, float beta2, float eps, float step_size, float bc2_sqrt) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float gr = g[i];
        const float w = p[i] * decay;                                   // param.mul_(1 - lr * weight_decay)
        const float mi = m[i] + (gr - m[i]) * (1.f - beta1);            // exp_avg.lerp_(grad, 1 - beta1)
        const float vi = v[i] * beta2 + (1.f - beta2) * gr * gr;
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] = w - step_size * (mi / denom);
    }
}

int ln_bwd_blocks(long long M) {
    long long b = (M + 3) / 4;
    return (int)(b > 256 ? 256 : b);
}

}  // namespace

extern "C" size_t ym_layernorm_bwd_workspace_bytes(int C) { return (size_t)256 * 2 * C * sizeof(float); }

extern "C" int ym_layernorm_bwd(const float* dy, const float* x, const float* gamma, float eps, int64_t M, int C, float* dx,
                                float* dgamma, float* dbeta, void* workspace, size_t workspace_bytes, ym_stream_t s) {
    YM_REQUIRE(dy && x && gamma && dx && dgamma && dbeta && workspace, "layernorm_bwd: null pointer");
    YM_REQUIRE(M > 0 && C > 0 && C % 4 == 0 && C <= 1536, "layernorm_bwd: C must be a multiple of 4, <= 1536");
    if (workspace_bytes < ym_layernorm_bwd_workspace_bytes(C)) { ym_set_error("layernorm_bwd: workspace too small"); return YM_ENOSPC; }
    const int blocks = ln_bwd_blocks(M);
    hipLaunchKernelGGL(k_layernorm_bwd<0>, dim3(blocks), dim3(256), 0, (hipStream_t)s, dy, x, gamma, eps, dx, (float*)workspace,
                       (long long)M, C, 0, 0, 0, 0);
    hipLaunchKernelGGL(k_ln_param_grad, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)s, (const float*)workspace, blocks, C, dgamma, dbeta);
    return ym_check_launch("layernorm_bwd");
}

extern "C" int ym_patch_merge_layernorm_bwd(const float* dy, const float* x, int B, int H, int W, int C, const float* gamma, float eps,
                                            float* dx, float* dgamma, float* dbeta, void* workspace, size_t workspace_bytes,
                                            ym_stream_t s) {
    YM_REQUIRE(dy && x && gamma && dx && dgamma && dbeta && workspace, "patch_merge_layernorm_bwd: null pointer");
    YM_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0 && 4 * C <= 1536, "patch_merge_layernorm_bwd: C %% 4 == 0, 4C <= 1536");
    if (workspace_bytes < ym_layernorm_bwd_workspace_bytes(4 * C)) { ym_set_error("patch_merge_layernorm_bwd: workspace too small"); return YM_ENOSPC; }
    const long long M = (long long)B * ((H + 1) / 2) * ((W + 1) / 2);
    const int blocks = ln_bwd_blocks(M);
    hipLaunchKernelGGL(k_layernorm_bwd<1>, dim3(blocks), dim3(256), 0, (hipStream_t)s, dy, x, gamma, eps, dx, (float*)workspace, M, 4 * C,
                       B, H, W, C);
    hipLaunchKernelGGL(k_ln_param_grad, dim3((4 * C + 255) / 256), dim3(256), 0, (hipStream_t)s, (const float*)workspace, blocks, 4 * C,
                       dgamma, dbeta);
    return ym_check_launch("patch_merge_layernorm_bwd");
}

extern "C" int ym_gelu_fwd(const float* z, float* out, int64_t n, ym_stream_t s) {
    YM_REQUIRE(z && out && n > 0 && n % 4 == 0, "gelu_fwd: n must be a positive multiple of 4");
    long long blocks = (n / 4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_gelu_fwd, dim3((int)blocks), dim3(256), 0, (hipStream_t)s, z, out, (long long)(n / 4));
    return ym_check_launch("gelu_fwd");
}

extern "C" int ym_gelu_bwd(const float* dy, const float* z, float* dz, int64_t n, ym_stream_t s) {
    YM_REQUIRE(dy && z && dz && n > 0 && n % 4 == 0, "gelu_bwd: n must be a positive multiple of 4");
    long long blocks = (n / 4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_gelu_bwd, dim3((int)blocks), dim3(256), 0, (hipStream_t)s, dy, z, dz, (long long)(n / 4));
    return ym_check_launch("gelu_bwd");
}

extern "C" int ym_swin_window_attention_bwd(const float* qkv, const float* qkv_bias, const float* rel_bias_table, const float* dout,
                                            int B, int H, int W, int C, int heads, int window, int shift, float* dqkv,
                                            float* dqkv_bias_pad, float* dtable, ym_stream_t s) {
    YM_REQUIRE(qkv && qkv_bias && rel_bias_table && dout && dqkv && dqkv_bias_pad && dtable, "window_attention_bwd: null pointer");
    YM_REQUIRE(window == WS && heads > 0 && C == heads * HD && shift >= 0 && shift < WS,
               "window_attention_bwd: window must be 7 and head dim 32 (C = %d, heads = %d)", C, heads);
    AttnBwdP p;
    p.qkv = qkv; p.qkv_bias = qkv_bias; p.table = rel_bias_table; p.dout = dout; p.dqkv = dqkv; p.dbias_pad = dqkv_bias_pad;
    p.dtable = dtable;
    p.B = B; p.H = H; p.W = W; p.C = C; p.heads = heads; p.shift = shift;
    p.nWh = (H + WS - 1) / WS; p.nWw = (W + WS - 1) / WS;
    p.Hp = p.nWh * WS; p.Wp = p.nWw * WS;
    p.scale = 1.0f / sqrtf((float)HD);
    const long long nwin = (long long)B * p.nWh * p.nWw;
    long long nblk = (2048 + heads - 1) / heads;              // ~8 workgroups per CU
    if (nblk > nwin) nblk = nwin;
    p.nblk = (int)nblk;
    hipLaunchKernelGGL(k_window_attention_bwd, dim3((int)(nblk * heads)), dim3(64), 0, (hipStream_t)s, p);
    return ym_check_launch("window_attention_bwd");
}

extern "C" int ym_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1,
                             float beta2, float eps, float weight_decay, int step, ym_stream_t s) {
    YM_REQUIRE(param && grad && exp_avg && exp_avg_sq && n > 0 && step >= 1, "adamw_step: bad args");
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    long long blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_adamw, dim3((int)blocks), dim3(256), 0, (hipStream_t)s, param, grad, exp_avg, exp_avg_sq, (long long)n,
                       (float)(1.0 - (double)lr * (double)weight_decay), beta1, beta2, eps, (float)((double)lr / bc1), (float)sqrt(bc2));
    return ym_check_launch("adamw_step");
}
