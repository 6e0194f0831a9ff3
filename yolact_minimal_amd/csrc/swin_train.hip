// Backward kernels of the Swin-T backbone (SURVEY.md §8 row a18 under loss.backward(), reference train.py:126 with
// cfg swin_tiny_coco) and the AdamW step the reference selects for it (train.py:62-63).  The Linear layers reuse the conv
// data / weight gradient kernels (a Linear is a 1x1 convolution of the NHWC token tensor).
//   LayerNorm backward (modules/swin_transformer.py:225-228,245,273,310,321 are the forward call sites): one wave per row,
//     statistics recomputed from x; per-workgroup partial dgamma/dbeta + an ordered second pass (deterministic).
//   Patch-merge LayerNorm backward: same, with the 2x2 gather of the forward turned into a scatter of dx.
//   GELU (exact erf form, :92-96) forward / backward on the saved pre-activation.
//   Window attention backward (:172-200 + pad / roll / partition of :249-283): one wave per (window, head) on the f32 MFMA, the
//     49x49 matrices held in registers in both orientations (see k_window_attention_bwd).
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

// Rows of <= 128 floats (stage 1, C = 96: 74 % of the tokens): a HALF wave per row, U rows of each half in flight (x and dy of all of
// them requested first).  One row per wave left 40 of the 64 lanes idle there and two dependent round trips per row.  Same arithmetic
// per row as k_layernorm_bwd<0> (its xor-32 butterfly step adds zeros for such rows); the per-workgroup dgamma / dbeta partials sum
// the rows in another order (rounding-level difference in those two vectors).
template <int U, int LPR>
__global__ __launch_bounds__(256) void k_layernorm_bwd_small(const float* __restrict__ dy, const float* __restrict__ x,
                                                              const float* __restrict__ gamma, float eps, float* __restrict__ dx,
                                                              float* __restrict__ part /*[grid][2][C]*/, long long M, int C) {
    constexpr int RPW = 64 / LPR, SLOTS = 4 * RPW;                      // LPR lanes per row (32: C <= 128, 64: C <= 256)
    __shared__ float s_red[SLOTS][2][4 * LPR];
    const int lane = threadIdx.x & 63, half = lane / LPR, l32 = lane % LPR, hwv = (threadIdx.x >> 6) * RPW + half;
    const long long hw0 = (((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6) * RPW + half;
    const long long nhw = (((long long)gridDim.x * blockDim.x) >> 6) * RPW;
    const int C4 = C >> 2;
    const bool on = l32 < C4;
    f32x4 gm = {0.f, 0.f, 0.f, 0.f}, ag = {0.f, 0.f, 0.f, 0.f}, ab = {0.f, 0.f, 0.f, 0.f};
    if (on) gm = *reinterpret_cast<const f32x4*>(gamma + l32 * 4);
    for (long long m0 = hw0; m0 < M; m0 += nhw * U) {
        f32x4 v[U], d[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long long m = m0 + u * nhw;
            v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            d[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (on && m < M) {
                v[u] = *reinterpret_cast<const f32x4*>(x + (size_t)m * C + l32 * 4);
                d[u] = *reinterpret_cast<const f32x4*>(dy + (size_t)m * C + l32 * 4);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long long m = m0 + u * nhw;
            float sum = 0.f;
            if (on) sum += v[u][0] + v[u][1] + v[u][2] + v[u][3];
#pragma unroll
            for (int o = LPR / 2; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
            const float mean = sum / (float)C;
            float sq = 0.f;
            if (on) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float t = v[u][e] - mean; sq += t * t; }
            }
#pragma unroll
            for (int o = LPR / 2; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
            const float rstd = 1.f / sqrtf(sq / (float)C + eps);
            float c1 = 0.f, c2 = 0.f;
            f32x4 xh = {0.f, 0.f, 0.f, 0.f}, g = {0.f, 0.f, 0.f, 0.f};
            if (on && m < M) {
                xh = (v[u] - mean) * rstd;
                g = d[u] * gm;
                ag += d[u] * xh;
                ab += d[u];
#pragma unroll
                for (int e = 0; e < 4; ++e) { c1 += g[e]; c2 += g[e] * xh[e]; }
            }
#pragma unroll
            for (int o = LPR / 2; o > 0; o >>= 1) { c1 += __shfl_xor(c1, o); c2 += __shfl_xor(c2, o); }
            c1 /= (float)C; c2 /= (float)C;
            if (on && m < M) *reinterpret_cast<f32x4*>(dx + (size_t)m * C + l32 * 4) = (g - c1 - xh * c2) * rstd;
        }
    }
    if (on) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { s_red[hwv][0][l32 * 4 + e] = ag[e]; s_red[hwv][1][l32 * 4 + e] = ab[e]; }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float a = 0.f, b = 0.f;
#pragma unroll
        for (int r = 0; r < SLOTS; ++r) { a += s_red[r][0][c]; b += s_red[r][1][c]; }
        part[((size_t)blockIdx.x * 2 + 0) * C + c] = a;
        part[((size_t)blockIdx.x * 2 + 1) * C + c] = b;
    }
}

// dgamma / dbeta = ordered sum of the per-workgroup partials [blocks][2][C]: 16 channels x 16 block slices per workgroup, slices
// combined through LDS in a fixed order (deterministic).  (One thread per channel walking all the blocks in a dependent chain took
// 68 us per call whatever the size: 2.1 ms of a Swin-T training step.)
__global__ __launch_bounds__(256) void k_ln_param_grad(const float* __restrict__ part, int blocks, int C, float* __restrict__ dgamma,
                                                        float* __restrict__ dbeta) {
    __shared__ double red[2][16][17];
    const int cl = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl;
    double sg = 0.0, sb = 0.0;
    if (c < C) {
        int b = sl;
        for (; b + 48 < blocks; b += 64) {
            float g[4], bb[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                g[u] = part[((size_t)(b + 16 * u) * 2) * C + c];
                bb[u] = part[((size_t)(b + 16 * u) * 2 + 1) * C + c];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) { sg += g[u]; sb += bb[u]; }
        }
        for (; b < blocks; b += 16) { sg += part[((size_t)b * 2) * C + c]; sb += part[((size_t)b * 2 + 1) * C + c]; }
    }
    red[0][sl][cl] = sg;
    red[1][sl][cl] = sb;
    __syncthreads();
    if (sl == 0 && c < C) {
        for (int r = 1; r < 16; ++r) { sg += red[0][r][cl]; sb += red[1][r][cl]; }
        dgamma[c] = (float)sg;
        dbeta[c] = (float)sb;
    }
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

// ---- DropPath + residual (modules/swin_transformer.py:71-82 and the `x = shortcut + self.drop_path(x)` call sites :285,288) ------
// out = res + (y / keep) * floor(keep + rand[b]) in the reference's operation order, one pass (the reference graph is div, mul, add
// = three passes forward and two backward); rand [B] is the raw torch.rand draw, so the random stream is the reference's.
__global__ __launch_bounds__(256) void k_drop_path_add(const float* __restrict__ res, const float* __restrict__ y,
                                                        const float* __restrict__ rnd, float keep, float* __restrict__ out,
                                                        long long per4, long long total4) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long long)gridDim.x * 256) {
        const float m = floorf(keep + rnd[i / per4]);
        const f32x4 v = *reinterpret_cast<const f32x4*>(y + i * 4), r = *reinterpret_cast<const f32x4*>(res + i * 4);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = r[e] + (v[e] / keep) * m;
        *reinterpret_cast<f32x4*>(out + i * 4) = o;
    }
}

// dy = (dout * mask) / keep (autograd's order for y.div(keep) * mask); the residual's gradient is dout itself
__global__ __launch_bounds__(256) void k_drop_path_bwd(const float* __restrict__ dout, const float* __restrict__ rnd, float keep,
                                                        float* __restrict__ dy, long long per4, long long total4) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long long)gridDim.x * 256) {
        const float m = floorf(keep + rnd[i / per4]);
        const f32x4 d = *reinterpret_cast<const f32x4*>(dout + i * 4);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (d[e] * m) / keep;
        *reinterpret_cast<f32x4*>(dy + i * 4) = o;
    }
}

// ---- window attention backward ------------------------------------------------------------------------------------------
constexpr int WS = 7, NTOK = 49, HD = 32, NREL = (2 * WS - 1) * (2 * WS - 1);

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

// One WAVE per (window, head) on the f32 MFMA (v_mfma_f32_32x32x2_f32), 49 tokens padded to 64.  Workgroup = 4 waves of ONE head:
// workgroup (head, k) walks windows 4k + wave, 4(k + nblk) + wave, ... so the head's relative-position-bias gradient is summed in
// LDS and flushed once.  A 64x64 matrix of the wave lives in registers in one of two layouts:
//   T: X^T[key][q] — key = kt*32 + (r&3) + 8*(r>>2) + 4h across the 16 accumulator registers r and the lane half h, q = qt*32 +
//      (lane&31) across lanes (the D layout of MFMA(A = key-major operand, B = query-major operand));
//   N: X[q][key]  — the same with the roles of the operands swapped.
// A register of layout T is exactly the A-operand element (i = q, k = key) of one MFMA step of (X . B[key][d]); layout N serves
// (X^T . B[q][d]).  So no matrix is ever transposed through LDS; both layouts are simply computed (7 x 64 MFMAs per unit):
//   phase T:  S^T = K Q_s^T -> P^T (softmax over keys = over registers + one cross-half shuffle, as in the forward kernel);
//             dP^T = V dO^T;  rowsum[q] = sum_key P^T dP^T;  dS^T = P^T o (dP^T - rowsum);  dbias[rel(q,key)] += dS;
//             dQ = scale * dS K          (A = dS^T registers)
//   phase N:  S = Q_s K^T -> P with the row max / 1/sum of phase T (per-wave LDS vectors);  dP = dO V^T;  dS likewise;
//             dV = P^T dO, dK = dS^T Q_s (A = P / dS registers).
// The scalar-FMA predecessor of this kernel (one lane per token, operands broadcast from LDS) took 633 us per launch against
// 54 us for the forward kernel: 15.6 % of a Swin-T training step.
// The wave's four operand matrices (Q_s = scale * q, K, V, dO), each [64 tokens][32] floats at pitch MP in LDS, staged once per
// unit with coalesced loads (tokens >= 49 stay zero; padded tokens carry the qkv bias, dO = 0 there).  Every fragment below is an
// LDS read: fetched straight from global memory (16 bytes per lane at a 3C-float stride, each tile ~8 times per unit) the kernel
// ran at 351 us per launch, bound by L1/L2 traffic, for ~32 us of MFMA work.
constexpr int MP = 36, MAT = 64 * MP;
enum { M_Q = 0, M_K = 1, M_V = 2, M_DO = 3 };

// operand fragments of ONE 32-token tile: 4 float4 (d = 8g + 4h ..) of token tl*32 + (lane & 31)
__device__ __forceinline__ void attn_frag(const float* m, int tl, int row, int h, f32x4 (&f)[4]) {
#pragma unroll
    for (int g = 0; g < 4; ++g) f[g] = *reinterpret_cast<const f32x4*>(m + (tl * 32 + row) * MP + g * 8 + h * 4);
}

// B operand of the second-stage products: element (k = token f(r) + 4h of tile t, j = d = lane & 31)
__device__ __forceinline__ void attn_rows(const float* m, int row, int h, float (&v)[2][16]) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) v[t][r] = m[(t * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * MP + row];
}

// D[i = token f(r) + 4h of tile t][j = d = lane & 31] -> row of dqkv (or the padded tokens' share of the qkv-bias gradient)
// `pad`: the workgroup's LDS accumulator [32] of this matrix for the padded tokens (they carry the qkv bias, so their gradient is
// a bias gradient).  All workgroups of a head used to add into the same 96 global floats with one atomic per (padded token, d):
// 2.6 M atomics on 288 addresses in a stage-1 launch (5.8 % of its tokens are padding), serialised in the L2 — most of the 350 us
// this kernel took with every operand already in LDS.
__device__ __forceinline__ void attn_store(const AttnBwdP& p, const f32x16& o, int t, int col, const int* tokrow, int row, int h,
                                           float mul, float* pad) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int tk = tokrow[t * 32 + (r & 3) + 8 * (r >> 2) + 4 * h];
        if (tk >= 0) p.dqkv[(size_t)tk * (3 * p.C) + col + row] = o[r] * mul;
        else if (tk == -1) atomicAdd(pad + row, o[r] * mul);
    }
}

// acc[t] (t = 0, 1) = (tile t of matrix a as the A operand) x (the tile bf as the B operand)
__device__ __forceinline__ void attn_mm(const float* a, int row, int h, const f32x4 (&bf)[4], f32x16 (&acc)[2]) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        f32x4 af[4];
        attn_frag(a, t, row, h, af);
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2)
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[g][s2], bf[g][s2], acc[t], 0, 0, 0);
    }
}

// SLOTS = units (window, head) a workgroup works on at a time, a PAIR of waves each: 4 = one 144 KB workgroup per CU (default), 2 =
// two 72 KB workgroups per CU whose staging / compute phases are not coupled by a common barrier (YM_ATTN_BWD_SLOTS=2).  Measured in
// the Swin-T bs=8 training step, round 6: 34.27 ms with 4 slots, 34.55-34.61 with 2 -- alone on the chip the staging round trip of a
// unit is exposed (MFMA-busy 0.24), in the step the weight-gradient stream runs in those holes and the decoupling buys nothing.
template <int SLOTS>
__global__ __launch_bounds__(128 * SLOTS, 2) void k_window_attention_bwd(const AttnBwdP p) {
    constexpr int NT = 128 * SLOTS;
    __shared__ float s_bias[176], s_dbias[176], s_dpad[3][32];
    __shared__ int s_tok[SLOTS][64], s_reg[SLOTS][64];
    __shared__ float s_mx[SLOTS][64], s_inv[SLOTS][64], s_rs[SLOTS][64];
    extern __shared__ __attribute__((aligned(16))) float s_mat[];       // [SLOTS][4 matrices][64][MP]
    const int lane = threadIdx.x & 63, wv = (threadIdx.x >> 6) >> 1, half = (threadIdx.x >> 6) & 1;   // wv = unit slot
    const int head = blockIdx.x % p.heads, k0 = blockIdx.x / p.heads;
    const int C3 = 3 * p.C;
    const long long nwin = (long long)p.B * p.nWh * p.nWw;
    for (int i = threadIdx.x; i < NREL; i += NT) { s_bias[i] = p.table[i * p.heads + head]; s_dbias[i] = 0.f; }
    for (int i = threadIdx.x; i < SLOTS * 4 * MAT; i += NT) s_mat[i] = 0.f;     // (rows 49..63 of every tile stay zero)
    if (threadIdx.x < 96) s_dpad[threadIdx.x >> 5][threadIdx.x & 31] = 0.f;
    __syncthreads();
    // relative-position-bias gradient: element (key, q) of dS^T always lives in the same (register, lane) of a phase-T wave, so the
    // sum over the workgroup's units is kept in registers and scattered to the 169 bins ONCE (it used to be 2401 LDS atomics per
    // unit on 169 addresses)
    f32x16 dbacc[2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) dbacc[kt][r] = 0.f;
    float* const mq = s_mat + wv * 4 * MAT;
    const float *const mk = mq + M_K * MAT, *const mv = mq + M_V * MAT, *const mdo = mq + M_DO * MAT;
    const int row = lane & 31, h = lane >> 5;
    const int* tokrow = s_tok[wv];
    const int qc = head * HD, kc = p.C + head * HD, vc = 2 * p.C + head * HD;
    // every wave runs the same number of iterations (the pair of a unit meets at workgroup barriers): a slot past the last window idles
    for (long long base = (long long)k0 * SLOTS; base < nwin; base += (long long)p.nblk * SLOTS) {
        const long long win = base + wv;
        const bool valid = win < nwin;
        long long t = valid ? win : 0;
        const int wx = (int)(t % p.nWw); t /= p.nWw;
        const int wy = (int)(t % p.nWh);
        const int b = (int)(t / p.nWh);
        {
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
            // (both waves of the pair write the same values; the previous unit ended on a workgroup barrier)
            s_tok[wv][lane] = tok;
            s_reg[wv][lane] = reg;
        }
        __syncthreads();                      // token table visible to both waves of the unit
        // stage Q_s, K, V, dO of the 49 tokens: 8 lanes x 16 bytes per token row, 8 tokens per pass, passes split between the pair
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int j = (it * 2 + half) * 8 + (lane >> 3), c = (lane & 7) * 4;
            if (valid && j < NTOK) {
                const int tk = tokrow[j];
                const float* src = tk >= 0 ? p.qkv + (size_t)tk * C3 : p.qkv_bias;
                f32x4 dv = {0.f, 0.f, 0.f, 0.f};
                if (tk >= 0) dv = *reinterpret_cast<const f32x4*>(p.dout + (size_t)tk * p.C + qc + c);
                *reinterpret_cast<f32x4*>(mq + M_Q * MAT + j * MP + c) = *reinterpret_cast<const f32x4*>(src + qc + c) * p.scale;
                *reinterpret_cast<f32x4*>(mq + M_K * MAT + j * MP + c) = *reinterpret_cast<const f32x4*>(src + kc + c);
                *reinterpret_cast<f32x4*>(mq + M_V * MAT + j * MP + c) = *reinterpret_cast<const f32x4*>(src + vc + c);
                *reinterpret_cast<f32x4*>(mq + M_DO * MAT + j * MP + c) = dv;
            }
        }
        __syncthreads();

        // ------------------------------------------------- phase T: [key registers][query lanes]; this wave: query tile `half`
        if (valid) {
            const int qt = half;
            const int q = qt * 32 + row;
            const int qiy = q / WS, qix = q - qiy * WS;
            f32x16 st[2], dpt[2];             // [kt]
            {
                f32x4 qf[4];
                attn_frag(mq, qt, row, h, qf);
                attn_mm(mk, row, h, qf, st);                 // S^T = K Q_s^T
            }
            // bias + mask + softmax over keys of this query column: the forward kernel's sequence (swin_ops.hip k_window_attention)
            {
                const int qreg = s_reg[wv][q];
                float mx = -INFINITY;
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                        float v = -INFINITY;
                        if (key < NTOK && q < NTOK) {
                            const int kiy = key / WS, kix = key - kiy * WS;
                            v = st[kt][r] + s_bias[(qiy - kiy + WS - 1) * (2 * WS - 1) + (qix - kix + WS - 1)];
                            if (p.shift > 0 && s_reg[wv][key] != qreg) v += -100.f;
                        } else if (key < NTOK) {
                            v = 0.f;
                        }
                        st[kt][r] = v;
                        mx = fmaxf(mx, v);
                    }
                mx = fmaxf(mx, __shfl_xor(mx, 32));
                float sum = 0.f;
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float e = expf(st[kt][r] - mx);
                        st[kt][r] = e;
                        sum += e;
                    }
                sum += __shfl_xor(sum, 32);
                const float inv = 1.f / sum;
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) st[kt][r] *= inv;
                s_mx[wv][q] = mx;                       // (both lane halves hold the same q: same value twice)
                s_inv[wv][q] = inv;
            }
            {
                f32x4 dof[4];
                attn_frag(mdo, qt, row, h, dof);
                attn_mm(mv, row, h, dof, dpt);               // dP^T = V dO^T
            }
            float rs = 0.f;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) rs += st[kt][r] * dpt[kt][r];
            rs += __shfl_xor(rs, 32);
            s_rs[wv][q] = rs;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    const bool live = key < NTOK && q < NTOK;
                    const float ds = live ? st[kt][r] * (dpt[kt][r] - rs) : 0.f;
                    dpt[kt][r] = ds;
                    dbacc[kt][r] += ds;
                }
            {
                float kk[2][16];
                attn_rows(mk, row, h, kk);
                f32x16 o;
#pragma unroll
                for (int r = 0; r < 16; ++r) o[r] = 0.f;
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o = __builtin_amdgcn_mfma_f32_32x32x2f32(dpt[kt][r], kk[kt][r], o, 0, 0, 0);
                attn_store(p, o, qt, qc, tokrow, row, h, p.scale, s_dpad[0]);                               // dQ = scale * dS K
            }
        }
        __syncthreads();                      // s_mx / s_inv / s_rs of both query tiles are there

        // ------------------------------------------------- phase N: [query registers][key lanes]; this wave: key tile `half`
        if (valid) {
            const int kt = half;
            const int key = kt * 32 + row;
            const int kiy = key / WS, kix = key - kiy * WS;
            const int kreg = s_reg[wv][key];
            f32x16 sn[2], dpn[2];             // [qt]
            {
                f32x4 kf[4];
                attn_frag(mk, kt, row, h, kf);
                attn_mm(mq, row, h, kf, sn);                 // S = Q_s K^T
            }
            {
                f32x4 vf[4];
                attn_frag(mv, kt, row, h, vf);
                attn_mm(mdo, row, h, vf, dpn);               // dP = dO V^T
            }
#pragma unroll
            for (int qt = 0; qt < 2; ++qt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int q = qt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    float pv = 0.f, ds = 0.f;
                    if (key < NTOK && q < NTOK) {
                        const int qiy = q / WS, qix = q - qiy * WS;
                        float v = sn[qt][r] + s_bias[(qiy - kiy + WS - 1) * (2 * WS - 1) + (qix - kix + WS - 1)];
                        if (p.shift > 0 && s_reg[wv][q] != kreg) v += -100.f;
                        pv = expf(v - s_mx[wv][q]) * s_inv[wv][q];
                        ds = pv * (dpn[qt][r] - s_rs[wv][q]);
                    }
                    sn[qt][r] = pv;
                    dpn[qt][r] = ds;
                }
            {
                float dd[2][16];
                attn_rows(mdo, row, h, dd);
                f32x16 o;
#pragma unroll
                for (int r = 0; r < 16; ++r) o[r] = 0.f;
#pragma unroll
                for (int qt = 0; qt < 2; ++qt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o = __builtin_amdgcn_mfma_f32_32x32x2f32(sn[qt][r], dd[qt][r], o, 0, 0, 0);
                attn_store(p, o, kt, vc, tokrow, row, h, 1.f, s_dpad[2]);                                    // dV = P^T dO
            }
            {
                float qq[2][16];
                attn_rows(mq, row, h, qq);
                f32x16 o;
#pragma unroll
                for (int r = 0; r < 16; ++r) o[r] = 0.f;
#pragma unroll
                for (int qt = 0; qt < 2; ++qt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o = __builtin_amdgcn_mfma_f32_32x32x2f32(dpn[qt][r], qq[qt][r], o, 0, 0, 0);
                attn_store(p, o, kt, kc, tokrow, row, h, 1.f, s_dpad[1]);                                    // dK = dS^T Q_s
            }
        }
        __syncthreads();                      // both waves are done with the unit's tiles before the next one is staged
    }
    {
        const int row = lane & 31, h = lane >> 5, q = half * 32 + row;          // this wave's phase-T coordinates (query tile `half`)
        const int qiy = q / WS, qix = q - qiy * WS;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (key < NTOK && q < NTOK) {
                    const int kiy = key / WS, kix = key - kiy * WS;
                    atomicAdd(&s_dbias[(qiy - kiy + WS - 1) * (2 * WS - 1) + (qix - kix + WS - 1)], dbacc[kt][r]);
                }
            }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < NREL; i += NT)
        if (s_dbias[i] != 0.f) atomicAdd(p.dtable + i * p.heads + head, s_dbias[i]);
    if (threadIdx.x < 96) {
        const float v = s_dpad[threadIdx.x >> 5][threadIdx.x & 31];
        if (v != 0.f) atomicAdd(p.dbias_pad + (threadIdx.x >> 5) * p.C + head * HD + (threadIdx.x & 31), v);
    }
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

constexpr int LN_BWD_MAX_BLOCKS = 2048;      // (256 left one workgroup per CU: 0.5 TB/s on the 170 MB of a stage-1 LayerNorm)
int ln_bwd_blocks(long long M) {
    long long b = (M + 15) / 16;                 // >= 4 rows per wave
    if (b < 1) b = 1;
    return (int)(b > LN_BWD_MAX_BLOCKS ? LN_BWD_MAX_BLOCKS : b);
}

}  // namespace

extern "C" size_t ym_layernorm_bwd_workspace_bytes(int C) { return (size_t)LN_BWD_MAX_BLOCKS * 2 * C * sizeof(float); }

extern "C" int ym_layernorm_bwd(const float* dy, const float* x, const float* gamma, float eps, int64_t M, int C, float* dx,
                                float* dgamma, float* dbeta, void* workspace, size_t workspace_bytes, ym_stream_t s) {
    YM_REQUIRE(dy && x && gamma && dx && dgamma && dbeta && workspace, "layernorm_bwd: null pointer");
    YM_REQUIRE(M > 0 && C > 0 && C % 4 == 0 && C <= 1536, "layernorm_bwd: C must be a multiple of 4, <= 1536");
    if (workspace_bytes < ym_layernorm_bwd_workspace_bytes(C)) { ym_set_error("layernorm_bwd: workspace too small"); return YM_ENOSPC; }
    const int blocks = ln_bwd_blocks(M);
    if (C <= 128)
        hipLaunchKernelGGL((k_layernorm_bwd_small<2, 32>), dim3(blocks), dim3(256), 0, (hipStream_t)s, dy, x, gamma, eps, dx, (float*)workspace,
                           (long long)M, C);
    else if (C <= 256)
        hipLaunchKernelGGL((k_layernorm_bwd_small<2, 64>), dim3(blocks), dim3(256), 0, (hipStream_t)s, dy, x, gamma, eps, dx, (float*)workspace,
                           (long long)M, C);
    else
        hipLaunchKernelGGL(k_layernorm_bwd<0>, dim3(blocks), dim3(256), 0, (hipStream_t)s, dy, x, gamma, eps, dx, (float*)workspace,
                           (long long)M, C, 0, 0, 0, 0);
    hipLaunchKernelGGL(k_ln_param_grad, dim3((C + 15) / 16), dim3(256), 0, (hipStream_t)s, (const float*)workspace, blocks, C, dgamma, dbeta);
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
    hipLaunchKernelGGL(k_ln_param_grad, dim3((4 * C + 15) / 16), dim3(256), 0, (hipStream_t)s, (const float*)workspace, blocks, 4 * C,
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

extern "C" int ym_drop_path_add(const float* res, const float* y, const float* rnd, float keep, float* out, int B, int64_t per_sample,
                                ym_stream_t s) {
    YM_REQUIRE(res && y && rnd && out && B > 0 && per_sample > 0 && per_sample % 4 == 0 && keep > 0.f, "drop_path_add: bad args");
    const long long total4 = (long long)B * (per_sample / 4);
    long long blocks = (total4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_drop_path_add, dim3((int)blocks), dim3(256), 0, (hipStream_t)s, res, y, rnd, keep, out, (long long)(per_sample / 4), total4);
    return ym_check_launch("drop_path_add");
}

extern "C" int ym_drop_path_bwd(const float* dout, const float* rnd, float keep, float* dy, int B, int64_t per_sample, ym_stream_t s) {
    YM_REQUIRE(dout && rnd && dy && B > 0 && per_sample > 0 && per_sample % 4 == 0 && keep > 0.f, "drop_path_bwd: bad args");
    const long long total4 = (long long)B * (per_sample / 4);
    long long blocks = (total4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_drop_path_bwd, dim3((int)blocks), dim3(256), 0, (hipStream_t)s, dout, rnd, keep, dy, (long long)(per_sample / 4), total4);
    return ym_check_launch("drop_path_bwd");
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
    static int slots = 0;                                      // YM_ATTN_BWD_SLOTS=2: two half-size workgroups per CU (A/B; see the kernel)
    if (slots == 0) { const char* e = getenv("YM_ATTN_BWD_SLOTS"); slots = (e && atoi(e) == 2) ? 2 : 4; }
    const long long wgs_needed = (nwin + slots - 1) / slots;   // one unit (window, head) per pair of waves, `slots` units per workgroup
    long long nblk = (256 * (4 / slots)) / heads;              // workgroups resident at once (their operand tiles fill the LDS): the grid
    if (nblk < 1) nblk = 1;                                    // must not exceed them, or its last few workgroups run as a second round
    if (nblk > wgs_needed) nblk = wgs_needed;
    p.nblk = (int)nblk;
    const size_t lds = (size_t)slots * 4 * MAT * sizeof(float);      // 72 KB (2 slots: two workgroups per CU) / 144 KB (4 slots)
    static YmLdsAttr attr2 = {}, attr4 = {};
    if (slots == 4) {
        if (int rc = ym_ensure_dyn_lds(attr4, reinterpret_cast<const void*>(k_window_attention_bwd<4>), lds, "window_attention_bwd")) return rc;
        hipLaunchKernelGGL(k_window_attention_bwd<4>, dim3((int)(nblk * heads)), dim3(512), lds, (hipStream_t)s, p);
    } else {
        if (int rc = ym_ensure_dyn_lds(attr2, reinterpret_cast<const void*>(k_window_attention_bwd<2>), lds, "window_attention_bwd")) return rc;
        hipLaunchKernelGGL(k_window_attention_bwd<2>, dim3((int)(nblk * heads)), dim3(256), lds, (hipStream_t)s, p);
    }
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
