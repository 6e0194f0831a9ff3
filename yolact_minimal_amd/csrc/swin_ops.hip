// Swin-T building blocks for gfx950: LayerNorm, patch-merge gather + LayerNorm, and the
// shifted-window multi-head attention.  The Linear layers (qkv / proj / fc1 / fc2 / reduction) are 1x1 convolutions
// of the [tokens][C] = NHWC tensor and run on conv_igemm_f32 (GELU is one of its epilogue activations).
//
// Window attention (reference modules/swin_transformer.py:172-200 + the pad / roll / partition / reverse / un-roll
// of SwinTransformerBlock.forward :249-283): ONE WAVE per (window, head).  N = 49 tokens, head dim 32.
//   S^T = K * Q^T is computed instead of Q * K^T so that the softmax axis (keys) is lane-local: with the 32x32x2 f32
//   MFMA, D[i = key][j = query] puts query j in lane (j & 31) and 32 of its 64 keys in that lane's 2 x 16
//   accumulator registers; the other 32 keys sit in lane ^ 32 -> max / sum need one cross-half shuffle each.
//   The probabilities never leave registers: for P*V the MFMA wants A[i = query][k = key] with lane half h supplying
//   the second key of each K=2 step, and the accumulator layout already holds key (r&3)+8*(r>>2)+4h in register r,
//   so register r IS the A operand of step r (keys paired as (k, k+4)).  V is read as B[k = key][j = d].
//   Padding (tokens outside H x W carry q = k = v = bias because the reference pads AFTER norm1), the cyclic shift,
//   window partition and their inverses are address arithmetic; the relative-position bias of this head (169 floats),
//   per-key token offsets and shift-mask region ids are staged in a 1.3 KB per-wave LDS slice.
#include "ym_common.h"

namespace {

// ---- LayerNorm over the last dim; MODE 1 gathers the 2x2 patch-merge concat on the fly ---------------------------
// x: MODE 0 [M][C]; MODE 1 NHWC [B][H][W][C/4-per-source] gathered to rows of 4*Csrc (order (0,0),(1,0),(0,1),(1,1)).
template <int MODE>
__global__ __launch_bounds__(256) void k_layernorm(const float* __restrict__ x, const float* __restrict__ gamma,
                                                    const float* __restrict__ beta, float eps, float* __restrict__ out,
                                                    long long M, int C, int B, int H, int W, int Csrc) {
    const int lane = threadIdx.x & 63;
    const long long wave0 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long long nw = ((long long)gridDim.x * blockDim.x) >> 6;
    const int C4 = C >> 2;
    for (long long m = wave0; m < M; m += nw) {
        f32x4 v[6];
        float sum = 0.f;
        int Ho = 0, Wo = 0, b = 0, oy = 0, ox = 0;
        if (MODE == 1) {
            Ho = (H + 1) / 2; Wo = (W + 1) / 2;
            b = (int)(m / ((long long)Ho * Wo));
            const int rem = (int)(m - (long long)b * Ho * Wo);
            oy = rem / Wo; ox = rem - oy * Wo;
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int c4 = lane + 64 * i;
            v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (c4 < C4) {
                if (MODE == 0) {
                    v[i] = *reinterpret_cast<const f32x4*>(x + (size_t)m * C + c4 * 4);
                } else {
                    const int q = (c4 * 4) / Csrc, cc = c4 * 4 - q * Csrc;      // which of the 4 sources
                    const int iy = 2 * oy + (q & 1), ix = 2 * ox + (q >> 1);
                    if (iy < H && ix < W) v[i] = *reinterpret_cast<const f32x4*>(x + (((size_t)b * H + iy) * W + ix) * Csrc + cc);
                }
                sum += v[i][0] + v[i][1] + v[i][2] + v[i][3];
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
        const float mean = sum / (float)C;
        float sq = 0.f;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            if (lane + 64 * i < C4) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float d = v[i][e] - mean; sq += d * d; }
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
        const float rstd = 1.f / sqrtf(sq / (float)C + eps);
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int c4 = lane + 64 * i;
            if (c4 < C4) {
                const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + c4 * 4), bb = *reinterpret_cast<const f32x4*>(beta + c4 * 4);
                *reinterpret_cast<f32x4*>(out + (size_t)m * C + c4 * 4) = (v[i] - mean) * rstd * g + bb;
            }
        }
    }
}

// ---- shifted-window attention ---------------------------------------------------------------------------------------
constexpr int WS = 7, NTOK = 49, HD = 32;

struct AttnP {
    const float* qkv;     // [B*H*W][3C]
    const float* qkv_bias;  // [3C]
    const float* table;   // [(2*WS-1)^2][heads]
    float* out;           // [B*H*W][C]
    int B, H, W, C, heads, shift, Hp, Wp, nWh, nWw;
    float scale;
};

__global__ __launch_bounds__(256) void k_window_attention(const AttnP p) {
    __shared__ float s_bias[4][176];
    __shared__ int s_tok[4][64];     // token row index, -1 = padded token (bias only), -2 = beyond the 49 tokens
    __shared__ int s_reg[4][64];     // shift-mask region id
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const long long unit = (long long)blockIdx.x * 4 + wv;
    const long long nunits = (long long)p.B * p.nWh * p.nWw * p.heads;
    if (unit >= nunits) return;                       // wave-uniform
    const int head = (int)(unit % p.heads);
    long long t = unit / p.heads;
    const int wx = (int)(t % p.nWw); t /= p.nWw;
    const int wy = (int)(t % p.nWh);
    const int b = (int)(t / p.nWh);
    const int C3 = 3 * p.C;

    // per-token metadata (lane = token index)
    {
        int tok = -2, reg = 0;
        if (lane < NTOK) {
            const int iy = lane / WS, ix = lane - iy * WS;
            const int py = wy * WS + iy, px = wx * WS + ix;               // coordinates in the shifted, padded map
            int oy = py + p.shift, ox = px + p.shift;                      // roll(-shift): shifted[py] = x[(py+shift) % Hp]
            if (oy >= p.Hp) oy -= p.Hp;
            if (ox >= p.Wp) ox -= p.Wp;
            tok = (oy < p.H && ox < p.W) ? (b * p.H + oy) * p.W + ox : -1;
            if (p.shift > 0) {
                const int hr = py < p.Hp - WS ? 0 : (py < p.Hp - p.shift ? 1 : 2);
                const int wr = px < p.Wp - WS ? 0 : (px < p.Wp - p.shift ? 1 : 2);
                reg = hr * 3 + wr;
            }
        }
        s_tok[wv][lane] = tok;
        s_reg[wv][lane] = reg;
        for (int i = lane; i < (2 * WS - 1) * (2 * WS - 1); i += 64) s_bias[wv][i] = p.table[i * p.heads + head];
    }
    // same-wave LDS visibility: DS ops of one wave execute in order; make the compiler wait for the writes
    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0)
    __builtin_amdgcn_wave_barrier();

    const int row = lane & 31, h = lane >> 5;
    // operand fragments: 4 float4 (k = 8g + 4h ..) per 32-row tile
    f32x4 kf[2][4], qf[2][4];
#pragma unroll
    for (int tl = 0; tl < 2; ++tl) {
        const int tk = s_tok[wv][tl * 32 + row];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int dd = g * 8 + h * 4;
            f32x4 kv = {0.f, 0.f, 0.f, 0.f}, qv = {0.f, 0.f, 0.f, 0.f};
            if (tk >= 0) {
                kv = *reinterpret_cast<const f32x4*>(p.qkv + (size_t)tk * C3 + p.C + head * HD + dd);
                qv = *reinterpret_cast<const f32x4*>(p.qkv + (size_t)tk * C3 + head * HD + dd);
            } else if (tk == -1) {
                kv = *reinterpret_cast<const f32x4*>(p.qkv_bias + p.C + head * HD + dd);
                qv = *reinterpret_cast<const f32x4*>(p.qkv_bias + head * HD + dd);
            }
            kf[tl][g] = kv;
            qf[tl][g] = qv * p.scale;                                          // q = q * scale (:176)
        }
    }
    f32x16 st[2][2];   // [key tile][query tile]
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) st[kt][qt][r] = 0.f;
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int s = 0; s < 4; ++s)
                    st[kt][qt] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[kt][g][s], qf[qt][g][s], st[kt][qt], 0, 0, 0);
        }

    // bias + mask + softmax over keys, per query column (lane & 31 within a query tile)
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        const int q = qt * 32 + row;
        const int qiy = q / WS, qix = q - qiy * WS;
        const int qreg = s_reg[wv][q & 63];
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                float v = -INFINITY;
                if (key < NTOK && q < NTOK) {
                    const int kiy = key / WS, kix = key - kiy * WS;
                    v = st[kt][qt][r] + s_bias[wv][(qiy - kiy + WS - 1) * (2 * WS - 1) + (qix - kix + WS - 1)];
                    if (p.shift > 0 && s_reg[wv][key] != qreg) v += -100.f;
                } else if (key < NTOK) {
                    v = 0.f;                                                    // unused query column: keep finite
                }
                st[kt][qt][r] = v;
                mx = fmaxf(mx, v);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float e = expf(st[kt][qt][r] - mx);                        // exp(-inf) = 0 for keys >= 49
                st[kt][qt][r] = e;
                sum += e;
            }
        sum += __shfl_xor(sum, 32);
        const float inv = 1.f / sum;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) st[kt][qt][r] *= inv;
    }

    // V operand: B[k = key][j = d = lane & 31] for the key this lane half supplies in step (kt, r)
    float vv[2][16];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            const int tk = s_tok[wv][key];
            float v = 0.f;
            if (tk >= 0) v = p.qkv[(size_t)tk * C3 + 2 * p.C + head * HD + row];
            else if (tk == -1) v = p.qkv_bias[2 * p.C + head * HD + row];
            vv[kt][r] = v;
        }
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        f32x16 o;
#pragma unroll
        for (int r = 0; r < 16; ++r) o[r] = 0.f;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) o = __builtin_amdgcn_mfma_f32_32x32x2f32(st[kt][qt][r], vv[kt][r], o, 0, 0, 0);
        // D[i = query][j = d]: col = lane & 31 = d, row = (r&3) + 8*(r>>2) + 4h
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int q = qt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (q < NTOK) {
                const int tk = s_tok[wv][q];
                if (tk >= 0) p.out[(size_t)tk * p.C + head * HD + row] = o[r];   // window_reverse + un-roll + crop
            }
        }
    }
}

// Rows of <= 128 floats (Swin-T stage 1: C = 96, the stage with 74 % of the tokens): a row fills only 24 of a wave's 64 lanes, and one
// row per wave per iteration means one 16-byte load in flight per lane (2.3 TB/s on the 114 MB of a stage-1 LayerNorm at batch 8).
// Here a HALF wave owns a row and works on U rows at a time (2 U rows per wave in flight).  Bit-identical to k_layernorm<0>: there
// the upper lanes hold zeros, so its xor-32 butterfly step adds nothing and the remaining steps are the ones below.
// LPR = lanes per row: 32 (C <= 128, two rows per wave) or 64 (C <= 256: Swin-T stage 2, one row per wave, still U rows in flight).
template <int U, int LPR>
__global__ __launch_bounds__(256) void k_layernorm_small(const float* __restrict__ x, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float eps, float* __restrict__ out,
                                                          long long M, int C) {
    constexpr int RPW = 64 / LPR;                                       // rows per wave
    const int lane = threadIdx.x & 63, half = lane / LPR, l32 = lane % LPR;
    const long long hw0 = (((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6) * RPW + half;      // row-slot index
    const long long nhw = (((long long)gridDim.x * blockDim.x) >> 6) * RPW;
    const int C4 = C >> 2;
    const bool on = l32 < C4;
    f32x4 g = {0.f, 0.f, 0.f, 0.f}, bb = {0.f, 0.f, 0.f, 0.f};
    if (on) { g = *reinterpret_cast<const f32x4*>(gamma + l32 * 4); bb = *reinterpret_cast<const f32x4*>(beta + l32 * 4); }
    for (long long m0 = hw0; m0 < M; m0 += nhw * U) {
        f32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long long m = m0 + u * nhw;
            v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (on && m < M) v[u] = *reinterpret_cast<const f32x4*>(x + (size_t)m * C + l32 * 4);
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
                for (int e = 0; e < 4; ++e) { const float d = v[u][e] - mean; sq += d * d; }
            }
#pragma unroll
            for (int o = LPR / 2; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
            const float rstd = 1.f / sqrtf(sq / (float)C + eps);
            if (on && m < M) *reinterpret_cast<f32x4*>(out + (size_t)m * C + l32 * 4) = (v[u] - mean) * rstd * g + bb;
        }
    }
}

}  // namespace

extern "C" int ym_layernorm(const float* x, const float* gamma, const float* beta, float eps, float* out, int64_t M, int C,
                            ym_stream_t s) {
    YM_REQUIRE(x && gamma && beta && out && M > 0 && C > 0 && C % 4 == 0 && C <= 1536, "layernorm: C must be a multiple of 4, <= 1536");
    if (C <= 256) {
        constexpr int U = 4;
        const int rows_per_wg = C <= 128 ? 8 : 4;
        long long blocks = (M + rows_per_wg * U - 1) / (rows_per_wg * U);          // U rows per row slot and iteration
        if (blocks > 4096) blocks = 4096;
        if (C <= 128) hipLaunchKernelGGL((k_layernorm_small<U, 32>), dim3((int)blocks), dim3(256), 0, (hipStream_t)s, x, gamma, beta, eps, out, (long long)M, C);
        else hipLaunchKernelGGL((k_layernorm_small<U, 64>), dim3((int)blocks), dim3(256), 0, (hipStream_t)s, x, gamma, beta, eps, out, (long long)M, C);
        return ym_check_launch("layernorm");
    }
    long long blocks = (M + 3) / 4;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(k_layernorm<0>, dim3((int)blocks), dim3(256), 0, (hipStream_t)s, x, gamma, beta, eps, out, (long long)M, C,
                       0, 0, 0, 0);
    return ym_check_launch("layernorm");
}

extern "C" int ym_patch_merge_layernorm(const float* x, int B, int H, int W, int C, const float* gamma, const float* beta,
                                        float eps, float* out, ym_stream_t s) {
    YM_REQUIRE(x && gamma && beta && out && B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0 && 4 * C <= 1536,
               "patch_merge_layernorm: C must be a multiple of 4, 4C <= 1536");
    const long long M = (long long)B * ((H + 1) / 2) * ((W + 1) / 2);
    long long blocks = (M + 3) / 4;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(k_layernorm<1>, dim3((int)blocks), dim3(256), 0, (hipStream_t)s, x, gamma, beta, eps, out, M, 4 * C, B, H,
                       W, C);
    return ym_check_launch("patch_merge_layernorm");
}

extern "C" int ym_swin_window_attention(const float* qkv, const float* qkv_bias, const float* rel_bias_table, int B, int H, int W,
                                        int C, int heads, int window, int shift, float* out, ym_stream_t s) {
    YM_REQUIRE(qkv && qkv_bias && rel_bias_table && out, "window_attention: null pointer");
    YM_REQUIRE(window == WS && heads > 0 && C == heads * HD && shift >= 0 && shift < WS,
               "window_attention: window must be 7 and head dim 32 (C = %d, heads = %d)", C, heads);
    AttnP p;
    p.qkv = qkv; p.qkv_bias = qkv_bias; p.table = rel_bias_table; p.out = out;
    p.B = B; p.H = H; p.W = W; p.C = C; p.heads = heads; p.shift = shift;
    p.nWh = (H + WS - 1) / WS; p.nWw = (W + WS - 1) / WS;
    p.Hp = p.nWh * WS; p.Wp = p.nWw * WS;
    p.scale = 1.0f / sqrtf((float)HD);
    const long long units = (long long)B * p.nWh * p.nWw * heads;
    hipLaunchKernelGGL(k_window_attention, dim3((int)((units + 3) / 4)), dim3(256), 0, (hipStream_t)s, p);
    return ym_check_launch("window_attention");
}
