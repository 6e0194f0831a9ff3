// The ResNet stem of the inference path as ONE launch for gfx950:
//   conv 7x7 stride 2 pad 3 (3 -> 64) + folded BatchNorm + ReLU + max-pool 3x3 stride 2 pad 1
//   (reference: modules/resnet.py:86-91 `conv1 -> bn1 -> relu -> maxpool`, eval mode).
//
// As two launches (ym_conv2d_fwd in stem mode + ym_maxpool3x3s2_fwd) the 272x272x64 conv output makes a round trip through HBM
// (18.9 MB written, read again by the pooling kernel) and the 128x64-tile conv ran at 0.16 of the f32 MFMA peak at batch 1.  Here
// a workgroup owns a 5x8 tile of POOLED pixels x all 64 channels:
//   * the 27x39-pixel input patch its 11x17 conv pixels see (stride 2, 7 taps) is read once, straight from the NCHW image
//     (three coalesced plane reads per pixel -> one 16-byte LDS slot [r, g, b, 0]: the NHWC4 conversion launch is gone too),
//   * the whole packed filter (64 x 224 floats, `ym_pack_conv_weight` with cin_pad 4) sits in LDS at a 228-float pitch
//     (16 rows of a ds_read_b128 phase fall into 16 distinct bank groups),
//   * implicit GEMM on v_mfma_f32_32x32x2_f32: M = 192 (187 conv pixels, six 32-row tiles), N = 64, K = 224; a filter tap is one
//     16-byte LDS slot, so the A fragment of K group g is ONE ds_read_b128 at patch[(2*cr + kh) * 39 + 2*cc + kw]; the four waves
//     take (n tile, three m tiles) each -- 12 MFMAs per 4 LDS reads, no barrier inside the K loop,
//   * epilogue: fma(acc, scale, shift), ReLU, the conv tile staged in LDS (over the dead filter image), 3x3/2 max over it with the
//     pooling kernel's own rules (window clipped to the image, NaN propagates), 16-byte stores of the pooled tile.
// Same products in the same order as conv_igemm_f32<128, 64, 1> (lane half h, step s of group g: k = 8g + 4h + s) and the same
// epilogue arithmetic -> the same bits as the two-launch path (tests/test_gpu_ops.py::test_fused_stem_equals_conv_then_maxpool).
#pragma clang fp contract(off)
#include "ym_common.h"

namespace {

constexpr int PH = 5, PW = 8;                         // pooled tile of a workgroup
constexpr int CR = 2 * PH + 1, CC = 2 * PW + 1;       // conv pixels it needs: 11 x 17
constexpr int MPIX = CR * CC;                         // 187
constexpr int MT = (MPIX + 31) / 32;                  // 6 m tiles
constexpr int IR = 2 * CR + 5, IC = 2 * CC + 5;       // input patch: 27 x 39 pixels of 16 bytes
constexpr int KP = 224, WP = 228;                     // packed filter row (floats), its pitch in LDS
constexpr int OP = 68;                                // pitch of the staged conv tile [MT * 32][64]
constexpr int PATCH_F = IR * IC * 4, W_F = 64 * WP;
static_assert(MT == 6, "three m tiles per wave pair");
static_assert(MT * 32 * OP <= W_F, "the staged conv tile aliases the filter image");
constexpr size_t STEM_LDS = (size_t)(PATCH_F + W_F) * sizeof(float);

__global__ __launch_bounds__(256) void k_stem_pool(const float* __restrict__ img, const float* __restrict__ wpk,
                                                   const float* __restrict__ scale, const float* __restrict__ shift,
                                                   float* __restrict__ out, int H, int W, int Ho, int Wo, int Hp, int Wp,
                                                   int tiles_w, int tiles_hw) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* patch = smem;
    float* wl = smem + PATCH_F;
    const int tid = threadIdx.x;
    const int b = blockIdx.x / tiles_hw, trem = blockIdx.x - b * tiles_hw;
    const int th = trem / tiles_w, tw = trem - th * tiles_w;
    const int ph0 = th * PH, pw0 = tw * PW;
    const int cr0 = 2 * ph0 - 1, cc0 = 2 * pw0 - 1;          // conv coordinates of the region's origin (may be -1)
    const int iy0 = 2 * cr0 - 3, ix0 = 2 * cc0 - 3;          // input coordinates of the patch's origin
    const size_t plane = (size_t)H * W;
    const float* src = img + (size_t)b * 3 * plane;
    for (int e = tid; e < IR * IC; e += 256) {
        const int r = e / IC, c = e - r * IC;
        const int iy = iy0 + r, ix = ix0 + c;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};                      // zero padding of the convolution
        if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) {
            const float* q = src + (size_t)iy * W + ix;
            v[0] = q[0]; v[1] = q[plane]; v[2] = q[2 * plane];
        }
        *reinterpret_cast<f32x4*>(patch + e * 4) = v;
    }
    for (int e = tid; e < 64 * (KP / 4); e += 256) {
        const int row = e / (KP / 4), c4 = e - row * (KP / 4);
        *reinterpret_cast<f32x4*>(wl + row * WP + c4 * 4) = *reinterpret_cast<const f32x4*>(wpk + (size_t)row * KP + c4 * 4);
    }
    __syncthreads();

    const int wave = tid >> 6, lane = tid & 63, i = lane & 31, h = lane >> 5;
    const int nt = wave & 1, mh = wave >> 1;
    int abase[3];
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        int p = (mh + 2 * u) * 32 + i;
        p = p < MPIX ? p : MPIX - 1;                        // rows past the region: computed, staged, never pooled
        const int cr = p / CC, cc = p - cr * CC;
        abase[u] = (2 * cr * IC + 2 * cc) * 4;
    }
    const float* bfrag = wl + (nt * 32 + i) * WP + h * 4;
    f32x16 acc[3];
#pragma unroll
    for (int u = 0; u < 3; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[u][r] = 0.f;
#pragma unroll
    for (int t = 0; t < KP / 32; ++t) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            // lane half h multiplies filter tap 8t + 2g + h (four channels = one 16-byte slot); taps 49..55 are the zero padding
            // of K (their filter entries are 0; the A side is forced to 0 as well so that 0 * inf cannot appear)
            constexpr int T0 = 0;
            const int tap0 = 8 * t + 2 * g + T0, tap1 = tap0 + 1;
            const int c0 = tap0 < 49 ? tap0 : 48, c1 = tap1 < 49 ? tap1 : 48;
            const int off0 = ((c0 / 7) * IC + (c0 % 7)) * 4, off1 = ((c1 / 7) * IC + (c1 % 7)) * 4;
            const int off = h ? off1 : off0;
            const bool dead = h ? tap1 >= 49 : tap0 >= 49;
            const f32x4 fb = *reinterpret_cast<const f32x4*>(bfrag + t * 32 + g * 8);
            f32x4 fa[3];
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                fa[u] = *reinterpret_cast<const f32x4*>(patch + abase[u] + off);
                if (tap1 >= 49 && dead) fa[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int u = 0; u < 3; ++u) acc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[u][s], fb[s], acc[u], 0, 0, 0);
        }
    }
    __syncthreads();                                         // every wave is done with the filter image: the conv tile takes its place
    float* tile = wl;
    {
        const int ch = nt * 32 + i;
        const float sc = scale[ch], sh = shift[ch];
#pragma unroll
        for (int u = 0; u < 3; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int prow = (mh + 2 * u) * 32 + 8 * (r >> 2) + 4 * h + (r & 3);
                float v = __builtin_fmaf(acc[u][r], sc, sh);
                v = v < 0.f ? 0.f : v;                       // ReLU; NaN stays NaN
                tile[prow * OP + ch] = v;
            }
    }
    __syncthreads();
    for (int e = tid; e < PH * PW * 16; e += 256) {
        const int c4 = e & 15, pp = e >> 4;
        const int pr = pp / PW, pc = pp - pr * PW;
        const int ph = ph0 + pr, pw = pw0 + pc;
        if (ph >= Hp || pw >= Wp) continue;
        f32x4 m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const int gy = ph * 2 - 1 + dy;
            if ((unsigned)gy >= (unsigned)Ho) continue;
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int gx = pw * 2 - 1 + dx;
                if ((unsigned)gx >= (unsigned)Wo) continue;
                const f32x4 v = *reinterpret_cast<const f32x4*>(tile + ((2 * pr + dy) * CC + 2 * pc + dx) * OP + c4 * 4);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const bool take = v[q] > m[q] || v[q] != v[q];
                    m[q] = take ? v[q] : m[q];
                }
            }
        }
        *reinterpret_cast<f32x4*>(out + (((size_t)b * Hp + ph) * Wp + pw) * 64 + c4 * 4) = m;
    }
}

}  // namespace

extern "C" int ym_stem_conv_bn_relu_maxpool(const float* img_nchw, const float* w_packed, const float* scale, const float* shift,
                                            float* out, int B, int H, int W, int k_pad, ym_stream_t s) {
    YM_REQUIRE(img_nchw && w_packed && scale && shift && out && B > 0 && H > 0 && W > 0, "stem: bad arguments");
    YM_REQUIRE(k_pad == KP, "stem: the filter must be packed with cin_pad 4 and k_pad %d (7x7 taps x 4 channels, padded), got %d", KP, k_pad);
    YM_REQUIRE((((uintptr_t)w_packed | (uintptr_t)out) & 15) == 0, "stem: filter / output must be 16-byte aligned");
    const int Ho = (H + 6 - 7) / 2 + 1, Wo = (W + 6 - 7) / 2 + 1;
    YM_REQUIRE(Ho > 0 && Wo > 0, "stem: image too small");
    const int Hp = (Ho + 2 - 3) / 2 + 1, Wp = (Wo + 2 - 3) / 2 + 1;
    const int tiles_h = ym_cdiv(Hp, PH), tiles_w = ym_cdiv(Wp, PW);
    const long long grid = (long long)B * tiles_h * tiles_w;
    YM_REQUIRE(grid < (1ll << 31), "stem: grid too large");
    static YmLdsAttr attr = {};
    if (int rc = ym_ensure_dyn_lds(attr, reinterpret_cast<const void*>(k_stem_pool), STEM_LDS, "stem_conv_bn_relu_maxpool")) return rc;
    hipLaunchKernelGGL(k_stem_pool, dim3((unsigned)grid), dim3(256), STEM_LDS, (hipStream_t)s, img_nchw, w_packed, scale, shift, out,
                       H, W, Ho, Wo, Hp, Wp, tiles_w, tiles_h * tiles_w);
    return ym_check_launch("stem_conv_bn_relu_maxpool");
}
