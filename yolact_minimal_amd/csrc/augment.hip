// train_aug pixel work on the device (reference utils/augmentations.py:60-77,138-216,230-252): ONE launch
// for the image, one for the masks.  The reference makes ~10 numpy / cv2 passes (photometric distortion at full resolution,
// mirror, crop, pad to square, resize, pad / crop to the train size, normalise, BGR->RGB, HWC->CHW); here every OUTPUT pixel walks
// that chain backwards to its (up to) four source texels, applies the photometric distortion to those texels only, blends them
// with ATen / OpenCV half-pixel bilinear weights and writes the normalised RGB planes.  The random decisions arrive as a plan
// (ym_aug_plan) drawn on the host in the reference's `random` call order.  HBM-bound: source image read ~once, output written once.
#include "ym_common.h"

namespace {

struct Texel { float b, g, r; };

__device__ __forceinline__ float clip255(float v) { return fminf(fmaxf(v, 0.f), 255.f); }

// photometric_distort (:60-77) of one BGR texel: brightness, contrast, BGR->HSV (OpenCV float32: H degrees, S 0..1, V 0..255),
// saturation, hue (+ wrap), HSV->BGR, clip
__device__ __forceinline__ Texel photometric(Texel t, const ym_aug_plan& p) {
    if (p.has_brightness) { t.b = clip255(t.b + p.brightness); t.g = clip255(t.g + p.brightness); t.r = clip255(t.r + p.brightness); }
    if (p.has_contrast) { t.b = clip255(t.b * p.contrast); t.g = clip255(t.g * p.contrast); t.r = clip255(t.r * p.contrast); }
    const float v = fmaxf(fmaxf(t.b, t.g), t.r), mn = fminf(fminf(t.b, t.g), t.r), diff = v - mn;
    float s = v > 0.f ? diff / v : 0.f;
    float h = 0.f;
    if (diff > 0.f) {
        if (v == t.r) h = (t.g - t.b) / diff;
        else if (v == t.g) h = 2.f + (t.b - t.r) / diff;
        else h = 4.f + (t.r - t.g) / diff;
        h *= 60.f;
        if (h < 0.f) h += 360.f;
    }
    s *= p.saturation;
    h += p.hue;
    if (h > 360.f) h -= 360.f;
    if (h < 0.f) h += 360.f;
    // HSV -> BGR
    float hh = h;
    if (hh < 0.f) hh += 360.f;
    if (hh >= 360.f) hh -= 360.f;
    hh /= 60.f;
    const float fi = floorf(hh), f = hh - fi;
    const int i = ((int)fi) % 6;
    const float pp = v * (1.f - s), q = v * (1.f - s * f), tt = v * (1.f - s * (1.f - f));
    float r, g, b;
    switch (i) {
        case 0: r = v; g = tt; b = pp; break;
        case 1: r = q; g = v; b = pp; break;
        case 2: r = pp; g = v; b = tt; break;
        case 3: r = pp; g = q; b = v; break;
        case 4: r = tt; g = pp; b = v; break;
        default: r = v; g = pp; b = q; break;
    }
    return Texel{clip255(b), clip255(g), clip255(r)};
}

// output pixel -> coordinates in the resized r x r image (false: it lies in the final padding)
__device__ __forceinline__ bool to_resized(const ym_aug_plan& p, int y, int x, int& ry, int& rx) {
    ry = y; rx = x;
    if (p.final_mode == 1) { ry = y - p.fy; rx = x - p.fx; }
    else if (p.final_mode == 2) { ry = y + p.fy; rx = x + p.fx; }
    return ry >= 0 && rx >= 0 && ry < p.r && rx < p.r;
}

// ATen upsample_bilinear2d(align_corners=False) source index (== cv2.INTER_LINEAR's half-pixel centres)
__device__ __forceinline__ void src_index(float scale, int dst, int in_size, int& i0, int& i1, float& l1) {
    float s = scale * ((float)dst + 0.5f) - 0.5f;
    if (s < 0.f) s = 0.f;
    i0 = (int)s;
    if (i0 > in_size - 1) i0 = in_size - 1;
    i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
    l1 = s - (float)i0;
}

// square texel -> source pixel (false: it lies in the pad-to-square border)
__device__ __forceinline__ bool to_source(const ym_aug_plan& p, int sy, int sx, int& Y, int& X) {
    const int oy = sy - p.py, ox = sx - p.px;
    if (oy < 0 || ox < 0 || oy >= p.ch || ox >= p.cw) return false;
    Y = p.cy + oy;
    X = p.cx + ox;
    if (p.mirror) X = p.W - 1 - X;
    return true;
}

template <typename T>
__global__ __launch_bounds__(256) void k_train_aug_image(const T* __restrict__ img, const ym_aug_plan p, float* __restrict__ out) {
    const int S = p.S;
    const long long total = (long long)S * S;
    const float scale = (float)p.q / (float)p.r;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const int y = (int)(e / S), x = (int)(e - (long long)y * S);
        float vb = p.mean[0], vg = p.mean[1], vr = p.mean[2];
        int ry, rx;
        if (to_resized(p, y, x, ry, rx)) {
            int y0, y1, x0, x1;
            float ly, lx;
            src_index(scale, ry, p.q, y0, y1, ly);
            src_index(scale, rx, p.q, x0, x1, lx);
            Texel t[2][2];
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    int Y, X;
                    if (to_source(p, a ? y1 : y0, b ? x1 : x0, Y, X)) {
                        const T* s = img + ((size_t)Y * p.W + X) * 3;
                        t[a][b] = photometric(Texel{(float)s[0], (float)s[1], (float)s[2]}, p);
                    } else {
                        t[a][b] = Texel{p.mean[0], p.mean[1], p.mean[2]};
                    }
                }
            const float hy = 1.f - ly, hx = 1.f - lx;
            vb = hy * (hx * t[0][0].b + lx * t[0][1].b) + ly * (hx * t[1][0].b + lx * t[1][1].b);
            vg = hy * (hx * t[0][0].g + lx * t[0][1].g) + ly * (hx * t[1][0].g + lx * t[1][1].g);
            vr = hy * (hx * t[0][0].r + lx * t[0][1].r) + ly * (hx * t[1][0].r + lx * t[1][1].r);
        }
        out[e] = (vr - p.mean[2]) / p.std[2];                    // RGB planes (normalize_and_toRGB :212-216)
        out[total + e] = (vg - p.mean[1]) / p.std[1];
        out[2 * total + e] = (vb - p.mean[0]) / p.std[0];
    }
}

// cv2.resize on uint8 (INTER_LINEAR) is fixed point: taps of one axis as OpenCV computes them (imgproc/resize.cpp) — source index
// from the half-pixel centre in double, weights saturate_cast<short>(w * 2048) (round half to even); along x the weights are forced
// to (1, 0) at the borders, along y the row indices are clipped instead.
__device__ __forceinline__ void u8_taps(int n_src, int n_dst, int d, bool clamp_weights, int& i0, int& i1, int& w0, int& w1) {
    const float f = (float)(((double)d + 0.5) * ((double)n_src / (double)n_dst) - 0.5);
    int i = (int)floorf(f);
    float fr = f - (float)i;
    if (clamp_weights) {
        if (i < 0) { fr = 0.f; i = 0; }
        if (i >= n_src - 1) { fr = 0.f; i = n_src - 1; }
        i0 = i;
        i1 = i + 1 < n_src ? i + 1 : n_src - 1;
    } else {
        i1 = i + 1 < 0 ? 0 : (i + 1 > n_src - 1 ? n_src - 1 : i + 1);
        i0 = i < 0 ? 0 : (i > n_src - 1 ? n_src - 1 : i);
    }
    w0 = (int)__builtin_rintf((1.f - fr) * 2048.f);
    w1 = (int)__builtin_rintf(fr * 2048.f);
}

// `fixed`: uint8 masks of an already-square crop — the reference resizes them as uint8 (pad_to_square :138-141 returns them
// untouched, multi_scale_resize :180-181), i.e. through OpenCV's 8-bit fixed-point path, and a {0,1} mask stays {0,1}
// (oracle/augment_ref.py resize_bilinear_u8); every other sample's masks went through the float32 pad buffer first.
template <typename T>
__global__ __launch_bounds__(256) void k_train_aug_masks(const T* __restrict__ masks, const int32_t* __restrict__ keep, int k,
                                                         const ym_aug_plan p, float* __restrict__ out, int fixed) {
    const int S = p.S;
    const long long plane = (long long)S * S, total = plane * k;
    const float scale = (float)p.q / (float)p.r;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const int j = (int)(e / plane);
        const long long rem = e - (long long)j * plane;
        const int y = (int)(rem / S), x = (int)(rem - (long long)y * S);
        const T* m = masks + (size_t)keep[j] * p.H * p.W;
        float v = 0.f;
        int ry, rx;
        if (fixed && to_resized(p, y, x, ry, rx)) {
            int y0, y1, x0, x1, a0, a1, b0, b1;
            u8_taps(p.q, p.r, rx, true, x0, x1, a0, a1);
            u8_taps(p.q, p.r, ry, false, y0, y1, b0, b1);
            int t[2][2];
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    int Y, X;
                    t[a][b] = to_source(p, a ? y1 : y0, b ? x1 : x0, Y, X) ? (int)m[(size_t)Y * p.W + X] : 0;
                }
            const int h0 = t[0][0] * a0 + t[0][1] * a1, h1 = t[1][0] * a0 + t[1][1] * a1;          // horizontal pass, scale 2048
            v = (float)((((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2);
        } else if (to_resized(p, y, x, ry, rx)) {
            int y0, y1, x0, x1;
            float ly, lx;
            src_index(scale, ry, p.q, y0, y1, ly);
            src_index(scale, rx, p.q, x0, x1, lx);
            float t[2][2];
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    int Y, X;
                    t[a][b] = to_source(p, a ? y1 : y0, b ? x1 : x0, Y, X) ? (float)m[(size_t)Y * p.W + X] : 0.f;
                }
            v = (1.f - ly) * ((1.f - lx) * t[0][0] + lx * t[0][1]) + ly * ((1.f - lx) * t[1][0] + lx * t[1][1]);
        }
        out[e] = v;
    }
}

int check_plan(const ym_aug_plan* p) {
    YM_REQUIRE(p, "train_aug: null plan");
    YM_REQUIRE(p->H > 0 && p->W > 0 && p->cw > 0 && p->ch > 0 && p->cx >= 0 && p->cy >= 0 && p->cx + p->cw <= p->W && p->cy + p->ch <= p->H,
               "train_aug: crop outside the image");
    YM_REQUIRE(p->q >= p->cw && p->q >= p->ch && p->px >= 0 && p->py >= 0 && p->px + p->cw <= p->q && p->py + p->ch <= p->q,
               "train_aug: pad-to-square inconsistent");
    YM_REQUIRE(p->r > 0 && p->S > 0 && p->final_mode >= 0 && p->final_mode <= 2, "train_aug: bad sizes");
    YM_REQUIRE(p->final_mode != 0 || p->r == p->S, "train_aug: final_mode 0 needs r == S");
    YM_REQUIRE(p->final_mode != 1 || (p->fx >= 0 && p->fy >= 0 && p->fx + p->r <= p->S && p->fy + p->r <= p->S), "train_aug: final pad");
    YM_REQUIRE(p->final_mode != 2 || (p->fx >= 0 && p->fy >= 0 && p->fx + p->S <= p->r && p->fy + p->S <= p->r), "train_aug: final crop");
    return YM_OK;
}

int grid_for(long long n) {
    long long g = (n + 255) / 256;
    return (int)(g > 8192 ? 8192 : g);
}

}  // namespace

extern "C" int ym_train_aug_image(const void* img_hwc_bgr, int is_u8, const ym_aug_plan* plan, float* out_chw, ym_stream_t s) {
    int rc = check_plan(plan);
    if (rc != YM_OK) return rc;
    YM_REQUIRE(img_hwc_bgr && out_chw, "train_aug_image: null pointer");
    const int g = grid_for((long long)plan->S * plan->S);
    if (is_u8) hipLaunchKernelGGL(k_train_aug_image<uint8_t>, dim3(g), dim3(256), 0, (hipStream_t)s, (const uint8_t*)img_hwc_bgr, *plan, out_chw);
    else hipLaunchKernelGGL(k_train_aug_image<float>, dim3(g), dim3(256), 0, (hipStream_t)s, (const float*)img_hwc_bgr, *plan, out_chw);
    return ym_check_launch("train_aug_image");
}

extern "C" int ym_train_aug_masks(const void* masks, int is_u8, const int32_t* keep, int k, const ym_aug_plan* plan, float* out,
                                  ym_stream_t s) {
    int rc = check_plan(plan);
    if (rc != YM_OK) return rc;
    YM_REQUIRE(k >= 0, "train_aug_masks: k < 0");
    if (k == 0) return YM_OK;
    YM_REQUIRE(masks && keep && out, "train_aug_masks: null pointer");
    const int g = grid_for((long long)plan->S * plan->S * k);
    const int fixed = (is_u8 && plan->cw == plan->ch) ? 1 : 0;          // already-square sample: the reference's masks stay uint8
    if (is_u8) hipLaunchKernelGGL(k_train_aug_masks<uint8_t>, dim3(g), dim3(256), 0, (hipStream_t)s, (const uint8_t*)masks, keep, k, *plan, out, fixed);
    else hipLaunchKernelGGL(k_train_aug_masks<float>, dim3(g), dim3(256), 0, (hipStream_t)s, (const float*)masks, keep, k, *plan, out, 0);
    return ym_check_launch("train_aug_masks");
}
