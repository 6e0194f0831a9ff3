// HBM-bound NHWC helpers: layout conversion, weight packing, BN folding, max-pool, bilinear x2, row softmax.
// All of these move each byte once; they are written for coalesced 16-byte accesses along C.
#include <stdarg.h>
#include "ym_common.h"

// ---- error plumbing (shared by every translation unit) ---------------------------------------------
static thread_local char g_err[512] = "";

void ym_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char* ym_last_error(void) { return g_err; }
extern "C" int ym_abi_version(void) { return 1; }

namespace {

__global__ void k_nchw_to_nhwc4(const float* __restrict__ in, float* __restrict__ out, int B, int C, int HW) {
    const size_t total = (size_t)B * HW;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t b = i / HW, px = i - b * HW;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        const float* src = in + b * C * HW + px;
        for (int c = 0; c < C; ++c) v[c] = src[(size_t)c * HW];
        *reinterpret_cast<f32x4*>(out + i * 4) = v;
    }
}

__global__ void k_pack_weight(const float* __restrict__ w, float* __restrict__ out, int Cout, int Cin, int KH, int KW,
                              int cin_pad, int k_pad) {
    const size_t total = (size_t)Cout * k_pad;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int n = (int)(i / k_pad), k = (int)(i - (size_t)n * k_pad);
        const int tap = k / cin_pad, c = k - tap * cin_pad;
        float v = 0.f;
        if (tap < KH * KW && c < Cin) {
            const int kh = tap / KW, kw = tap - kh * KW;
            v = w[(((size_t)n * Cin + c) * KH + kh) * KW + kw];
        }
        out[i] = v;
    }
}

__global__ void k_fold_bn(const float* gamma, const float* beta, const float* mean, const float* var, float eps,
                          float* scale, float* shift, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < C) {
        const float sc = gamma[c] / sqrtf(var[c] + eps);
        scale[c] = sc;
        shift[c] = beta[c] - mean[c] * sc;
    }
}

// idx (optional, training): per output element the window position 3*dy + dx of the element that won under ATen's rule
// (a later value replaces the running maximum if it is larger or NaN), which is what max_pool2d's backward routes the gradient to.
__global__ void k_maxpool3x3s2(const float* __restrict__ in, float* __restrict__ out, int B, int H, int W, int C4,
                               int Ho, int Wo, uint8_t* __restrict__ idx = nullptr) {
    const size_t total = (size_t)B * Ho * Wo * C4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4);
        size_t t = i / C4;
        const int ow = (int)(t % Wo); t /= Wo;
        const int oh = (int)(t % Ho);
        const int b = (int)(t / Ho);
        f32x4 m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        uint32_t arg = 0;                                         // four position bytes
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const int ih = oh * 2 - 1 + dy;
            if ((unsigned)ih >= (unsigned)H) continue;
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int iw = ow * 2 - 1 + dx;
                if ((unsigned)iw >= (unsigned)W) continue;
                const f32x4 v = *reinterpret_cast<const f32x4*>(in + (((size_t)b * H + ih) * W + iw) * C4 * 4 + c * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const bool take = v[e] > m[e] || v[e] != v[e];
                    m[e] = take ? v[e] : m[e];
                    if (take) arg = (arg & ~(0xFFu << (8 * e))) | ((uint32_t)(dy * 3 + dx) << (8 * e));
                }
            }
        }
        *reinterpret_cast<f32x4*>(out + i * 4) = m;
        if (idx) reinterpret_cast<uint32_t*>(idx)[i] = arg;
    }
}

// Source index maths follow ATen's upsample_bilinear2d (area_pixel_compute_source_index):
//   align_corners: src = dst * (in-1)/(out-1);  else: src = max(0, (dst+0.5)*in/out - 0.5)   (scale 0.5 for x2)
__device__ __forceinline__ void bil_coord(int dst, int in_sz, int out_sz, int align, int& i0, int& i1, float& l1) {
    float src;
    if (align) {
        const float sc = out_sz > 1 ? (float)(in_sz - 1) / (float)(out_sz - 1) : 0.f;
        src = sc * dst;
    } else {
        const float sc = (float)in_sz / (float)out_sz;
        src = sc * (dst + 0.5f) - 0.5f;
        if (src < 0.f) src = 0.f;
    }
    i0 = (int)src;
    if (i0 > in_sz - 1) i0 = in_sz - 1;
    i1 = i0 + ((i0 < in_sz - 1) ? 1 : 0);
    l1 = src - (float)i0;
}

__global__ void k_bilinear2x(const float* __restrict__ in, float* __restrict__ out, int B, int H, int W, int C4,
                             int align) {
    const int Ho = 2 * H, Wo = 2 * W;
    const size_t total = (size_t)B * Ho * Wo * C4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4);
        size_t t = i / C4;
        const int ow = (int)(t % Wo); t /= Wo;
        const int oh = (int)(t % Ho);
        const int b = (int)(t / Ho);
        int y0, y1, x0, x1; float ly, lx;
        bil_coord(oh, H, Ho, align, y0, y1, ly);
        bil_coord(ow, W, Wo, align, x0, x1, lx);
        const float hy = 1.f - ly, hx = 1.f - lx;
        const float* base = in + (size_t)b * H * W * C4 * 4 + c * 4;
        const f32x4 v00 = *reinterpret_cast<const f32x4*>(base + ((size_t)y0 * W + x0) * C4 * 4);
        const f32x4 v01 = *reinterpret_cast<const f32x4*>(base + ((size_t)y0 * W + x1) * C4 * 4);
        const f32x4 v10 = *reinterpret_cast<const f32x4*>(base + ((size_t)y1 * W + x0) * C4 * 4);
        const f32x4 v11 = *reinterpret_cast<const f32x4*>(base + ((size_t)y1 * W + x1) * C4 * 4);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = hy * (hx * v00[e] + lx * v01[e]) + ly * (hx * v10[e] + lx * v11[e]);
        *reinterpret_cast<f32x4*>(out + i * 4) = o;
    }
}

// One wave per row (C <= 128: two elements per lane), shuffle reductions.  rows*C*8 bytes of traffic.
__global__ __launch_bounds__(256) void k_softmax_rows(const float* __restrict__ in, float* __restrict__ out, long long rows, int C) {
    const int lane = threadIdx.x & 63;
    const long long wave0 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long long nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
    for (long long r = wave0; r < rows; r += nwaves) {
        const float* x = in + r * C;
        float v[4]; float mx = -INFINITY;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = lane + 64 * j;
            v[j] = c < C ? x[c] : -INFINITY;
            mx = fmaxf(mx, v[j]);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = lane + 64 * j;
            v[j] = c < C ? expf(v[j] - mx) : 0.f;
            sum += v[j];
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
        float* y = out + r * C;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = lane + 64 * j;
            if (c < C) y[c] = v[j] / sum;
        }
    }
}

inline int grid_for(size_t total, int block = 256, int cap = 8192) {
    size_t g = (total + block - 1) / block;
    if (g > (size_t)cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

}  // namespace

extern "C" int ym_nchw_to_nhwc4(const float* in, float* out, int B, int C, int H, int W, ym_stream_t s) {
    YM_REQUIRE(in && out && B > 0 && C >= 1 && C <= 4 && H > 0 && W > 0, "nchw_to_nhwc4: bad args");
    const size_t total = (size_t)B * H * W;
    hipLaunchKernelGGL(k_nchw_to_nhwc4, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)s, in, out, B, C, H * W);
    return ym_check_launch("nchw_to_nhwc4");
}

extern "C" int ym_pack_conv_weight(const float* w_oihw, float* w_packed, int Cout, int Cin, int KH, int KW,
                                   int cin_pad, int k_pad, ym_stream_t s) {
    YM_REQUIRE(w_oihw && w_packed && Cout > 0 && Cin > 0 && cin_pad >= Cin && k_pad >= KH * KW * cin_pad && k_pad % 32 == 0,
               "pack_conv_weight: bad args (Cin %d cin_pad %d k_pad %d)", Cin, cin_pad, k_pad);
    const size_t total = (size_t)Cout * k_pad;
    hipLaunchKernelGGL(k_pack_weight, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)s, w_oihw, w_packed, Cout, Cin,
                       KH, KW, cin_pad, k_pad);
    return ym_check_launch("pack_conv_weight");
}

extern "C" int ym_fold_bn(const float* gamma, const float* beta, const float* mean, const float* var, float eps,
                          float* scale, float* shift, int C, ym_stream_t s) {
    YM_REQUIRE(gamma && beta && mean && var && scale && shift && C > 0, "fold_bn: bad args");
    hipLaunchKernelGGL(k_fold_bn, dim3(ym_cdiv(C, 256)), dim3(256), 0, (hipStream_t)s, gamma, beta, mean, var, eps, scale,
                       shift, C);
    return ym_check_launch("fold_bn");
}

extern "C" int ym_maxpool3x3s2_fwd_idx(const float* in, float* out, uint8_t* idx, int B, int H, int W, int C, ym_stream_t s) {
    YM_REQUIRE(in && out && idx && B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "maxpool: bad args (C %% 4 != 0?)");
    YM_REQUIRE(((uintptr_t)idx & 3) == 0, "maxpool: idx must be 4-byte aligned");
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    const size_t total = (size_t)B * Ho * Wo * (C / 4);
    hipLaunchKernelGGL(k_maxpool3x3s2, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)s, in, out, B, H, W, C / 4, Ho, Wo, idx);
    return ym_check_launch("maxpool3x3s2");
}

extern "C" int ym_maxpool3x3s2_fwd(const float* in, float* out, int B, int H, int W, int C, ym_stream_t s) {
    YM_REQUIRE(in && out && B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "maxpool: bad args (C %% 4 != 0?)");
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    const size_t total = (size_t)B * Ho * Wo * (C / 4);
    hipLaunchKernelGGL(k_maxpool3x3s2, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)s, in, out, B, H, W, C / 4, Ho, Wo,
                       (uint8_t*)nullptr);
    return ym_check_launch("maxpool3x3s2");
}

extern "C" int ym_bilinear2x_fwd(const float* in, float* out, int B, int H, int W, int C, int align_corners, ym_stream_t s) {
    YM_REQUIRE(in && out && B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "bilinear2x: bad args (C %% 4 != 0?)");
    const size_t total = (size_t)B * 2 * H * 2 * W * (C / 4);
    hipLaunchKernelGGL(k_bilinear2x, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)s, in, out, B, H, W, C / 4,
                       align_corners ? 1 : 0);
    return ym_check_launch("bilinear2x");
}

extern "C" int ym_softmax_rows(const float* in, float* out, int64_t rows, int C, ym_stream_t s) {
    YM_REQUIRE(in && out && rows > 0 && C > 0 && C <= 256, "softmax_rows: C must be in 1..256");
    const int64_t waves = rows;
    int grid = (int)((waves + 3) / 4);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(k_softmax_rows, dim3(grid), dim3(256), 0, (hipStream_t)s, in, out, (long long)rows, C);
    return ym_check_launch("softmax_rows");
}

// ---- val_aug on the GPU (reference utils/augmentations.py:138-165,168-189,212-227) ----------
// HWC BGR image (uint8 or float32) -> pad to a square with norm_mean (top-left placement) -> bilinear resize to S x S
// (cv2.resize INTER_LINEAR on float32 == half-pixel centres, source index clamped at 0, i1 = min(i0+1, n-1)) ->
// (x - mean) / std -> BGR->RGB -> CHW.  The padded square is never materialised: a source pixel outside the image
// simply evaluates to norm_mean.  One thread per output pixel, 3 channels.
namespace {
template <typename T>
__global__ __launch_bounds__(256) void k_val_preprocess(const T* __restrict__ img, int H, int W, int S, f32x4 mean, f32x4 stdv,
                                                         float* __restrict__ out) {
    const int P = H > W ? H : W;
    const float scale = (float)P / (float)S;
    const int total = S * S;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int oy = i / S, ox = i - oy * S;
        float sy = scale * ((float)oy + 0.5f) - 0.5f, sx = scale * ((float)ox + 0.5f) - 0.5f;
        if (sy < 0.f) sy = 0.f;
        if (sx < 0.f) sx = 0.f;
        int y0 = (int)sy, x0 = (int)sx;
        if (y0 > P - 1) y0 = P - 1;
        if (x0 > P - 1) x0 = P - 1;
        const int y1 = y0 + (y0 < P - 1 ? 1 : 0), x1 = x0 + (x0 < P - 1 ? 1 : 0);
        const float ly = sy - (float)y0, lx = sx - (float)x0, hy = 1.f - ly, hx = 1.f - lx;
        auto px = [&](int y, int x, int c) -> float {
            return (y < H && x < W) ? (float)img[((size_t)y * W + x) * 3 + c] : mean[c];
        };
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float v = hy * (hx * px(y0, x0, c) + lx * px(y0, x1, c)) + ly * (hx * px(y1, x0, c) + lx * px(y1, x1, c));
            out[(size_t)(2 - c) * total + i] = (v - mean[c]) / stdv[c];          // BGR -> RGB plane order
        }
    }
}
}  // namespace

extern "C" int ym_val_preprocess(const void* img_hwc_bgr, int is_uint8, int H, int W, int S, const float* mean_bgr,
                                 const float* std_bgr, float* out_chw_rgb, ym_stream_t s) {
    YM_REQUIRE(img_hwc_bgr && out_chw_rgb && mean_bgr && std_bgr && H > 0 && W > 0 && S > 0, "val_preprocess: bad args");
    const f32x4 mean = {mean_bgr[0], mean_bgr[1], mean_bgr[2], 0.f}, stdv = {std_bgr[0], std_bgr[1], std_bgr[2], 1.f};
    const int grid = (S * S + 255) / 256;
    if (is_uint8)
        hipLaunchKernelGGL(k_val_preprocess<unsigned char>, dim3(grid), dim3(256), 0, (hipStream_t)s,
                           (const unsigned char*)img_hwc_bgr, H, W, S, mean, stdv, out_chw_rgb);
    else
        hipLaunchKernelGGL(k_val_preprocess<float>, dim3(grid), dim3(256), 0, (hipStream_t)s, (const float*)img_hwc_bgr, H, W, S,
                           mean, stdv, out_chw_rgb);
    return ym_check_launch("val_preprocess");
}
