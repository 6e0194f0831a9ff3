// Mask assembly for gfx950.
//  k_mask_assemble : sigmoid(coef[n][32] x proto[P][32]^T) on the f32 MFMA pipe with the crop fused into the
//                    epilogue.  GEMM roles are chosen for the STORE side: MFMA "A" = coefficients (i = detection),
//                    "B" = prototypes (j = pixel), so D's column index (lane & 31) is the pixel and each accumulator
//                    register row is written as 32 consecutive pixels of one detection (128-B segments).
//                    Both operands are K-contiguous in HBM ([.][32] floats), so lane half h reads k = 8g+4h..+3
//                    as one 16-byte global load per group g — no LDS staging at all (K = 32 fits in registers).
//                    Algorithmic bytes: P*32*4 (proto, read once) + n*P*4 (masks, written once): HBM-bound.
//  k_mask_resize   : bilinear (align_corners=False) to S x S, > 0.5, cropped to img_h x img_w; the output
//                    (n*img_h*img_w*4 B) dominates, written with 16-byte stores.
#pragma clang fp contract(off)
#include "ym_common.h"

namespace {

__device__ __forceinline__ void crop_span(float a, float b, float size, float& lo, float& hi) {
    // utils/box_utils.py:117-132 with padding = 1
    a = a * size;
    b = b * size;
    lo = fminf(a, b);
    hi = fmaxf(a, b);
    lo = lo - 1.f;
    lo = lo < 0.f ? 0.f : lo;
    hi = hi + 1.f;
    hi = hi > size ? size : hi;
}

__global__ __launch_bounds__(256) void k_mask_assemble(const float* __restrict__ proto, const float* __restrict__ coefs,
                                                        const float* __restrict__ boxes, int n, int Hp, int Wp,
                                                        int do_crop, float* __restrict__ out) {
    const int P = Hp * Wp;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pix0 = (blockIdx.x * 4 + wave) * 32;
    if (pix0 >= P) return;
    const int pj = lane & 31, h = lane >> 5;
    const int pix = pix0 + pj;
    f32x4 pb[4];
#pragma unroll
    for (int g = 0; g < 4; ++g)
        pb[g] = pix < P ? *reinterpret_cast<const f32x4*>(proto + (size_t)pix * 32 + g * 8 + h * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
    const int py = pix / Wp, px = pix - py * Wp;
    const float fx = (float)px, fy = (float)py;

    for (int d0 = 0; d0 < n; d0 += 32) {
        const int det = d0 + pj;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 ca = det < n ? *reinterpret_cast<const f32x4*>(coefs + (size_t)det * 32 + g * 8 + h * 4)
                                     : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ca[s], pb[g][s], acc, 0, 0, 0);
        }
        if (pix >= P) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int d = d0 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (d >= n) continue;
            float v = 1.f / (1.f + expf(-acc[r]));
            if (do_crop) {
                const f32x4 b = *reinterpret_cast<const f32x4*>(boxes + (size_t)d * 4);
                float x1, x2, y1, y2;
                crop_span(b[0], b[2], (float)Wp, x1, x2);
                crop_span(b[1], b[3], (float)Hp, y1, y2);
                const bool inside = fx >= x1 && fx < x2 && fy >= y1 && fy < y2;
                v = inside ? v : 0.f;
            }
            out[(size_t)d * P + pix] = v;
        }
    }
}

__device__ __forceinline__ void src_coord(int dst, float scale, int in_sz, int& i0, int& i1, float& l1) {
    float src = scale * ((float)dst + 0.5f) - 0.5f;
    if (src < 0.f) src = 0.f;
    i0 = (int)src;
    if (i0 > in_sz - 1) i0 = in_sz - 1;
    i1 = i0 + ((i0 < in_sz - 1) ? 1 : 0);
    l1 = src - (float)i0;
}

__global__ __launch_bounds__(256) void k_mask_resize(const float* __restrict__ masks, int n, int Hp, int Wp, int img_h,
                                                      int img_w, float* __restrict__ out) {
    const int S = img_h > img_w ? img_h : img_w;
    const float sy = (float)Hp / (float)S, sx = (float)Wp / (float)S;
    const int wq = (img_w + 3) >> 2;
    const size_t total = (size_t)n * img_h * wq;
    const bool vec = (img_w & 3) == 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int xq = (int)(i % wq);
        size_t t = i / wq;
        const int y = (int)(t % img_h);
        const int d = (int)(t / img_h);
        int y0, y1; float ly;
        src_coord(y, sy, Hp, y0, y1, ly);
        const float hy = 1.f - ly;
        const float* r0 = masks + ((size_t)d * Hp + y0) * Wp;
        const float* r1 = masks + ((size_t)d * Hp + y1) * Wp;
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int x = xq * 4 + e;
            int x0, x1; float lx;
            src_coord(x < img_w ? x : img_w - 1, sx, Wp, x0, x1, lx);
            const float hx = 1.f - lx;
            const float v = hy * (hx * r0[x0] + lx * r0[x1]) + ly * (hx * r1[x0] + lx * r1[x1]);
            o[e] = v > 0.5f ? 1.f : 0.f;
        }
        float* dst = out + ((size_t)d * img_h + y) * img_w + xq * 4;
        if (vec) {
            *reinterpret_cast<f32x4*>(dst) = f32x4{o[0], o[1], o[2], o[3]};
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (xq * 4 + e < img_w) dst[e] = o[e];
        }
    }
}

__global__ void k_boxes_to_pixels(float* boxes, int32_t* px, int count, float S) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) {
        const float v = boxes[i] * S;
        boxes[i] = v;
        px[i] = (int32_t)v;   // trunc toward zero, like Tensor.int()
    }
}

}  // namespace

extern "C" int ym_mask_assemble(const float* proto, const float* coefs, const float* boxes, int n, int Hp, int Wp, int K,
                                int do_crop, float* out, ym_stream_t s) {
    YM_REQUIRE(K == 32, "mask_assemble: coefficient dim must be 32, got %d", K);
    YM_REQUIRE(n >= 0 && Hp > 0 && Wp > 0, "mask_assemble: bad shape");
    if (n == 0) return YM_OK;
    YM_REQUIRE(proto && coefs && out && (boxes || !do_crop), "mask_assemble: null pointer");
    const int P = Hp * Wp;
    const int waves = ym_cdiv(P, 32);
    hipLaunchKernelGGL(k_mask_assemble, dim3(ym_cdiv(waves, 4)), dim3(256), 0, (hipStream_t)s, proto, coefs, boxes, n, Hp,
                       Wp, do_crop, out);
    return ym_check_launch("mask_assemble");
}

extern "C" int ym_mask_resize_binarize(const float* masks, int n, int Hp, int Wp, int img_h, int img_w, float* out,
                                       ym_stream_t s) {
    YM_REQUIRE(n >= 0 && Hp > 0 && Wp > 0 && img_h > 0 && img_w > 0, "mask_resize: bad shape");
    if (n == 0) return YM_OK;
    YM_REQUIRE(masks && out, "mask_resize: null pointer");
    const size_t total = (size_t)n * img_h * ((img_w + 3) / 4);
    size_t grid = (total + 255) / 256;
    if (grid > 16384) grid = 16384;
    hipLaunchKernelGGL(k_mask_resize, dim3((int)grid), dim3(256), 0, (hipStream_t)s, masks, n, Hp, Wp, img_h, img_w, out);
    return ym_check_launch("mask_resize");
}

extern "C" int ym_boxes_to_pixels(float* boxes_f, int32_t* boxes_px, int n, float S, ym_stream_t s) {
    YM_REQUIRE(n >= 0, "boxes_to_pixels: n < 0");
    if (n == 0) return YM_OK;
    YM_REQUIRE(boxes_f && boxes_px, "boxes_to_pixels: null pointer");
    hipLaunchKernelGGL(k_boxes_to_pixels, dim3(ym_cdiv(n * 4, 256)), dim3(256), 0, (hipStream_t)s, boxes_f, boxes_px, n * 4, S);
    return ym_check_launch("boxes_to_pixels");
}
