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

// ---------------------------------------------------------------------------------------------------------------------------
// after_nms in ONE launch for a batch of images: assemble (coef x prototype, sigmoid, crop) -> bilinear resize to S x S ->
// > 0.5 -> slice to img_h x img_w  (utils/output_utils.py:217-228), without the [n][Hp][Wp] soft masks ever reaching HBM.
//
// grid = (output tiles, detection slot, image); a workgroup owns a FT_H x FT_W output tile of one detection.  The cropped soft
// mask is exactly 0 outside the detection's window, and bilinear interpolation of zeros is +0 -> "> 0.5" is false: a tile whose
// source patch lies outside the window is a pure zero fill (16-byte stores, the HBM-bound bulk of the n*img_h*img_w*4 output
// bytes).  Active tiles first build their source patch of the soft mask in LDS (thread = prototype pixel: 32-float dot with
// the coefficients + sigmoid + crop test, the same fp32 formula as k_mask_assemble up to the summation order), then resize
// from LDS with ATen's source-index arithmetic.  Detection slots >= the image's count (read on the device) exit at once.
// ---------------------------------------------------------------------------------------------------------------------------
constexpr int FT_W = 256, FT_H = 16;          // output tile: 64 lanes x float4 wide, 4 waves x 4 rows high
constexpr int FP_W = 96, FP_H = 10;           // source patch capacity (floats): checked on the host against the scale

__global__ __launch_bounds__(256) void k_masks_fused(const float* __restrict__ proto, const float* __restrict__ coefs,
                                                      const float* __restrict__ boxes, const int32_t* __restrict__ counts,
                                                      int max_det, int Hp, int Wp, int img_h, int img_w, int do_crop,
                                                      float* __restrict__ out) {
    __shared__ float patch[FP_H * FP_W];
    __shared__ __attribute__((aligned(16))) float cf[32];
    const int b = blockIdx.z, d = blockIdx.y;
    const int n = counts ? counts[b] : max_det;
    if (d >= n) return;
    const int tiles_x = (img_w + FT_W - 1) / FT_W;
    const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
    const int oy0 = ty * FT_H, ox0 = tx * FT_W;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int S = img_h > img_w ? img_h : img_w;
    const float sy = (float)Hp / (float)S, sx = (float)Wp / (float)S;
    const size_t slot = (size_t)b * max_det + d;

    float x1 = 0.f, x2 = (float)Wp, y1 = 0.f, y2 = (float)Hp;
    if (do_crop) {
        const f32x4 bx = *reinterpret_cast<const f32x4*>(boxes + slot * 4);
        crop_span(bx[0], bx[2], (float)Wp, x1, x2);
        crop_span(bx[1], bx[3], (float)Hp, y1, y2);
    }
    // source patch of this tile (src_coord is monotonic in the destination index)
    int py0, py1, px0, px1, t0, t1; float l;
    const int oy_last = min(oy0 + FT_H, img_h) - 1, ox_last = min(ox0 + FT_W, img_w) - 1;
    src_coord(oy0, sy, Hp, py0, t1, l);
    src_coord(oy_last, sy, Hp, t0, py1, l);
    src_coord(ox0, sx, Wp, px0, t1, l);
    src_coord(ox_last, sx, Wp, t0, px1, l);
    const int ph = py1 - py0 + 1, pw = px1 - px0 + 1;
    // any source pixel of the patch inside the crop window [x1,x2) x [y1,y2)?  (float compares like the reference's crop)
    const bool active = (float)px1 >= x1 && (float)px0 < x2 && (float)py1 >= y1 && (float)py0 < y2;
    float* obase = out + slot * (size_t)img_h * img_w;
    const bool vec = (img_w & 3) == 0;
    const int x = ox0 + lane * 4;
    if (!active) {
#pragma unroll
        for (int r = 0; r < FT_H / 4; ++r) {
            const int y = oy0 + wave * (FT_H / 4) + r;
            if (y >= img_h || x >= img_w) continue;
            float* dst = obase + (size_t)y * img_w + x;
            if (vec) *reinterpret_cast<f32x4*>(dst) = f32x4{0.f, 0.f, 0.f, 0.f};
            else
                for (int e = 0; e < 4 && x + e < img_w; ++e) dst[e] = 0.f;
        }
        return;
    }
    if (tid < 32) cf[tid] = coefs[slot * 32 + tid];
    __syncthreads();
    const float* pimg = proto + (size_t)b * Hp * Wp * 32;
    for (int i = tid; i < ph * pw; i += 256) {
        const int r = i / pw, c = i - r * pw;
        const int py = py0 + r, px = px0 + c;
        const float fx = (float)px, fy = (float)py;
        float v = 0.f;
        if (fx >= x1 && fx < x2 && fy >= y1 && fy < y2) {
            const f32x4* pr = reinterpret_cast<const f32x4*>(pimg + ((size_t)py * Wp + px) * 32);
            float acc = 0.f;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const f32x4 pv = pr[q];
                const f32x4 cv = *reinterpret_cast<const f32x4*>(cf + q * 4);
                acc = __builtin_fmaf(cv[0], pv[0], acc);
                acc = __builtin_fmaf(cv[1], pv[1], acc);
                acc = __builtin_fmaf(cv[2], pv[2], acc);
                acc = __builtin_fmaf(cv[3], pv[3], acc);
            }
            v = 1.f / (1.f + expf(-acc));
        }
        patch[r * FP_W + c] = v;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < FT_H / 4; ++r) {
        const int y = oy0 + wave * (FT_H / 4) + r;
        if (y >= img_h || x >= img_w) continue;
        int y0, y1i; float ly;
        src_coord(y, sy, Hp, y0, y1i, ly);
        const float hy = 1.f - ly;
        const float* r0 = patch + (y0 - py0) * FP_W - px0;
        const float* r1 = patch + (y1i - py0) * FP_W - px0;
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            int x0, x1i; float lx;
            src_coord(x + e < img_w ? x + e : img_w - 1, sx, Wp, x0, x1i, lx);
            const float hx = 1.f - lx;
            const float v = hy * (hx * r0[x0] + lx * r0[x1i]) + ly * (hx * r1[x0] + lx * r1[x1i]);
            o[e] = v > 0.5f ? 1.f : 0.f;
        }
        float* dst = obase + (size_t)y * img_w + x;
        if (vec) *reinterpret_cast<f32x4*>(dst) = f32x4{o[0], o[1], o[2], o[3]};
        else
            for (int e = 0; e < 4 && x + e < img_w; ++e) dst[e] = o[e];
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

// Does the fused kernel's LDS patch hold the source pixels of one output tile at this scale?
static bool fused_fits(int Hp, int Wp, int img_h, int img_w) {
    const int S = img_h > img_w ? img_h : img_w;
    const double sy = (double)Hp / S, sx = (double)Wp / S;
    return (int)(FT_H * sy) + 3 <= FP_H && (int)(FT_W * sx) + 3 <= FP_W;
}

extern "C" int ym_after_nms_batch(const float* proto, const float* coefs, float* boxes, const int32_t* counts, int B, int max_det,
                                  int Hp, int Wp, int K, int img_h, int img_w, int do_crop, float* masks, int32_t* boxes_px,
                                  void* workspace, size_t workspace_bytes, ym_stream_t s) {
    YM_REQUIRE(K == 32, "after_nms: coefficient dim must be 32, got %d", K);
    YM_REQUIRE(B >= 1 && B <= 65535 && max_det >= 1 && max_det <= 65535 && Hp > 0 && Wp > 0 && img_h > 0 && img_w > 0, "after_nms: bad shape");
    YM_REQUIRE(proto && coefs && boxes && masks && boxes_px, "after_nms: null pointer");
    hipStream_t st = (hipStream_t)s;
    if (fused_fits(Hp, Wp, img_h, img_w)) {
        const int tiles = ym_cdiv(img_w, FT_W) * ym_cdiv(img_h, FT_H);
        hipLaunchKernelGGL(k_masks_fused, dim3(tiles, max_det, B), dim3(256), 0, st, proto, coefs, boxes, counts, max_det, Hp, Wp,
                           img_h, img_w, do_crop, masks);
    } else {
        // strong down-scaling (image smaller than ~2.7x the prototype map): the two-kernel path through a soft-mask scratch
        const size_t soft_bytes = (size_t)max_det * Hp * Wp * sizeof(float);
        if (!workspace || workspace_bytes < soft_bytes) { ym_set_error("after_nms: workspace %zu B < %zu B", workspace_bytes, soft_bytes); return YM_ENOSPC; }
        for (int b = 0; b < B; ++b) {   // (all max_det slots: rows past the count are garbage the caller never reads)
            const size_t slot = (size_t)b * max_det;
            int rc = ym_mask_assemble(proto + (size_t)b * Hp * Wp * 32, coefs + slot * 32, boxes + slot * 4, max_det, Hp, Wp, K, do_crop,
                                      (float*)workspace, s);
            if (rc != YM_OK) return rc;
            rc = ym_mask_resize_binarize((const float*)workspace, max_det, Hp, Wp, img_h, img_w, masks + slot * (size_t)img_h * img_w, s);
            if (rc != YM_OK) return rc;
        }
    }
    const int S = img_h > img_w ? img_h : img_w;
    const int cnt = B * max_det * 4;
    hipLaunchKernelGGL(k_boxes_to_pixels, dim3(ym_cdiv(cnt, 256)), dim3(256), 0, st, boxes, boxes_px, cnt, (float)S);
    return ym_check_launch("after_nms_batch");
}

extern "C" size_t ym_after_nms_batch_workspace_bytes(int max_det, int Hp, int Wp, int img_h, int img_w) {
    return fused_fits(Hp, Wp, img_h, img_w) ? 0 : (size_t)max_det * Hp * Wp * sizeof(float);
}
