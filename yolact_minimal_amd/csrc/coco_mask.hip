// COCO annotation -> dense instance mask on the device (the reader's `self.coco.annToMask(aa)`,
// utils/coco.py:96): pycocotools annToRLE (frPyObjects on the polygon list, merge = union) + decode.  Algorithm = cocoapi
// common/maskApi.c rleFrPoly / rleDecode, restated in oracle/coco_ref.py.  Only the polygon vertices (a few hundred bytes) cross
// PCIe; the dense [n, H, W] uint8 masks are born in HBM where train_aug / the loss consume them.
//
// One workgroup per annotation.  The run-length code of a polygon is a sorted list of toggle positions in the column-major pixel
// order, so no sort is needed: every boundary point XORs one bit of a column-major bitmap (LDS when 2 bitmaps fit, else the
// caller's workspace), a prefix-XOR down each column plus a parity carry across the columns turns toggles into the fill, and
// the polygons of one annotation are OR-ed.  Each wave walks whole edges (lane = step along the edge); the doubles follow the C
// expression order without contraction (__dmul_rn / __dadd_rn / __ddiv_rn).
#include "ym_common.h"

namespace {

constexpr int NT = 512;
constexpr int NW = NT / 64;
constexpr int MAXW = 4096;
constexpr size_t LDS_BUDGET = 160 * 1024 - MAXW - 256;     // static: one parity byte per column + scan scratch

struct Pt { int u, v; };

__device__ __forceinline__ int scaled5(double c) { return (int)__dadd_rn(__dmul_rn(5.0, c), 0.5); }

// d-th point of the walk from (xs, ys) to (xe, ye) on the 5x grid (rleFrPoly's edge loop; `flip` keeps the walk monotone in t
// while the points still come out in start -> end order)
__device__ __forceinline__ Pt edge_point(int xs, int ys, int xe, int ye, int d) {
    const int dx = abs(xe - xs), dy = abs(ys - ye);
    const bool flip = (dx >= dy && xs > xe) || (dx < dy && ys > ye);
    if (flip) { int t = xs; xs = xe; xe = t; t = ys; ys = ye; ye = t; }
    Pt p;
    if (dx >= dy) {
        const int t = flip ? dx - d : d;
        p.u = t + xs;
        if (dx == 0) p.v = ys;                                   // repeated vertex: C has 0/0 here; u never changes, v unused
        else {
            const double s = __ddiv_rn((double)(ye - ys), (double)dx);
            p.v = (int)__dadd_rn(__dadd_rn((double)ys, __dmul_rn(s, (double)t)), 0.5);
        }
    } else {
        const int t = flip ? dy - d : d;
        p.v = t + ys;
        const double s = __ddiv_rn((double)(xe - xs), (double)dy);
        p.u = (int)__dadd_rn(__dadd_rn((double)xs, __dmul_rn(s, (double)t)), 0.5);
    }
    return p;
}

__device__ __forceinline__ void toggle(uint32_t* tog, int HWS, int H, int W, int x, int y) {
    if (y >= H) { x += 1; y = 0; }                               // position x*H + H is the first pixel of the next column
    if (x >= W) return;                                          // ... or the end of the image
    atomicXor(&tog[(size_t)x * HWS + (y >> 5)], 1u << (y & 31));
}

// boundary points of one polygon (k vertices at v[0 .. 2k)) -> toggles
__device__ void polygon_toggles(const double* __restrict__ v, int k, int H, int W, int HWS, uint32_t* tog) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int j = wave; j < k; j += NW) {
        const int jn = j + 1 == k ? 0 : j + 1;
        const int xs = scaled5(v[2 * j]), ys = scaled5(v[2 * j + 1]), xe = scaled5(v[2 * jn]), ye = scaled5(v[2 * jn + 1]);
        const int L = max(abs(xe - xs), abs(ye - ys)) + 1;
        for (int d = lane; d < L; d += 64) {
            const Pt cur = edge_point(xs, ys, xe, ye, d);
            Pt prev;
            if (d > 0) prev = edge_point(xs, ys, xe, ye, d - 1);
            else if (j > 0) {                                    // the point before is the last one of the previous edge
                const int xp = scaled5(v[2 * j - 2]), yp = scaled5(v[2 * j - 1]);
                prev = edge_point(xp, yp, xs, ys, max(abs(xs - xp), abs(ys - yp)));
            } else continue;
            if (cur.u == prev.u) continue;
            double xd = (double)(cur.u < prev.u ? cur.u : cur.u - 1);
            xd = __dadd_rn(__ddiv_rn(__dadd_rn(xd, 0.5), 5.0), -0.5);
            if (floor(xd) != xd || xd < 0 || xd > (double)(W - 1)) continue;
            double yd = (double)(cur.v < prev.v ? cur.v : prev.v);
            yd = __dadd_rn(__ddiv_rn(__dadd_rn(yd, 0.5), 5.0), -0.5);
            if (yd < 0) yd = 0; else if (yd > (double)H) yd = (double)H;
            toggle(tog, HWS, H, W, (int)xd, (int)ceil(yd));
        }
    }
}

// run lengths (column-major, starting with zeros) -> toggles at every run boundary
__device__ void runs_toggles(const uint32_t* __restrict__ cnt, int R, int H, int W, int HWS, uint32_t* tog, uint32_t* s_scan) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t carry = 0;
    for (int j0 = 0; j0 < R; j0 += NT) {
        const int j = j0 + threadIdx.x;
        const uint32_t c = j < R ? cnt[j] : 0u;
        uint32_t inc = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t t = __shfl_up(inc, o);
            if (lane >= o) inc += t;
        }
        __syncthreads();
        if (lane == 63) s_scan[wave] = inc;
        __syncthreads();
        uint32_t before = carry;
        for (int w2 = 0; w2 < wave; ++w2) before += s_scan[w2];
        uint32_t total = 0;
        for (int w2 = 0; w2 < NW; ++w2) total += s_scan[w2];
        const uint32_t end = before + inc;                       // first pixel after run j
        if (j < R - 1 && end < (uint32_t)H * (uint32_t)W) toggle(tog, HWS, H, W, (int)(end / (uint32_t)H), (int)(end % (uint32_t)H));
        carry += total;
    }
}

template <int SRC>   // 0: polygons, 1: run lengths
__global__ __launch_bounds__(NT) void k_ann_to_mask(const double* __restrict__ xy, const uint32_t* __restrict__ runs,
                                                    const int32_t* __restrict__ item_off, const int32_t* __restrict__ ann_off, int H,
                                                    int W, int HWS, uint8_t* __restrict__ masks, uint32_t* __restrict__ gws) {
    extern __shared__ uint32_t smem[];
    __shared__ uint8_t s_par[MAXW];
    __shared__ uint32_t s_scan[NW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nwords = W * HWS, HW32 = (H + 31) >> 5;
    uint32_t* tog = gws ? gws + (size_t)blockIdx.x * 2 * nwords : smem;
    uint32_t* acc = tog + nwords;
    for (int i = tid; i < nwords; i += NT) acc[i] = 0;

    const int p0 = SRC == 0 ? ann_off[blockIdx.x] : blockIdx.x, p1 = SRC == 0 ? ann_off[blockIdx.x + 1] : blockIdx.x + 1;
    for (int p = p0; p < p1; ++p) {
        for (int i = tid; i < nwords; i += NT) tog[i] = 0;
        __syncthreads();
        const int b = item_off[p], k = item_off[p + 1] - b;
        if (SRC == 0) polygon_toggles(xy + 2 * (size_t)b, k, H, W, HWS, tog);
        else runs_toggles(runs + b, k, H, W, HWS, tog, s_scan);
        __syncthreads();
        // inclusive prefix-XOR down every column; s_par[x] = parity of the whole column
        for (int x = tid; x < W; x += NT) {
            uint32_t par = 0;
            uint32_t* col = tog + (size_t)x * HWS;
            for (int i = 0; i < HW32; ++i) {
                uint32_t w = col[i];
                w ^= w << 1; w ^= w << 2; w ^= w << 4; w ^= w << 8; w ^= w << 16;
                if (par) w = ~w;
                col[i] = w;
                par = w >> 31;
            }
            s_par[x] = (uint8_t)par;
        }
        __syncthreads();
        // the run-length order continues from one column into the next: carry[x] = parity of all the columns before x
        uint32_t carry = 0;
        for (int x0 = 0; x0 < W; x0 += NT) {
            const int x = x0 + tid;
            const uint32_t mine = x < W ? s_par[x] : 0u;
            uint32_t inc = mine;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t t = __shfl_up(inc, o);
                if (lane >= o) inc ^= t;
            }
            if (lane == 63) s_scan[wave] = inc;
            __syncthreads();
            uint32_t before = carry, total = 0;
            for (int w2 = 0; w2 < NW; ++w2) {
                if (w2 < wave) before ^= s_scan[w2];
                total ^= s_scan[w2];
            }
            if (x < W) {
                const uint32_t flipm = (before ^ inc ^ mine) ? ~0u : 0u;      // exclusive parity
                const uint32_t* col = tog + (size_t)x * HWS;
                uint32_t* a = acc + (size_t)x * HWS;
                for (int i = 0; i < HW32; ++i) a[i] |= col[i] ^ flipm;
            }
            carry ^= total;
            __syncthreads();
        }
    }
    __syncthreads();
    // row-major uint8 output, four pixels per store when the mask size allows aligned words
    const unsigned P = (unsigned)H * (unsigned)W;
    uint8_t* out = masks + (size_t)blockIdx.x * P;
    if ((P & 3u) == 0) {
        for (unsigned g = tid; g < P / 4; g += NT) {
            unsigned y = (4 * g) / (unsigned)W, x = 4 * g - y * (unsigned)W;
            uint32_t pack = 0;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                pack |= ((acc[(size_t)x * HWS + (y >> 5)] >> (y & 31)) & 1u) << (8 * b);
                if (++x == (unsigned)W) { x = 0; ++y; }
            }
            reinterpret_cast<uint32_t*>(out)[g] = pack;
        }
    } else {
        for (unsigned i = tid; i < P; i += NT) {
            const unsigned y = i / (unsigned)W, x = i - y * (unsigned)W;
            out[i] = (uint8_t)((acc[(size_t)x * HWS + (y >> 5)] >> (y & 31)) & 1u);
        }
    }
}

inline int col_stride(int H) { return ((H + 31) >> 5) | 1; }     // words per bitmap column, odd (LDS banks)
inline size_t bitmap_bytes(int H, int W) { return (size_t)2 * W * col_stride(H) * 4; }

template <int SRC>
int launch(const double* xy, const uint32_t* runs, const int32_t* item_off, const int32_t* ann_off, int n, int H, int W,
           uint8_t* masks, void* workspace, size_t workspace_bytes, ym_stream_t s, const char* what) {
    YM_REQUIRE(item_off && masks && (SRC == 1 || ann_off), "%s: null pointer", what);
    YM_REQUIRE(n > 0 && H > 0 && W > 0 && W <= MAXW && (long long)H * W < (1ll << 31), "%s: need n > 0, 0 < W <= %d, H*W < 2^31", what, MAXW);
    YM_REQUIRE(((uintptr_t)masks & 3) == 0, "%s: masks must be 4-byte aligned", what);
    const size_t bm = bitmap_bytes(H, W);
    size_t lds = bm;
    uint32_t* gws = nullptr;
    if (bm > LDS_BUDGET) {
        if (!workspace || workspace_bytes < (size_t)n * bm) { ym_set_error("%s: workspace < %zu bytes", what, (size_t)n * bm); return YM_ENOSPC; }
        gws = (uint32_t*)workspace;
        lds = 0;
    }
    static YmLdsAttr attr = {};                                   // (one per SRC instantiation of this function template)
    if (int rc = ym_ensure_dyn_lds(attr, reinterpret_cast<const void*>(k_ann_to_mask<SRC>), LDS_BUDGET, what)) return rc;
    hipLaunchKernelGGL(k_ann_to_mask<SRC>, dim3(n), dim3(NT), lds, (hipStream_t)s, xy, runs, item_off, ann_off, H, W, col_stride(H), masks, gws);
    return ym_check_launch(what);
}

}  // namespace

extern "C" size_t ym_ann_to_mask_workspace_bytes(int n, int H, int W) {
    if (n <= 0 || H <= 0 || W <= 0) return 0;
    const size_t bm = bitmap_bytes(H, W);
    return bm > LDS_BUDGET ? (size_t)n * bm : 0;
}

extern "C" int ym_poly_to_mask(const double* xy, const int32_t* poly_off, const int32_t* ann_off, int n, int H, int W, uint8_t* masks,
                               void* workspace, size_t workspace_bytes, ym_stream_t s) {
    YM_REQUIRE(xy, "poly_to_mask: null pointer");
    return launch<0>(xy, nullptr, poly_off, ann_off, n, H, W, masks, workspace, workspace_bytes, s, "poly_to_mask");
}

extern "C" int ym_runs_to_mask(const uint32_t* counts, const int32_t* run_off, int n, int H, int W, uint8_t* masks, void* workspace,
                               size_t workspace_bytes, ym_stream_t s) {
    YM_REQUIRE(counts, "runs_to_mask: null pointer");
    return launch<1>(nullptr, counts, run_off, nullptr, n, H, W, masks, workspace, workspace_bytes, s, "runs_to_mask");
}
