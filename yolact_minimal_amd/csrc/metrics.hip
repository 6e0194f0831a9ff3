// Evaluation inner products right after after_nms: mask IoU of the predicted binary masks against the
// ground-truth masks (reference utils/box_utils.py:189-200: a [n,HW]x[HW,g] fp32 matmul on {0,1} masks), pixel-box IoU
// (utils/box_utils.py:8-37) and the greedy per-class matching of prep_metrics (utils/common_utils.py:174-216).
// The masks are {0,1}, so the matmul is a popcount of ANDed bit rows: HBM-bound (every mask read exactly once, 123 MB for 100
// masks at 480x640), exact in integers -> the IoU is bit-identical to the reference's fp32 result (counts < 2^24).
#pragma clang fp contract(off)
#include "ym_common.h"

namespace {

// pixels per workgroup pass = SEGS x 256 (4 x SEGS 64-bit words per mask row; 2 x 128 rows of them in LDS).  The host picks SEGS so
// that the chunks of a row fill the 256 CUs in whole rounds (pick_segs): 480x640 masks = 300 chunks of 1024 pixels were 1.17 rounds
// (31.4 us for 100 + 15 masks), 240 chunks of 1280 are one (26.2 us).
constexpr int MAXR = 128;              // rows of each side held in LDS at once
constexpr int RPI = 4;                 // rows a wave has in flight per iteration of the packing pass

// Bits of a mask-row chunk: lane l of a 256-pixel segment holds pixels 4l..4l+3, one ballot per component = four 64-bit
// words.  The bit order inside a chunk is a fixed permutation of the pixel order, the same for both operands, so
// popcount(a & b) is unchanged.
// grid: (pixel chunks, groups of 128 gt rows).  Every workgroup writes its own partial counts inter[chunk][n][g], area_a[chunk][n],
// area_b[chunk][g] with plain stores (global atomics were the bottleneck: 1500 per workgroup); the finalize kernel sums the chunks.
template <int SEGS>
__global__ __launch_bounds__(512) void k_mask_inter(const float* __restrict__ A, int n, const float* __restrict__ Bm, int g, long long P,
                                                    int* __restrict__ inter, int* __restrict__ area_a, int* __restrict__ area_b) {
    constexpr int CHUNK = SEGS * 256, WORDS = SEGS * 4;
    __shared__ unsigned long long sa[MAXR][WORDS + 1];
    __shared__ unsigned long long sb[MAXR][WORDS + 1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
    const long long p0 = (long long)blockIdx.x * CHUNK;
    const int g0 = blockIdx.y * MAXR, gn = min(MAXR, g - g0);
    const bool fast = (P & 3) == 0 && p0 + CHUNK <= P;     // float4 loads need 16-byte aligned rows and a whole chunk
    inter += (size_t)blockIdx.x * n * g;
    area_a += (size_t)blockIdx.x * n;
    area_b += (size_t)blockIdx.x * g;
    for (int a0 = 0; a0 < n; a0 += MAXR) {
        const int an = min(MAXR, n - a0);
        __syncthreads();
        // pack rows: one wave takes a whole row chunk.  The pass is HBM-latency bound, so on the fast path (16-byte aligned rows,
        // a whole chunk inside the row: every chunk of a 480x640 mask) a wave requests RPI rows -- CHUNK/256 float4 loads per lane
        // and row -- before it turns the first one into ballots; lane q < WORDS keeps word q, one LDS store per row.
        const int rows = an + (a0 == 0 ? gn : 0);
        if (fast) {
            for (int r = wave; r < rows; r += RPI * nw) {
                f32x4 v[RPI][SEGS];
#pragma unroll
                for (int u = 0; u < RPI; ++u) {
                    const int rr = min(r + u * nw, rows - 1);       // (past the end: the last row again, not used)
                    const float* row = (rr < an ? A + (size_t)(a0 + rr) * P : Bm + (size_t)(g0 + rr - an) * P) + p0 + 4 * lane;
#pragma unroll
                    for (int sg = 0; sg < SEGS; ++sg) v[u][sg] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(row + sg * 256));
                }
#pragma unroll
                for (int u = 0; u < RPI; ++u) {
                    const int rr = r + u * nw;
                    if (rr < rows) {
                        unsigned long long mine = 0;
                        int c = 0;
#pragma unroll
                        for (int sg = 0; sg < SEGS; ++sg)
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const unsigned long long w = __ballot(v[u][sg][e] != 0.f);
                                mine = lane == sg * 4 + e ? w : mine;
                                c += __popcll(w);
                            }
                        const bool is_a = rr < an;
                        if (lane < WORDS) (is_a ? sa[rr] : sb[rr - an])[lane] = mine;
                        // areas (once per row: a rows only from the first gt group, gt rows only from the first a pass)
                        if (lane == 0) {
                            if (is_a) { if (blockIdx.y == 0) area_a[a0 + rr] = c; }
                            else area_b[g0 + rr - an] = c;
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);    // (row by row: the ballots of a row are wave-uniform SGPR pairs)
                }
            }
        } else {
            // the last, partial chunk of a row / rows that are not 16-byte aligned: pixel by pixel, one row per wave and iteration
            for (int rr = wave; rr < rows; rr += nw) {
                const bool is_a = rr < an;
                const float* row = is_a ? A + (size_t)(a0 + rr) * P : Bm + (size_t)(g0 + rr - an) * P;
                unsigned long long mine = 0;
                int c = 0;
                for (int q = 0; q < WORDS; ++q) {
                    const long long p = p0 + q * 64 + lane;
                    const unsigned long long w = __ballot(p < P && row[p] != 0.f);
                    mine = lane == q ? w : mine;
                    c += __popcll(w);
                }
                if (lane < WORDS) (is_a ? sa[rr] : sb[rr - an])[lane] = mine;
                if (lane == 0) {
                    if (is_a) { if (blockIdx.y == 0) area_a[a0 + rr] = c; }
                    else area_b[g0 + rr - an] = c;
                }
            }
        }
        __syncthreads();
        for (int pr = tid; pr < an * gn; pr += blockDim.x) {
            const int i = pr / gn, j = pr - i * gn;
            int c = 0;
#pragma unroll 8
            for (int w = 0; w < WORDS; ++w) c += __popcll(sa[i][w] & sb[j][w]);
            inter[(size_t)(a0 + i) * g + g0 + j] = c;
        }
    }
}

// sum the per-chunk partials (exact integers) and apply the reference's formula; 64 pairs x FSL chunk slices per workgroup (the
// slices' loads are independent: 4 slices took 16.5 us for 300 chunks -- a serial chain of 75 L2 round trips per thread)
constexpr int FSL = 16;
__global__ __launch_bounds__(64 * FSL) void k_mask_iou_finalize(const int* __restrict__ inter, const int* __restrict__ area_a,
                                                                const int* __restrict__ area_b, int chunks, int n, int g, float* __restrict__ iou) {
    __shared__ int s_i[FSL][64], s_a[FSL][64], s_b[FSL][64];
    const int lane = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + lane;
    int it = 0, aa = 0, ab = 0;
    if (e < n * g) {
        const int i = e / g, j = e - i * g;
#pragma unroll 4
        for (int c = sl; c < chunks; c += FSL) {
            it += inter[(size_t)c * n * g + e];
            aa += area_a[(size_t)c * n + i];
            ab += area_b[(size_t)c * g + j];
        }
    }
    s_i[sl][lane] = it; s_a[sl][lane] = aa; s_b[sl][lane] = ab;
    __syncthreads();
    if (sl == 0 && e < n * g) {
        int ti = 0, ta = 0, tb = 0;
#pragma unroll
        for (int q = 0; q < FSL; ++q) { ti += s_i[q][lane]; ta += s_a[q][lane]; tb += s_b[q][lane]; }
        const float fi = (float)ti, fa = (float)ta, fb = (float)tb;
        iou[e] = __fdiv_rn(fi, (fa + fb) - fi);          // inter / ((area1.t() + area2) - inter); 0/0 -> NaN like the reference
    }
}

__global__ void k_box_iou(const float* __restrict__ a, const float* __restrict__ b, int n, int g, float* __restrict__ iou) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n * g) return;
    const int i = e / g, j = e - i * g;
    const float* p = a + (size_t)i * 4;
    const float* q = b + (size_t)j * 4;
    float w = fminf(p[2], q[2]) - fmaxf(p[0], q[0]);
    float h = fminf(p[3], q[3]) - fmaxf(p[1], q[1]);
    w = w < 0.f ? 0.f : w;
    h = h < 0.f ? 0.f : h;
    const float inter = w * h;
    const float aa = (p[2] - p[0]) * (p[3] - p[1]), ab = (q[2] - q[0]) * (q[3] - q[1]);
    iou[e] = __fdiv_rn(inter, (aa + ab) - inter);
}

// prep_metrics' matching (utils/common_utils.py:186-216): block = (iou type, threshold), thread = class.  Predictions are
// visited in their given order; each takes the unused same-class gt with the largest IoU strictly above the running maximum
// (which starts AT the threshold; python compares float32->double IoUs with the double threshold).
constexpr int MAXG = 512;
__global__ __launch_bounds__(128) void k_match_detections(const float* __restrict__ iou_box, const float* __restrict__ iou_mask,
                                                          const int* __restrict__ pred_cls, const int* __restrict__ gt_cls, int n, int g,
                                                          const double* __restrict__ thr, int T, int num_classes,
                                                          uint8_t* __restrict__ matched) {
    const int type = blockIdx.x / T, k = blockIdx.x - type * T;
    const float* iou = type == 0 ? iou_box : iou_mask;
    uint8_t* out = matched + ((size_t)type * T + k) * n;
    const double th = thr[k];
    for (int c = threadIdx.x; c < num_classes; c += blockDim.x) {
        unsigned used[MAXG / 32];
#pragma unroll
        for (int w = 0; w < MAXG / 32; ++w) used[w] = 0u;
        for (int i = 0; i < n; ++i) {
            if (pred_cls[i] != c) continue;
            double best = th;
            int bj = -1;
            for (int j = 0; j < g; ++j) {
                if (gt_cls[j] != c || ((used[j >> 5] >> (j & 31)) & 1u)) continue;
                const double v = (double)iou[(size_t)i * g + j];
                if (v > best) { best = v; bj = j; }
            }
            if (bj >= 0) used[bj >> 5] |= 1u << (bj & 31);
            out[i] = bj >= 0 ? 1 : 0;
        }
    }
}

}  // namespace

// 256-pixel segments per chunk: the choice with the fewest (rounds over 256 CUs) x (work per chunk); ties go to the larger chunk
// (fewer partial counts for the finalize pass)
static int pick_segs(long long P) {
    int best = 4;
    long long best_cost = -1;
    for (int sg = 3; sg <= 5; ++sg) {
        const long long chunks = (P + sg * 256 - 1) / (sg * 256);
        const long long cost = ((chunks + 255) / 256) * sg;
        if (best_cost < 0 || cost <= best_cost) { best = sg; best_cost = cost; }
    }
    return best;
}

extern "C" size_t ym_mask_iou_workspace_bytes(int n, int g, int64_t P) {
    const int chunk = pick_segs(P) * 256;
    const size_t chunks = (size_t)((P + chunk - 1) / chunk);
    return chunks * ((size_t)n * g + n + g) * sizeof(int) + 256;
}

extern "C" int ym_mask_iou(const float* masks_a, int n, const float* masks_b, int g, int64_t P, float* iou, void* workspace,
                           size_t workspace_bytes, ym_stream_t s) {
    YM_REQUIRE(masks_a && masks_b && iou && workspace, "mask_iou: null pointer");
    YM_REQUIRE(n > 0 && g > 0 && P > 0 && P < (1ll << 24), "mask_iou: n, g > 0 and 0 < P < 2^24 (exact fp32 counts)");
    YM_REQUIRE((long long)n * g < (1ll << 24), "mask_iou: n*g too large");
    if (workspace_bytes < ym_mask_iou_workspace_bytes(n, g, P)) { ym_set_error("mask_iou: workspace too small"); return YM_ENOSPC; }
    hipStream_t st = (hipStream_t)s;
    const int segs = pick_segs(P), chunk = segs * 256;
    const int chunks = (int)((P + chunk - 1) / chunk);
    int* inter = (int*)workspace;
    int* area_a = inter + (size_t)chunks * n * g;
    int* area_b = area_a + (size_t)chunks * n;
    const dim3 grid((unsigned)chunks, (unsigned)((g + MAXR - 1) / MAXR));
    if (segs == 3) hipLaunchKernelGGL(k_mask_inter<3>, grid, dim3(512), 0, st, masks_a, n, masks_b, g, (long long)P, inter, area_a, area_b);
    else if (segs == 4) hipLaunchKernelGGL(k_mask_inter<4>, grid, dim3(512), 0, st, masks_a, n, masks_b, g, (long long)P, inter, area_a, area_b);
    else hipLaunchKernelGGL(k_mask_inter<5>, grid, dim3(512), 0, st, masks_a, n, masks_b, g, (long long)P, inter, area_a, area_b);
    hipLaunchKernelGGL(k_mask_iou_finalize, dim3((n * g + 63) / 64), dim3(64 * FSL), 0, st, inter, area_a, area_b, chunks, n, g, iou);
    return ym_check_launch("mask_iou");
}

extern "C" int ym_box_iou(const float* boxes_a, int n, const float* boxes_b, int g, float* iou, ym_stream_t s) {
    YM_REQUIRE(boxes_a && boxes_b && iou && n > 0 && g > 0, "box_iou: bad args");
    hipLaunchKernelGGL(k_box_iou, dim3((n * g + 255) / 256), dim3(256), 0, (hipStream_t)s, boxes_a, boxes_b, n, g, iou);
    return ym_check_launch("box_iou");
}

extern "C" int ym_match_detections(const float* iou_box, const float* iou_mask, const int32_t* pred_cls, const int32_t* gt_cls, int n,
                                   int g, const double* thresholds, int T, int num_classes, uint8_t* matched, ym_stream_t s) {
    YM_REQUIRE(iou_box && iou_mask && pred_cls && gt_cls && thresholds && matched, "match_detections: null pointer");
    YM_REQUIRE(n > 0 && g > 0 && g <= MAXG && T > 0 && num_classes > 0, "match_detections: need 0 < g <= %d", MAXG);
    hipLaunchKernelGGL(k_match_detections, dim3(2 * T), dim3(128), 0, (hipStream_t)s, iou_box, iou_mask, pred_cls, gt_cls, n, g,
                       thresholds, T, num_classes, matched);
    return ym_check_launch("match_detections");
}
