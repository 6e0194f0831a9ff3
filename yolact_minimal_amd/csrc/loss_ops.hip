// YOLACT training-loss bookkeeping on the device (reference modules/yolact.py:166-239,293-313, utils/box_utils.py:57-114):
// anchor matching, OHEM hard-negative selection + softmax cross-entropy, smooth-L1 box loss and the semantic-segmentation
// BCE, each emitting its gradient in the same pass (the losses are terminal nodes, so forward and backward fuse).
// These are latency/HBM-bound integer+float kernels over [B, 18525, *] tensors — no MFMA here.  All comparisons that
// decide a label or a selection are fp32 in the reference's operation order (no fma contraction).
#pragma clang fp contract(off)
#include <limits.h>
#include "ym_common.h"

namespace {

constexpr int NT = 1024;
constexpr int GMAX = 256;    // ground-truth boxes per image

__device__ __forceinline__ float iou_gt_prior(const float* g, float px1, float py1, float px2, float py2) {
    // box_iou(box_gt, decoded_priors): a = gt, b = prior (utils/box_utils.py:8-37)
    const float hx = fminf(g[2], px2), hy = fminf(g[3], py2);
    const float lx = fmaxf(g[0], px1), ly = fmaxf(g[1], py1);
    float w = hx - lx, h = hy - ly;
    w = w < 0.f ? 0.f : w;
    h = h < 0.f ? 0.f : h;
    const float inter = w * h;
    const float area_a = (g[2] - g[0]) * (g[3] - g[1]);
    const float area_b = (px2 - px1) * (py2 - py1);
    return __fdiv_rn(inter, (area_a + area_b) - inter);
}

// per-image arguments of the batched launches, passed by value (one workgroup row per image)
constexpr int MAXB = 32;
struct MatchBatch { const float* gt[MAXB]; int g[MAXB]; };
struct SemanticBatch { const float* ds[MAXB]; const int64_t* cls[MAXB]; int g[MAXB]; };

// ---- match() (utils/box_utils.py:57-83), workgroup = image ---------------------------------------------------------------
__global__ __launch_bounds__(NT) void k_match(MatchBatch mb, const float* __restrict__ anchors, int N,
                                              float pos_thre, float neg_thre, float* __restrict__ offsets,
                                              int64_t* __restrict__ conf, float* __restrict__ anchor_box,
                                              int64_t* __restrict__ anchor_gt, float* __restrict__ best_v) {
    const float* __restrict__ gt = mb.gt[blockIdx.x];
    const int g = mb.g[blockIdx.x];
    offsets += (size_t)blockIdx.x * N * 4;
    conf += (size_t)blockIdx.x * N;
    anchor_box += (size_t)blockIdx.x * N * 4;
    anchor_gt += (size_t)blockIdx.x * N;
    best_v += (size_t)blockIdx.x * N;
    __shared__ float s_gt[GMAX][5];
    __shared__ float s_rv[NT / 64];
    __shared__ int s_ri[NT / 64];
    __shared__ int s_gt_best[GMAX];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    for (int i = tid; i < g * 5; i += NT) s_gt[i / 5][i % 5] = gt[i];
    __syncthreads();
    // per anchor: best gt (first maximum), overlaps.max(0)
    for (int n = tid; n < N; n += NT) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(anchors + (size_t)n * 4);
        const float px1 = a[0] - a[2] / 2.f, py1 = a[1] - a[3] / 2.f, px2 = a[0] + a[2] / 2.f, py2 = a[1] + a[3] / 2.f;
        float bv = -INFINITY; int bj = 0;
        for (int j = 0; j < g; ++j) {
            const float v = iou_gt_prior(s_gt[j], px1, py1, px2, py2);
            if (v > bv || (v != v && bv == bv)) { bv = v; bj = j; }   // first max; NaN wins like torch.max
        }
        best_v[n] = bv;
        anchor_gt[n] = bj;
    }
    // per gt: best anchor (first maximum), overlaps.max(1)
    for (int j = 0; j < g; ++j) {
        float bv = -INFINITY; int bi = INT_MAX;
        for (int n = tid; n < N; n += NT) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(anchors + (size_t)n * 4);
            const float v = iou_gt_prior(s_gt[j], a[0] - a[2] / 2.f, a[1] - a[3] / 2.f, a[0] + a[2] / 2.f, a[1] + a[3] / 2.f);
            if (v > bv) { bv = v; bi = n; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(bv, o);
            const int oi = __shfl_xor(bi, o);
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        if (lane == 0) { s_rv[wv] = bv; s_ri[wv] = bi; }
        __syncthreads();
        if (tid == 0) {
            float v = s_rv[0]; int i0 = s_ri[0];
            for (int w = 1; w < NT / 64; ++w)
                if (s_rv[w] > v || (s_rv[w] == v && s_ri[w] < i0)) { v = s_rv[w]; i0 = s_ri[w]; }
            s_gt_best[j] = i0 == INT_MAX ? 0 : i0;
        }
        __syncthreads();
    }
    // each_anchor_max.index_fill_(0, gt_max_i, 2); then the sequential loop: the LAST gt wins a shared anchor (:72-73)
    if (tid == 0) {
        for (int j = 0; j < g; ++j) { best_v[s_gt_best[j]] = 2.f; anchor_gt[s_gt_best[j]] = j; }
    }
    __syncthreads();
    __threadfence_block();
    for (int n = tid; n < N; n += NT) {
        const int j = (int)anchor_gt[n];
        const float v = best_v[n];
        const float* m = s_gt[j];
        long long c = (long long)m[4] + 1;
        if (v < pos_thre) c = -1;
        if (v < neg_thre) c = 0;
        conf[n] = c;
        const f32x4 a = *reinterpret_cast<const f32x4*>(anchors + (size_t)n * 4);
        *reinterpret_cast<f32x4*>(anchor_box + (size_t)n * 4) = f32x4{m[0], m[1], m[2], m[3]};
        // encode (:104-114): ((g_c - a_c) / (0.1 * a_wh), log(g_wh / a_wh) / 0.2)
        f32x4 o;
        o[0] = __fdiv_rn((m[0] + m[2]) / 2.f - a[0], 0.1f * a[2]);
        o[1] = __fdiv_rn((m[1] + m[3]) / 2.f - a[1], 0.1f * a[3]);
        o[2] = __fdiv_rn(logf(__fdiv_rn(m[2] - m[0], a[2])), 0.2f);
        o[3] = __fdiv_rn(logf(__fdiv_rn(m[3] - m[1], a[3])), 0.2f);
        *reinterpret_cast<f32x4*>(offsets + (size_t)n * 4) = o;
    }
}

// ---- positives per image -------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void k_count_pos(const int64_t* __restrict__ conf, int N, int* __restrict__ num_pos) {
    __shared__ int s[NT / 64];
    const int b = blockIdx.x, tid = threadIdx.x;
    int c = 0;
    for (int n = tid; n < N; n += NT) c += conf[(size_t)b * N + n] > 0 ? 1 : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
    if ((tid & 63) == 0) s[tid >> 6] = c;
    __syncthreads();
    if (tid == 0) {
        int t = 0;
        for (int w = 0; w < NT / 64; ++w) t += s[w];
        num_pos[b] = t;
        atomicAdd(&num_pos[gridDim.x], t);        // total (slot B, zeroed by the caller)
    }
}

// ---- smooth-L1 box loss + gradient (modules/yolact.py:234-239) ----------------------------------------------------------
__global__ __launch_bounds__(256) void k_box_loss(const float* __restrict__ box_p, const float* __restrict__ offsets,
                                                   const int64_t* __restrict__ conf, long long total, const int* __restrict__ num_pos,
                                                   int B, float alpha, float* __restrict__ dbox, double* __restrict__ loss) {
    const float coeff = alpha / (float)num_pos[B];
    double acc = 0.0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        f32x4 g = {0.f, 0.f, 0.f, 0.f};
        if (conf[i] > 0) {
            const f32x4 p = *reinterpret_cast<const f32x4*>(box_p + i * 4), t = *reinterpret_cast<const f32x4*>(offsets + i * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float d = p[e] - t[e], ad = fabsf(d);
                acc += (double)(ad < 1.f ? 0.5f * d * d : ad - 0.5f);
                g[e] = (ad < 1.f ? d : (d > 0.f ? 1.f : -1.f)) * coeff;
            }
        }
        *reinterpret_cast<f32x4*>(dbox + i * 4) = g;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if ((threadIdx.x & 63) == 0 && acc != 0.0) atomicAdd(loss, acc * (double)coeff);
}

// ---- category loss: global max, OHEM marks, per-image rank threshold, CE + gradient (:205-232) -------------------------
__global__ __launch_bounds__(256) void k_global_max(const float* __restrict__ x, long long n, float* __restrict__ part) {
    float m = -INFINITY;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) m = fmaxf(m, x[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    __shared__ float s[4];
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3]));
}

// one wave per anchor row: mark = log(sum exp(x - gmax)) + gmax - x[0]; 0 for positives / neutrals
__global__ __launch_bounds__(256) void k_ohem_mark(const float* __restrict__ cls, const int64_t* __restrict__ conf, long long rows,
                                                    int C, const float* __restrict__ gmax_part, int nparts, float* __restrict__ mark) {
    __shared__ float s_gmax, s_w[4];
    {                                                   // max of the partial maxima, by the whole workgroup
        float m = -INFINITY;
        for (int i = threadIdx.x; i < nparts; i += 256) m = fmaxf(m, gmax_part[i]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = m;
        __syncthreads();
        if (threadIdx.x == 0) s_gmax = fmaxf(fmaxf(s_w[0], s_w[1]), fmaxf(s_w[2], s_w[3]));
    }
    __syncthreads();
    const float gmax = s_gmax;
    const int lane = threadIdx.x & 63;
    const long long w0 = ((long long)blockIdx.x * 256 + threadIdx.x) >> 6, nw = ((long long)gridDim.x * 256) >> 6;
    for (long long r = w0; r < rows; r += nw) {
        const float* x = cls + r * C;
        float s = 0.f;
        for (int c = lane; c < C; c += 64) s += expf(x[c] - gmax);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        if (lane == 0) mark[r] = conf[r] != 0 ? 0.f : (logf(s) + gmax) - x[0];
    }
}

// per image: rank threshold of the top-`num_neg` marks (descending; equal marks by index), then neg = rank < num_neg & bg
__global__ __launch_bounds__(NT) void k_ohem_select(const float* __restrict__ mark, const int64_t* __restrict__ conf, int N,
                                                    const int* __restrict__ num_pos, int ratio, uint8_t* __restrict__ sel) {
    __shared__ uint32_t hist[256];
    __shared__ int sh_digit, sh_rem, wave_tot[NT / 64], running;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const float* mk = mark + (size_t)b * N;
    const int64_t* cf = conf + (size_t)b * N;
    uint8_t* out = sel + (size_t)b * N;
    int R = ratio * num_pos[b];
    if (R > N - 1) R = N - 1;
    auto key = [&](int i) { return __float_as_uint(mk[i]) ^ 0x80000000u; };      // marks are >= 0 (or NaN)
    if (R <= 0) {
        for (int i = tid; i < N; i += NT) out[i] = cf[i] > 0 ? 1 : 0;
        return;
    }
    uint32_t prefix = 0u, mask = 0u;
    int remaining = R;
    for (int shift = 24; shift >= 0; shift -= 8) {
        for (int i = tid; i < 256; i += NT) hist[i] = 0u;
        __syncthreads();
        for (int i = tid; i < N; i += NT) {
            const uint32_t k = key(i);
            if ((k & mask) == prefix) atomicAdd(&hist[(k >> shift) & 255u], 1u);
        }
        __syncthreads();
        if (tid == 0) {
            int acc = 0, d = 255;
            for (; d > 0; --d) {
                const int h = (int)hist[d];
                if (acc + h >= remaining) break;
                acc += h;
            }
            sh_digit = d; sh_rem = remaining - acc;
        }
        __syncthreads();
        prefix |= (uint32_t)sh_digit << shift;
        mask |= 0xFFu << shift;
        remaining = sh_rem;
        __syncthreads();
    }
    const uint32_t T = prefix;
    const int r_eq = remaining;
    if (tid == 0) running = 0;
    __syncthreads();
    for (int base = 0; base < N; base += NT) {
        const int i = base + tid;
        const uint32_t k = i < N ? key(i) : 0u;
        const bool eq = i < N && k == T;
        const unsigned long long bal = __ballot(eq);
        const int pre = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wave_tot[wv] = __popcll(bal);
        __syncthreads();
        int off = running;
        for (int w = 0; w < wv; ++w) off += wave_tot[w];
        if (i < N) {
            const bool neg = (k > T || (eq && off + pre < r_eq)) && cf[i] == 0;
            out[i] = (cf[i] > 0 || neg) ? 1 : 0;
        }
        __syncthreads();
        if (tid == 0) { int t = 0; for (int w = 0; w < NT / 64; ++w) t += wave_tot[w]; running += t; }
        __syncthreads();
    }
}

// Positives of one image for the mask loss (modules/yolact.py:255-267): all of them in anchor order when there are at most `cap`
// (masks_to_train), else a uniformly random subset of exactly `cap` — the `cap` largest of the caller's iid uniform keys among the
// positives (radix select, equal keys by anchor index), written in anchor order.  Workgroup = image.  idx[b][0 .. min(P, cap)) are
// valid; the rest is never read (the mask-loss kernel takes the counts from num_pos).  Replaces a masked_fill + topk chain of six
// ATen launches per step.
__global__ __launch_bounds__(NT) void k_select_positives(const int64_t* __restrict__ conf, const float* __restrict__ keys, int N,
                                                         int cap, const int* __restrict__ num_pos, int64_t* __restrict__ idx) {
    __shared__ uint32_t hist[256];
    __shared__ int sh_digit, sh_rem, wave_tot[NT / 64], wave_eq[NT / 64], running, run_eq;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int64_t* cf = conf + (size_t)b * N;
    const float* ky = keys + (size_t)b * N;
    int64_t* out = idx + (size_t)b * cap;
    const int P = num_pos[b];
    uint32_t T = 0u;
    int r_eq = 0;
    const bool all = P <= cap;
    auto key = [&](int i) { return __float_as_uint(ky[i]); };               // keys are in [0, 1): the bit pattern orders them
    if (!all) {
        uint32_t prefix = 0u, mask = 0u;
        int remaining = cap;
        for (int shift = 24; shift >= 0; shift -= 8) {
            for (int i = tid; i < 256; i += NT) hist[i] = 0u;
            __syncthreads();
            for (int i = tid; i < N; i += NT) {
                if (cf[i] <= 0) continue;
                const uint32_t k = key(i);
                if ((k & mask) == prefix) atomicAdd(&hist[(k >> shift) & 255u], 1u);
            }
            __syncthreads();
            if (tid == 0) {
                int acc = 0, d = 255;
                for (; d > 0; --d) {
                    const int h = (int)hist[d];
                    if (acc + h >= remaining) break;
                    acc += h;
                }
                sh_digit = d; sh_rem = remaining - acc;
            }
            __syncthreads();
            prefix |= (uint32_t)sh_digit << shift;
            mask |= 0xFFu << shift;
            remaining = sh_rem;
            __syncthreads();
        }
        T = prefix;
        r_eq = remaining;                                                    // how many keys equal to T are taken (lowest anchors)
    }
    if (tid == 0) { running = 0; run_eq = 0; }
    __syncthreads();
    for (int base = 0; base < N; base += NT) {
        const int i = base + tid;
        const bool p = i < N && cf[i] > 0;
        const uint32_t k = p ? key(i) : 0u;
        const bool eq = p && !all && k == T;
        const unsigned long long bal_eq = __ballot(eq);
        if (lane == 0) wave_eq[wv] = __popcll(bal_eq);
        __syncthreads();
        int off_eq = run_eq;
        for (int w = 0; w < wv; ++w) off_eq += wave_eq[w];
        const bool take = p && (all || k > T || (eq && off_eq + __popcll(bal_eq & ((1ull << lane) - 1ull)) < r_eq));
        const unsigned long long bal = __ballot(take);
        if (lane == 0) wave_tot[wv] = __popcll(bal);
        __syncthreads();
        int off = running;
        for (int w = 0; w < wv; ++w) off += wave_tot[w];
        if (take) {
            const int slot = off + __popcll(bal & ((1ull << lane) - 1ull));
            if (slot < cap) out[slot] = i;
        }
        __syncthreads();                                                      // everyone has read running / run_eq and both count arrays
        if (tid == 0) {
            int t = 0, e = 0;
            for (int w = 0; w < NT / 64; ++w) { t += wave_tot[w]; e += wave_eq[w]; }
            running += t;
            run_eq += e;
        }
        __syncthreads();
    }
}

// cross entropy over the selected rows (sum) / total positives, and its gradient; one wave per row
__global__ __launch_bounds__(256) void k_ce_loss(const float* __restrict__ cls, const int64_t* __restrict__ conf,
                                                  const uint8_t* __restrict__ sel, long long rows, int C, const int* __restrict__ num_pos,
                                                  int B, float alpha, float* __restrict__ dcls, double* __restrict__ loss) {
    const float coeff = alpha / (float)num_pos[B];
    const int lane = threadIdx.x & 63;
    const long long w0 = ((long long)blockIdx.x * 256 + threadIdx.x) >> 6, nw = ((long long)gridDim.x * 256) >> 6;
    double acc = 0.0;
    for (long long r = w0; r < rows; r += nw) {
        const float* x = cls + r * C;
        float* dx = dcls + r * C;
        if (!sel[r]) {
            for (int c = lane; c < C; c += 64) dx[c] = 0.f;
            continue;
        }
        float v[4], mx = -INFINITY;
#pragma unroll
        for (int j = 0; j < 4; ++j) { const int c = lane + 64 * j; v[j] = c < C ? x[c] : -INFINITY; mx = fmaxf(mx, v[j]); }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) { const int c = lane + 64 * j; v[j] = c < C ? expf(v[j] - mx) : 0.f; s += v[j]; }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        const int t = (int)conf[r];
        if (lane == 0) acc += (double)((logf(s) + mx) - x[t]);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = lane + 64 * j;
            if (c < C) dx[c] = (v[j] / s - (c == t ? 1.f : 0.f)) * coeff;
        }
    }
    if (lane == 0 && acc != 0.0) atomicAdd(loss, acc * (double)coeff);
}

// ---- semantic segmentation loss (:293-313), thread = (pixel, channel), blockIdx.y = image -----------------------------------
// seg: NHWC [B][P][pitch] logits (first nc channels real), per image ds: [g][P] binarised down-sampled gt masks, cls: [g] class ids.
__global__ __launch_bounds__(256) void k_semantic_loss(const float* __restrict__ seg, int P, int pitch, int nc, SemanticBatch sb,
                                                        int gt_stride, float coeff, float* __restrict__ dseg,
                                                        double* __restrict__ loss) {
    const int img = blockIdx.y, g = sb.g[img];
    const float* __restrict__ ds = sb.ds[img];
    const int64_t* __restrict__ cls = sb.cls[img];
    const unsigned total = (unsigned)P * (unsigned)pitch;        // < 2^31 (checked by the caller): 32-bit index arithmetic
    const float* x = seg + (size_t)img * total;
    float* dx = dseg + (size_t)img * total;
    double acc = 0.0;
    for (unsigned e = blockIdx.x * 256u + threadIdx.x; e < total; e += gridDim.x * 256u) {
        const unsigned pix = e / (unsigned)pitch;
        const int c = (int)(e - pix * (unsigned)pitch);
        float gval = 0.f;
        if (c < nc) {
            float t = 0.f;                                        // target = max over the gts of class c at this pixel
            for (int j = 0; j < g; ++j)
                if ((int)cls[(size_t)j * gt_stride] == c && ds[(size_t)j * P + pix] > 0.f) t = 1.f;
            const float v = x[e];
            acc += (double)(fmaxf(v, 0.f) - v * t + log1pf(expf(-fabsf(v))));
            gval = (1.f / (1.f + expf(-v)) - t) * coeff;
        }
        dx[e] = gval;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    __shared__ double s_acc[4];
    if ((threadIdx.x & 63) == 0) s_acc[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        const double t = (s_acc[0] + s_acc[1]) + (s_acc[2] + s_acc[3]);
        if (t != 0.0) atomicAdd(loss, t * (double)coeff);
    }
}

}  // namespace

extern "C" int ym_match_anchors_batch(const float* const* gt_boxes_cls, const int32_t* g, int B, const float* anchors, int N,
                                      float pos_thre, float neg_thre, float* offsets, int64_t* conf, float* anchor_box,
                                      int64_t* anchor_gt, void* workspace, size_t workspace_bytes, ym_stream_t s) {
    YM_REQUIRE(gt_boxes_cls && g && anchors && offsets && conf && anchor_box && anchor_gt && workspace, "match_anchors: null pointer");
    YM_REQUIRE(B > 0 && N > 0, "match_anchors: need B > 0, N > 0");
    for (int i = 0; i < B; ++i) YM_REQUIRE(gt_boxes_cls[i] && g[i] >= 1 && g[i] <= GMAX, "match_anchors: 1 <= g <= %d", GMAX);
    if (workspace_bytes < (size_t)B * N * 4) { ym_set_error("match_anchors: workspace < 4*B*N bytes"); return YM_ENOSPC; }
    for (int b0 = 0; b0 < B; b0 += MAXB) {
        const int nb = B - b0 < MAXB ? B - b0 : MAXB;
        MatchBatch mb;
        for (int i = 0; i < nb; ++i) { mb.gt[i] = gt_boxes_cls[b0 + i]; mb.g[i] = g[b0 + i]; }
        hipLaunchKernelGGL(k_match, dim3(nb), dim3(NT), 0, (hipStream_t)s, mb, anchors, N, pos_thre, neg_thre,
                           offsets + (size_t)b0 * N * 4, conf + (size_t)b0 * N, anchor_box + (size_t)b0 * N * 4,
                           anchor_gt + (size_t)b0 * N, (float*)workspace + (size_t)b0 * N);
    }
    return ym_check_launch("match_anchors");
}

extern "C" int ym_match_anchors(const float* gt_boxes_cls, int g, const float* anchors, int N, float pos_thre, float neg_thre,
                                float* offsets, int64_t* conf, float* anchor_box, int64_t* anchor_gt, void* workspace,
                                size_t workspace_bytes, ym_stream_t s) {
    const int32_t g1 = g;
    return ym_match_anchors_batch(&gt_boxes_cls, &g1, 1, anchors, N, pos_thre, neg_thre, offsets, conf, anchor_box, anchor_gt,
                                  workspace, workspace_bytes, s);
}

extern "C" size_t ym_loss_workspace_bytes(int B, int N) {
    return (size_t)B * N * 4 /*mark*/ + (size_t)B * N /*sel*/ + 1024 * 4 /*gmax parts*/ + 1024;
}

// class + box losses for a batch.  num_pos: int32 [B+1] (per image, total) is produced here (device side, no host sync).
extern "C" int ym_class_box_loss(const float* class_p, const float* box_p, const float* offsets, const int64_t* conf, int B, int N,
                                 int C, float conf_alpha, float bbox_alpha, int neg_pos_ratio, float* dclass, float* dbox,
                                 int32_t* num_pos, double* loss_c, double* loss_b, void* workspace, size_t workspace_bytes,
                                 ym_stream_t s) {
    YM_REQUIRE(class_p && box_p && offsets && conf && dclass && dbox && num_pos && loss_c && loss_b && workspace, "class_box_loss: null pointer");
    YM_REQUIRE(B > 0 && N > 0 && C > 1 && C <= 256, "class_box_loss: bad shape");
    if (workspace_bytes < ym_loss_workspace_bytes(B, N)) { ym_set_error("class_box_loss: workspace too small"); return YM_ENOSPC; }
    hipStream_t st = (hipStream_t)s;
    char* w = (char*)workspace;
    float* mark = (float*)w; w += (size_t)B * N * 4;
    float* gpart = (float*)w; w += 1024 * 4;
    uint8_t* sel = (uint8_t*)w;
    const long long rows = (long long)B * N;
    (void)hipMemsetAsync(num_pos, 0, (size_t)(B + 1) * 4, st);
    (void)hipMemsetAsync(loss_c, 0, 8, st);
    (void)hipMemsetAsync(loss_b, 0, 8, st);
    hipLaunchKernelGGL(k_count_pos, dim3(B), dim3(NT), 0, st, conf, N, num_pos);
    const int nparts = 512;
    hipLaunchKernelGGL(k_global_max, dim3(nparts), dim3(256), 0, st, class_p, rows * C, gpart);
    int wgrid = (int)((rows + 3) / 4);
    if (wgrid > 8192) wgrid = 8192;
    hipLaunchKernelGGL(k_ohem_mark, dim3(wgrid), dim3(256), 0, st, class_p, conf, rows, C, gpart, nparts, mark);
    hipLaunchKernelGGL(k_ohem_select, dim3(B), dim3(NT), 0, st, mark, conf, N, num_pos, neg_pos_ratio, sel);
    hipLaunchKernelGGL(k_ce_loss, dim3(wgrid), dim3(256), 0, st, class_p, conf, sel, rows, C, num_pos, B, conf_alpha, dclass, loss_c);
    int bgrid = (int)((rows + 255) / 256);
    if (bgrid > 2048) bgrid = 2048;
    hipLaunchKernelGGL(k_box_loss, dim3(bgrid), dim3(256), 0, st, box_p, offsets, conf, rows, num_pos, B, bbox_alpha, dbox, loss_b);
    return ym_check_launch("class_box_loss");
}

extern "C" int ym_select_positives(const int64_t* conf, const float* keys, int B, int N, int cap, const int32_t* num_pos, int64_t* idx,
                                   ym_stream_t s) {
    YM_REQUIRE(conf && keys && num_pos && idx && B > 0 && N > 0 && cap > 0, "select_positives: bad args");
    hipLaunchKernelGGL(k_select_positives, dim3(B), dim3(NT), 0, (hipStream_t)s, conf, keys, N, cap, num_pos, idx);
    return ym_check_launch("select_positives");
}

extern "C" int ym_semantic_loss_batch(const float* seg_nhwc, int B, int P, int pitch, int num_classes,
                                      const float* const* gt_masks_ds, const int64_t* const* gt_cls, int gt_cls_stride,
                                      const int32_t* g, float coeff, float* dseg, double* loss_accum, ym_stream_t s) {
    YM_REQUIRE(seg_nhwc && dseg && loss_accum && g && B > 0 && P > 0 && pitch >= num_classes && num_classes <= 256,
               "semantic_loss: bad args");
    for (int i = 0; i < B; ++i) YM_REQUIRE(g[i] == 0 || (gt_masks_ds && gt_cls && gt_masks_ds[i] && gt_cls[i]), "semantic_loss: null gt");
    YM_REQUIRE((long long)P * pitch < (1ll << 31), "semantic_loss: P * pitch must stay below 2^31");
    const size_t total = (size_t)P * pitch;
    int grid = (int)((total + 255) / 256);
    if (grid > 96) grid = 96;                // one fp64 atomic per workgroup on ONE address: few workgroups, each looping
    for (int b0 = 0; b0 < B; b0 += MAXB) {
        const int nb = B - b0 < MAXB ? B - b0 : MAXB;
        SemanticBatch sb;
        for (int i = 0; i < nb; ++i) {
            sb.g[i] = g[b0 + i];
            sb.ds[i] = sb.g[i] ? gt_masks_ds[b0 + i] : nullptr;
            sb.cls[i] = sb.g[i] ? gt_cls[b0 + i] : nullptr;
        }
        hipLaunchKernelGGL(k_semantic_loss, dim3(grid, nb), dim3(256), 0, (hipStream_t)s, seg_nhwc + (size_t)b0 * total, P, pitch,
                           num_classes, sb, gt_cls_stride, coeff, dseg + (size_t)b0 * total, loss_accum);
    }
    return ym_check_launch("semantic_loss");
}

extern "C" int ym_semantic_loss(const float* seg_nhwc, int P, int pitch, int num_classes, const float* gt_masks_ds,
                                const int64_t* gt_cls, int gt_cls_stride, int g, float coeff, float* dseg, double* loss_accum,
                                ym_stream_t s) {
    const int32_t g1 = g;
    return ym_semantic_loss_batch(seg_nhwc, 1, P, pitch, num_classes, &gt_masks_ds, &gt_cls, gt_cls_stride, &g1, coeff, dseg,
                                  loss_accum, s);
}
