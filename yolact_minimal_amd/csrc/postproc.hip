// Detection post-processing for gfx950: score filter + ordered compaction + box decode, Fast-NMS
// (per-class radix-select top-k, wavefront-reduction IoU column test, global top-100), greedy
// per-class NMS (the cython_nms replacement), MFMA mask assembly with fused sigmoid+crop, and the
// bilinear resize + binarise that produces the final masks.
//
// Bit-exactness: every comparison that decides an index (score > thre, IoU <= thre, ordering) is
// evaluated in fp32 with the reference's operation order, IEEE division and NO fma contraction —
// hence the pragma below.  Sort ties (torch.sort is unstable for n > 16, so the reference's tie order is
// implementation-defined) are resolved here as "lower index first" (= a stable descending sort).
#pragma clang fp contract(off)
#include <limits.h>
#include <type_traits>
#include "ym_common.h"

namespace {

constexpr int NT = 1024;      // threads of the selection kernels
// phase stamps of the single-workgroup kernels (debug build `make trace` only; tools/nms_stamps.py): s_memrealtime (100 MHz) of
// thread 0 into the workspace's counter block, 64-bit slot `i` behind the 8 integer counters
#ifdef YM_TRACE
#define YM_NMS_STAMP(cnt, i) do { if (threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0) reinterpret_cast<long long*>((cnt) + 8)[i] = (long long)__builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define YM_NMS_STAMP(cnt, i) do { } while (0)
#endif
constexpr int TOPK_CAP = 256; // >= cfg.top_k
constexpr int DET_CAP = 128;  // >= cfg.max_detections

__device__ __forceinline__ uint32_t f2key(float f) {
    const uint32_t u = __float_as_uint(f);
    return u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
__device__ __forceinline__ float key2f(uint32_t k) {
    return __uint_as_float(k ^ ((k >> 31) ? 0x80000000u : 0xFFFFFFFFu));
}

// Maximum of an unsigned value over the 64 lanes of a wave, returned to every lane: inclusive max-scan with DPP (row_shr 1 / 2 / 4 / 8
// inside each row of 16 lanes, row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3; lanes without a source keep the
// identity 0), lane 63 then holds the total.  ~14 VALU instructions; a ds_bpermute butterfly is 12 LDS-crossbar round trips.
__device__ __forceinline__ uint32_t wave_umax(uint32_t x) {
    int v = (int)x;
#define YM_DPP_MAX(ctrl, rmask) v = (int)max((uint32_t)v, (uint32_t)__builtin_amdgcn_update_dpp(0, v, ctrl, rmask, 0xf, false))
    YM_DPP_MAX(0x111, 0xf);      // row_shr:1
    YM_DPP_MAX(0x112, 0xf);      // row_shr:2
    YM_DPP_MAX(0x114, 0xf);      // row_shr:4
    YM_DPP_MAX(0x118, 0xf);      // row_shr:8
    YM_DPP_MAX(0x142, 0xa);      // row_bcast:15 -> rows 1, 3
    YM_DPP_MAX(0x143, 0xc);      // row_bcast:31 -> rows 2, 3
#undef YM_DPP_MAX
    return (uint32_t)__builtin_amdgcn_readlane(v, 63);
}

// Inclusive prefix sum over the 64 lanes of a wave by the same DPP steps (Hillis-Steele inside the rows of 16, then the row totals).
__device__ __forceinline__ int wave_incl_scan_add(int x) {
    int v = x;
#define YM_DPP_ADD(ctrl, rmask) v += __builtin_amdgcn_update_dpp(0, v, ctrl, rmask, 0xf, false)
    YM_DPP_ADD(0x111, 0xf);      // row_shr:1
    YM_DPP_ADD(0x112, 0xf);      // row_shr:2
    YM_DPP_ADD(0x114, 0xf);      // row_shr:4
    YM_DPP_ADD(0x118, 0xf);      // row_shr:8
    YM_DPP_ADD(0x142, 0xa);      // row_bcast:15 -> rows 1, 3
    YM_DPP_ADD(0x143, 0xc);      // row_bcast:31 -> rows 2, 3
#undef YM_DPP_ADD
    return v;
}

// utils/box_utils.py:8-37, one pair: inter / (area_a + area_b - inter); 0/0 -> NaN like torch.
__device__ __forceinline__ float iou_pair(const f32x4 a, const f32x4 b) {
    const float hx = fminf(a[2], b[2]), hy = fminf(a[3], b[3]);
    const float lx = fmaxf(a[0], b[0]), ly = fmaxf(a[1], b[1]);
    float w = hx - lx, h = hy - ly;
    w = w < 0.f ? 0.f : w;
    h = h < 0.f ? 0.f : h;
    const float inter = w * h;
    const float area_a = (a[2] - a[0]) * (a[3] - a[1]);
    const float area_b = (b[2] - b[0]) * (b[3] - b[1]);
    return __fdiv_rn(inter, (area_a + area_b) - inter);
}

// ---------------------------------------------------------------------------------------------------
// block-level "top R of L, sorted" (radix select on order-preserving keys + bitonic sort in LDS)
// ---------------------------------------------------------------------------------------------------
template <int CAP>
struct TopkShared {
    uint32_t hist[4][256];      // one histogram per radix pass, all zeroed up front: a pass costs two barriers instead of four
    uint32_t keys[CAP];
    int idx[CAP];
    int wave_tot[NT / 64];
    int sel_digit[4], sel_remaining[4], sel_eq_total[4];   // per pass: nothing of pass p is overwritten while a wave may still read it
    int cnt_gt, cnt_eq, running;
};

// (Measured: LDS atomics on a hot bin are NOT the bottleneck here — a ballot-aggregated histogram was 1.5x slower.
//  The serial 256-bin walk by one thread was: ~8 us per radix pass of dependent LDS reads -> done by one wave below.)
constexpr int KCACHE = 24;   // keys cached per thread: lists up to KCACHE*NT = 24576 entries are read from HBM once

template <int CAP, typename KeyAt>
__device__ int block_topk_sorted(KeyAt key_at, int L, int R, TopkShared<CAP>& sh, int* dbg = nullptr) {
    const int tid = threadIdx.x, nt = blockDim.x;
    for (int i = tid; i < CAP; i += nt) { sh.keys[i] = 0u; sh.idx[i] = INT_MAX - i; }      // (empty slots: distinct indices, so that the sort below has no ties)
    for (int i = tid; i < 4 * 256; i += nt) (&sh.hist[0][0])[i] = 0u;
    if (tid == 0) { sh.cnt_gt = 0; sh.cnt_eq = 0; sh.running = 0; }
    __syncthreads();
    int cnt;
    if (L <= R) {
        for (int i = tid; i < L; i += nt) { sh.keys[i] = key_at(i); sh.idx[i] = i; }
        cnt = L;
    } else {
        // The radix passes re-scan the list 5+ times; every scan used to be a round of dependent global loads
        // (latency-bound: ~20 rounds x 5 passes).  Read the keys ONCE into registers when the list fits.
        const bool cached = L <= KCACHE * nt;
        uint32_t ck[KCACHE];
        if (cached) {
#pragma unroll
            for (int j = 0; j < KCACHE; ++j) {
                const int i = tid + j * nt;
                ck[j] = i < L ? key_at(i) : 0u;
            }
        }
        uint32_t prefix = 0u, mask = 0u;
        int remaining = R;
        int eq_total_last = 0;
        // (walking only the key bits that differ inside the list -- scores of one class share their top 6 -- so that no pass piles every
        //  key onto 1-3 bins was built and measured: 35.5 -> 35.3 us per launch.  The select's time is not in the atomics: of its
        //  27 us, 6 are the key loads, 9 the four passes, 4 the collection, 8 the rank-by-counting sort: tools/nms_stamps.py.)
        for (int pass = 0; pass < 4; ++pass) {
            const int shift = 24 - 8 * pass;
            const uint32_t dmask = 255u;
            uint32_t* hist = sh.hist[pass];
            if (cached) {
#pragma unroll
                for (int j = 0; j < KCACHE; ++j) {
                    const int i = tid + j * nt;
                    if (i < L && ck[j] != 0u && (ck[j] & mask) == prefix) atomicAdd(&hist[(ck[j] >> shift) & dmask], 1u);
                }
            } else {
                for (int base = 0; base < L; base += nt) {
                    const int i = base + tid;
                    const uint32_t k = i < L ? key_at(i) : 0u;
                    if (i < L && k != 0u && (k & mask) == prefix) atomicAdd(&hist[(k >> shift) & dmask], 1u);
                }
            }
            __syncthreads();
            if (tid < 64) {
                // key 0 marks an absent entry and is never counted: with fewer than R real keys the walk ends at
                // digit 0 and T becomes 0, i.e. "take every real key" (the callers drop key-0 slots afterwards).
                // Walk the 256 bins from the top with ONE WAVE: lane l owns digits 255-4l .. 252-4l.
                int c[4], lane_total = 0;
#pragma unroll
                for (int j = 0; j < 4; ++j) { c[j] = (int)hist[255 - (4 * tid + j)]; lane_total += c[j]; }
                const int incl = wave_incl_scan_add(lane_total);      // (tid < 64: all lanes of wave 0 are active)
                int run = incl - lane_total, found_r = -1, acc_before = 0;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (found_r < 0 && run + c[j] >= remaining) { found_r = 4 * tid + j; acc_before = run; }
                    run += c[j];
                }
                const unsigned long long hit = __ballot(found_r >= 0);
                const int total = __builtin_amdgcn_readlane(incl, 63);
                if (hit) {
                    const int leader = __ffsll((long long)hit) - 1;
                    if (tid == leader) {
                        int d = 255 - found_r, acc = acc_before;
                        if (d == 0) { d = 0; }      // reached the last bin normally
                        sh.sel_digit[pass] = d;
                        sh.sel_remaining[pass] = remaining - acc;
                        sh.sel_eq_total[pass] = (int)hist[d];
                    }
                } else if (tid == 0) {
                    sh.sel_digit[pass] = 0;                            // fewer real keys than requested
                    sh.sel_remaining[pass] = remaining - (total - (int)hist[0]);
                    sh.sel_eq_total[pass] = (int)hist[0];
                }
            }
            __syncthreads();
            prefix |= (uint32_t)sh.sel_digit[pass] << shift;
            mask |= dmask << shift;
            remaining = sh.sel_remaining[pass];
            eq_total_last = sh.sel_eq_total[pass];
        }
        if (dbg) YM_NMS_STAMP(dbg, 13);
        const uint32_t T = prefix;
        const int r_eq = remaining, n_gt = R - r_eq, eq_total = eq_total_last;
        auto collect = [&](uint32_t k, int i) {
            if (k > T) {
                const int p = atomicAdd(&sh.cnt_gt, 1);
                sh.keys[p] = k; sh.idx[p] = i;
            } else if (k == T && eq_total == r_eq && T != 0u) {
                const int p = atomicAdd(&sh.cnt_eq, 1);
                sh.keys[n_gt + p] = k; sh.idx[n_gt + p] = i;
            }
        };
        if (cached) {
#pragma unroll
            for (int j = 0; j < KCACHE; ++j) {
                const int i = tid + j * nt;
                if (i < L) collect(ck[j], i);
            }
        } else {
            for (int i = tid; i < L; i += nt) collect(key_at(i), i);
        }
        if (eq_total != r_eq && T != 0u) {
            // more ties at the cut than slots: take the r_eq LOWEST indices (ordered block scan); T == 0 means fewer
            // than R real keys exist — the remaining slots stay empty (key 0)
            const int lane = tid & 63, wv = tid >> 6;
            for (int base = 0; base < L; base += nt) {
                const int i = base + tid;
                const bool f = (i < L) && key_at(i) == T;
                const unsigned long long bal = __ballot(f);
                const int pre = __popcll(bal & ((1ull << lane) - 1ull));
                if (lane == 0) sh.wave_tot[wv] = __popcll(bal);
                __syncthreads();
                int off = sh.running;
                for (int w = 0; w < wv; ++w) off += sh.wave_tot[w];
                if (f && off + pre < r_eq) { sh.keys[n_gt + off + pre] = T; sh.idx[n_gt + off + pre] = i; }
                __syncthreads();
                if (tid == 0) { int t = 0; for (int w = 0; w < nt / 64; ++w) t += sh.wave_tot[w]; sh.running += t; }
                __syncthreads();
                if (sh.running >= r_eq) break;
            }
        }
        cnt = R;
    }
    __syncthreads();
    if (dbg) YM_NMS_STAMP(dbg, 14);
    // Sort the CAP slots (key descending, index ascending; empty slots = key 0 last) by counting: NT / CAP threads share one
    // element, each ranks it against a slice of the list (broadcast LDS reads), the slice counts meet in LDS, one scatter.
    // Two barriers instead of the 28-36 of a bitonic network (a barrier of a 1024-thread workgroup costs ~1 us).
    static_assert(NT % CAP == 0, "NT must be a multiple of CAP");
    constexpr int SL = NT / CAP, SPAN = CAP / SL;
    const int e = tid % CAP, sl = tid / CAP;
    const uint32_t ka = sh.keys[e];
    const int ia = sh.idx[e];
    // (key, ~index) packed into 64 bits: "b sorts before a" is ONE unsigned compare (indices are distinct, real and empty slots
    // alike), i.e. ds_read_b64 + v_cmp + add-with-carry per pair instead of two reads and a four-term predicate: the 65 k pair
    // tests of this sort are VALU-bound (16 waves on 4 SIMDs), 8.2 us of the class kernel's 35.  hist[2..3] are free by now.
    static_assert(CAP <= 256, "the packed sort keys live in two of the histograms");
    unsigned long long* pk = reinterpret_cast<unsigned long long*>(&sh.hist[2][0]);
    const unsigned long long pa = ((unsigned long long)ka << 32) | (unsigned)~ia;
    if (sl == 0) pk[e] = pa;
    __syncthreads();
    int before = 0;
    if (tid < NT) {
#pragma unroll 8
        for (int x = sl * SPAN; x < (sl + 1) * SPAN; ++x) before += pk[x] > pa ? 1 : 0;
    }
    if (tid < CAP) sh.hist[0][tid] = 0u;                        // hist[0] (256 >= CAP entries) is free now: rank accumulators
    __syncthreads();
    atomicAdd(&sh.hist[0][e], (uint32_t)before);
    __syncthreads();
    if (sl == 0) {
        const int r = (int)sh.hist[0][e];
        sh.keys[r] = ka;
        sh.idx[r] = ia;
    }
    __syncthreads();
    return cnt;
}

// ---------------------------------------------------------------------------------------------------
// workspace
// ---------------------------------------------------------------------------------------------------
struct NmsWs {
    int* counters;       // [0] = K (anchors over threshold)  [1] = scratch
    uint8_t* flag;       // [N]
    int* keep_idx;       // [N]   compacted anchor indices, ascending
    float* boxes_k;      // [N][4] decoded + clipped boxes of the kept anchors
    float* scores_t;     // [C-1][N] class-major scores of the kept anchors (background dropped)
    int* top_idx;        // [C-1][TOPK_CAP] index into the compacted list
    float* top_score;    // [C-1][TOPK_CAP]
    int* top_cnt;        // [C-1]     entries of top_idx / top_score that SURVIVED the IoU test (compacted, still sorted)
    int* chunk_cnt;      // [ceil(N / CHUNK)] anchors over threshold per chunk of CHUNK consecutive anchors
    size_t bytes;
};
constexpr int CHUNK = 64;    // anchors per workgroup of stage A

inline size_t align_up(size_t x) { return (x + 255) & ~(size_t)255; }

// Batched launches (ym_detect_fast_nms_batch): blockIdx.y = image; image b owns the workspace carved at base + b * stride.
__device__ __forceinline__ NmsWs image_ws(const NmsWs& w, size_t stride, int b) {
    NmsWs o = w;
    const size_t sh = stride * (size_t)b;
    o.counters = (int*)((char*)w.counters + sh);
    o.flag = w.flag + sh;
    o.keep_idx = (int*)((char*)w.keep_idx + sh);
    o.boxes_k = (float*)((char*)w.boxes_k + sh);
    o.scores_t = (float*)((char*)w.scores_t + sh);
    o.top_idx = (int*)((char*)w.top_idx + sh);
    o.top_score = (float*)((char*)w.top_score + sh);
    o.top_cnt = (int*)((char*)w.top_cnt + sh);
    o.chunk_cnt = (int*)((char*)w.chunk_cnt + sh);
    return o;
}

NmsWs carve(void* base, int N, int C) {
    NmsWs w;
    size_t off = 0;
    auto take = [&](size_t bytes) { char* p = (char*)base + off; off += align_up(bytes); return (void*)p; };
    w.counters = (int*)take(64 * sizeof(int));
    w.flag = (uint8_t*)take((size_t)N);
    w.keep_idx = (int*)take((size_t)N * sizeof(int));
    w.boxes_k = (float*)take((size_t)N * 4 * sizeof(float));
    w.scores_t = (float*)take((size_t)(C - 1) * N * sizeof(float));
    w.top_idx = (int*)take((size_t)(C - 1) * TOPK_CAP * sizeof(int));
    w.top_score = (float*)take((size_t)(C - 1) * TOPK_CAP * sizeof(float));
    w.top_cnt = (int*)take((size_t)(C - 1) * sizeof(int));
    w.chunk_cnt = (int*)take((size_t)((N + CHUNK - 1) / CHUNK) * sizeof(int));
    w.bytes = off;
    return w;
}

// ---------------------------------------------------------------------------------------------------
// stage A: score filter, ordered compaction, decode + transpose  (utils/output_utils.py:135-153)
// ---------------------------------------------------------------------------------------------------
// One workgroup per chunk of CHUNK consecutive anchors (16 waves x 4 anchors): flag = max over the foreground classes > threshold,
// and the chunk's count of flagged anchors -- what the ordered compaction of the next launch needs to know about every OTHER chunk.
__global__ __launch_bounds__(NT) void k_score_flag_count(const float* __restrict__ cls, int N, int C, float thre,
                                                          uint8_t* __restrict__ flag, int* __restrict__ chunk_cnt, size_t ws_stride) {
    cls += (size_t)blockIdx.y * N * C;
    flag += ws_stride * blockIdx.y;
    chunk_cnt = (int*)((char*)chunk_cnt + ws_stride * blockIdx.y);
    __shared__ int cnt;
    if (threadIdx.x == 0) cnt = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    constexpr int PER = CHUNK / (NT / 64);
    const int a0 = blockIdx.x * CHUNK + wv * PER;
    float m[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) {                       // (all rows' loads in flight before the first reduction)
        const int a = a0 + i;
        m[i] = -INFINITY;
        if (a < N) {
            const float* row = cls + (size_t)a * C;
            for (int c = 1 + lane; c < C; c += 64) m[i] = fmaxf(m[i], row[c]);
        }
    }
    int local = 0;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        float v = m[i];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
        const int a = a0 + i;
        if (a < N) {
            const bool keep = v > thre;
            if (lane == 0) flag[a] = keep ? 1 : 0;
            local += keep ? 1 : 0;
        }
    }
    if (lane == 0 && local) atomicAdd(&cnt, local);
    __syncthreads();
    if (threadIdx.x == 0) chunk_cnt[blockIdx.x] = cnt;
}

// exp() of the box decode (utils/output_utils.py:150).  The reference's torch.exp is Intel MKL VML here (closed source, host-ISA
// dependent, 1 ulp off the correctly rounded value in 1.1 % of inputs), so the anchor for this primitive is exp(x) ROUNDED TO
// NEAREST float: evaluated in IEEE double with a fixed operation sequence (v_rndne_f64, v_fma_f64, v_mul_f64, one
// v_cvt_f32_f64) that oracle/expf_cr.c states independently -> bit-identical to the oracle on every input.
__device__ __forceinline__ float expf_cr(float xf) {
    if (xf != xf) return xf;
    const double x = (double)xf;
    if (x > 89.0) return __builtin_inff();
    if (x < -104.0) return 0.f;
    const double k = __builtin_rint(x * 1.44269504088896338700e+00);
    double r = __builtin_fma(k, -6.93147180369123816490e-01, x);
    r = __builtin_fma(k, -1.90821492927058770002e-10, r);
    double p = 1.0 / 6227020800.0;
    p = __builtin_fma(p, r, 1.0 / 479001600.0);
    p = __builtin_fma(p, r, 1.0 / 39916800.0);
    p = __builtin_fma(p, r, 1.0 / 3628800.0);
    p = __builtin_fma(p, r, 1.0 / 362880.0);
    p = __builtin_fma(p, r, 1.0 / 40320.0);
    p = __builtin_fma(p, r, 1.0 / 5040.0);
    p = __builtin_fma(p, r, 1.0 / 720.0);
    p = __builtin_fma(p, r, 1.0 / 120.0);
    p = __builtin_fma(p, r, 1.0 / 24.0);
    p = __builtin_fma(p, r, 1.0 / 6.0);
    p = __builtin_fma(p, r, 0.5);
    p = __builtin_fma(p, r, 1.0);
    p = __builtin_fma(p, r, 1.0);
    const long long bits = ((long long)k + 1023) << 52;
    return (float)(p * __builtin_bit_cast(double, bits));
}

__global__ __launch_bounds__(256) void k_expf_cr(const float* __restrict__ x, float* __restrict__ y, long long n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) y[i] = expf_cr(x[i]);
}

// Ordered compaction + box decode + class-major score matrix in ONE pass over the flagged anchors (boolean-mask gather order =
// ascending anchor index, utils/output_utils.py:135-153).  Workgroup b owns chunk b: its first output slot is the sum of the
// counts of the chunks before it (<= a few hundred integers, summed by the workgroup itself), so no single-workgroup scan launch
// stands between the score filter and the decode; chunks without a flagged anchor exit at once.
__global__ __launch_bounds__(256) void k_compact_decode(const float* __restrict__ cls, const float* __restrict__ box,
                                                         const float* __restrict__ anchors, int N, int C,
                                                         const uint8_t* __restrict__ flag, const int* __restrict__ chunk_cnt,
                                                         int* __restrict__ keep_idx, int* __restrict__ counters,
                                                         float* __restrict__ boxes_k, float* __restrict__ scores_t, size_t ws_stride) {
    {
        const size_t sh = ws_stride * blockIdx.y;
        cls += (size_t)blockIdx.y * N * C;
        box += (size_t)blockIdx.y * N * 4;
        flag += sh;
        chunk_cnt = (const int*)((const char*)chunk_cnt + sh);
        keep_idx = (int*)((char*)keep_idx + sh);
        counters = (int*)((char*)counters + sh);
        boxes_k = (float*)((char*)boxes_k + sh);
        scores_t = (float*)((char*)scores_t + sh);
    }
    extern __shared__ float tile[];  // [CHUNK][C]
    __shared__ int wsum[4];
    __shared__ int kept_a[CHUNK];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, CC = C - 1;
    const int b = blockIdx.x, nchunks = gridDim.x;
    int part = 0;
    for (int i = tid; i < b; i += 256) part += chunk_cnt[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
    if (lane == 0) wsum[wv] = part;
    __syncthreads();
    const int k0 = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    const int cntb = chunk_cnt[b];
    if (b == nchunks - 1 && tid == 0) counters[0] = k0 + cntb;         // K: anchors over threshold
    if (cntb == 0) return;
    if (tid < CHUNK) {
        const int a = b * CHUNK + tid;
        const bool f = a < N && flag[a];
        const unsigned long long bal = __ballot(f);
        const int pos = __popcll(bal & ((1ull << lane) - 1ull));
        if (f) {
            kept_a[pos] = a;
            keep_idx[k0 + pos] = a;
            const f32x4 an = *reinterpret_cast<const f32x4*>(anchors + (size_t)a * 4);
            const f32x4 bx = *reinterpret_cast<const f32x4*>(box + (size_t)a * 4);
            const float cx = an[0] + (bx[0] * 0.1f) * an[2];
            const float cy = an[1] + (bx[1] * 0.1f) * an[3];
            const float w = an[2] * expf_cr(bx[2] * 0.2f);
            const float h = an[3] * expf_cr(bx[3] * 0.2f);
            float x1 = cx - w / 2.f, y1 = cy - h / 2.f;
            float x2 = w + x1, y2 = h + y1;
            auto clip01 = [](float v) { return v != v ? v : fminf(fmaxf(v, 0.f), 1.f); };
            f32x4 o = {clip01(x1), clip01(y1), clip01(x2), clip01(y2)};
            *reinterpret_cast<f32x4*>(boxes_k + (size_t)(k0 + pos) * 4) = o;
        }
    }
    __syncthreads();
    for (int e = tid; e < cntb * CC; e += 256) {
        const int r = e / CC, c = e - r * CC;
        tile[r * C + c] = cls[(size_t)kept_a[r] * C + 1 + c];
    }
    __syncthreads();
    const int r = tid & 63;
    if (r < cntb)
        for (int c = tid >> 6; c < CC; c += 4) scores_t[(size_t)c * N + k0 + r] = tile[r * C + c];
}

// ---------------------------------------------------------------------------------------------------
// stage B: per-class top-k + IoU column test  (utils/output_utils.py:12-26)
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void k_class_topk_iou(const NmsWs w0, int N, int top_k, float iou_thre, size_t ws_stride) {
    const NmsWs w = image_ws(w0, ws_stride, blockIdx.y);
    __shared__ TopkShared<TOPK_CAP> sh;
    __shared__ __attribute__((aligned(16))) float sbox[TOPK_CAP * 4];
    YM_NMS_STAMP(w.counters, 8);
    const int K = w.counters[0];
    if (K == 0) return;
    const int c = blockIdx.x, tid = threadIdx.x;
    const float* srow = w.scores_t + (size_t)c * N;
    const int cnt = block_topk_sorted<TOPK_CAP>([&](int i) { return f2key(srow[i]); }, K, top_k, sh, w.counters);
    YM_NMS_STAMP(w.counters, 9);
    __shared__ uint8_t skeep[TOPK_CAP];
    __shared__ int wtot[TOPK_CAP / 64];
    for (int j = tid; j < cnt; j += NT)
        *reinterpret_cast<f32x4*>(sbox + j * 4) = *reinterpret_cast<const f32x4*>(w.boxes_k + (size_t)sh.idx[j] * 4);
    __syncthreads();
    // wavefront reduction: one wave per column j, lanes stride over the higher-scored rows i < j.
    // keep[j] = max_i<j IoU(i,j) <= thre with torch.max's NaN propagation  <=>  every IoU(i,j) <= thre.
    const int lane = tid & 63, wv = tid >> 6;
    for (int j = wv; j < cnt; j += NT / 64) {
        const f32x4 bj = *reinterpret_cast<const f32x4*>(sbox + j * 4);
        bool bad = false;
        for (int i = lane; i < j; i += 64) {
            const f32x4 bi = *reinterpret_cast<const f32x4*>(sbox + i * 4);
            const float v = iou_pair(bi, bj);
            bad |= !(v <= iou_thre);
        }
        const bool any_bad = __any(bad);
        if (lane == 0) skeep[j] = any_bad ? 0 : 1;
    }
    __syncthreads();
    YM_NMS_STAMP(w.counters, 10);
    // the survivors, compacted in rank order (still sorted by score, ties by index): what stage C merges
    bool f = false;
    int pre = 0;
    if (tid < TOPK_CAP) {
        f = tid < cnt && skeep[tid];
        const unsigned long long bal = __ballot(f);
        pre = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wtot[wv] = __popcll(bal);
    }
    __syncthreads();
    if (tid < TOPK_CAP && f) {
        int off = pre;
        for (int x = 0; x < wv; ++x) off += wtot[x];
        w.top_idx[c * TOPK_CAP + off] = sh.idx[tid];
        w.top_score[c * TOPK_CAP + off] = key2f(sh.keys[tid]);
    }
    if (tid == 0) {
        int t = 0;
        for (int x = 0; x < TOPK_CAP / 64; ++x) t += wtot[x];
        w.top_cnt[c] = t;
    }
    YM_NMS_STAMP(w.counters, 11);
}

// ---------------------------------------------------------------------------------------------------
// stage C: global top max_det over the kept (class, rank) pairs + gather  (utils/output_utils.py:31-43)
// ---------------------------------------------------------------------------------------------------
// Every class's survivors arrive sorted (stage B), and an entry with >= max_det survivors of its OWN class ahead of it cannot be
// among the global top max_det: the answer is the first max_det entries of the merge of the leading min(count, max_det) entries
// of ncls sorted lists (key descending, ties by the lower flat slot = class-major, rank-minor).
//
// Rounds 4-5 merged them with ONE wave, a head per lane, max_det dependent rounds (24 us for 100 detections: a lone wave's
// instruction count is its latency).  This form has no loop over the detections: a RADIX SELECT ON SORTED LISTS.  The max_det-th
// largest key T is found 4 bits per round; inside a class the keys that share the prefix chosen so far are a contiguous range
// [lo, hi) of its list, so "how many keys of class c fall into bin b" needs no histogram over the keys: lane (c, b) binary-searches
// the range for the first key below prefix | b << shift (<= 8 LDS reads in the first round, 1-2 once the range is a few entries),
// the 16 cumulative counts meet in LDS (one ds_add per lane), every wave reads them back with ONE lane-per-bin load + a ballot
// and narrows its classes' ranges with two ds_bpermutes.  After 8 rounds class c contributes lo_c keys above T plus its share of
// the ties (classes in ascending order); the <= max_det selected (key, ~slot) pairs are packed and ranked by counting.
struct FinalOut {
    const float* coef;
    int coef_dim;
    int32_t* out_count;
    int64_t* out_ids;
    float* out_scores;
    float* out_boxes;
    float* out_coefs;
};
__device__ __forceinline__ FinalOut image_out(FinalOut o, size_t b, int max_det, int N) {   // image b's outputs: [B][max_det] rows
    o.coef += b * (size_t)N * o.coef_dim;
    o.out_count += b;
    o.out_ids += b * max_det;
    o.out_scores += b * max_det;
    o.out_boxes += b * (size_t)max_det * 4;
    o.out_coefs += b * (size_t)max_det * o.coef_dim;
    return o;
}

struct FinalShared {
    unsigned long long cand[DET_CAP];   // (key << 32 | ~flat slot) of the selected entries, class-major
    uint32_t hist[8][16];               // per round: cumulative count of range keys with digit >= b, summed over the classes
    int sel[DET_CAP];                   // flat slot of output row j
    int s_k[DET_CAP], s_a[DET_CAP];
    int lcnt[256];                      // (ncls <= 255) leading survivors of a class that can matter: min(count, max_det)
    int s_gt[256], s_eq[256];           // keys above the threshold / equal to it; then s_gt = entries the class contributes
    int s_base[256];                    // first packed slot of a class
    int wtot[2][4];
};

// exclusive prefix sum of v over threads 0..255 (values of the other threads are ignored); every thread of the workgroup calls it
__device__ __forceinline__ int block256_excl_scan(int v, int* wtot) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int incl = wave_incl_scan_add(v);
    if (wv < 4 && lane == 63) wtot[wv] = incl;
    __syncthreads();
    int off = 0;
    for (int x = 0; x < 4; ++x) off += x < wv ? wtot[x] : 0;
    return off + incl - v;
}

// LPL = classes per lane group (ncls <= 64 * LPL)
template <int LPL>
__device__ __forceinline__ void final_stage(const NmsWs& w, int ncls, int max_det, const FinalOut& o, FinalShared& fs) {
    extern __shared__ uint32_t lkeys[];                    // [ncls][DET_CAP]: keys of each class's leading survivors, 0 = none
    const int tid = threadIdx.x, lane = tid & 63;
    YM_NMS_STAMP(w.counters, 0);
    if (tid < 256) fs.lcnt[tid] = tid < ncls ? min(w.top_cnt[tid], max_det) : 0;
    if (tid < 8 * 16) (&fs.hist[0][0])[tid] = 0u;
    __syncthreads();
    YM_NMS_STAMP(w.counters, 1);
    // the leading DET_CAP scores of every class row, all loads independent (entries past the class's survivor count are whatever an
    // earlier launch left there: masked by the count, never interpreted)
    for (int q4 = tid; q4 < ncls * (DET_CAP / 4); q4 += NT) {       // (<= 8 rounds of 1024 threads)
        const int c = q4 / (DET_CAP / 4), j = (q4 - c * (DET_CAP / 4)) * 4;
        const int cnt = fs.lcnt[c];
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (j < cnt) v = *reinterpret_cast<const f32x4*>(w.top_score + c * TOPK_CAP + j);
        uint32_t* dst = lkeys + c * DET_CAP + j;
#pragma unroll
        for (int e = 0; e < 4; ++e) dst[e] = j + e < cnt ? f2key(v[e]) : 0u;
    }
    __syncthreads();
    YM_NMS_STAMP(w.counters, 2);
    // ---- radix select on the sorted lists: lane group g = tid / 16 owns classes g, g + 64, ...; lane b of the group owns bin b
    const int g = tid >> 4, b = tid & 15;
    int lo[LPL], hi[LPL];
#pragma unroll
    for (int q = 0; q < LPL; ++q) {
        const int c = g + 64 * q;
        lo[q] = 0;
        hi[q] = c < ncls ? fs.lcnt[c] : 0;
    }
    uint32_t prefix = 0u;
    int above = 0, need = 0;
#pragma unroll 1
    for (int r = 0; r < 8; ++r) {
        const int shift = 28 - 4 * r;
        const uint32_t thr = prefix | ((uint32_t)b << shift);
        // F = entries of [lo, hi) with key >= thr (the range is sorted descending and shares the prefix: keys >= prefix throughout)
        int l[LPL], h[LPL];
#pragma unroll
        for (int q = 0; q < LPL; ++q) { l[q] = lo[q]; h[q] = hi[q]; }
        bool more;
        do {
            more = false;
#pragma unroll
            for (int q = 0; q < LPL; ++q) {
                if (l[q] < h[q]) {
                    const int m = (l[q] + h[q]) >> 1;
                    const uint32_t k = lkeys[(g + 64 * q) * DET_CAP + m];
                    if (k >= thr) l[q] = m + 1; else h[q] = m;
                    more |= l[q] < h[q];
                }
            }
        } while (__any(more));
        int F[LPL], part = 0;
#pragma unroll
        for (int q = 0; q < LPL; ++q) { F[q] = l[q] - lo[q]; part += F[q]; }
        if (part) atomicAdd(&fs.hist[r][b], (uint32_t)part);
        __syncthreads();
        // bin of T: the largest b with above + S(b) >= need (S is non-increasing in b, S(0) = every key of the ranges)
        const int S = (int)fs.hist[r][lane & 15];
        if (r == 0) need = min(max_det, __builtin_amdgcn_readfirstlane(S));      // S(0) of round 0 = every real key
        const unsigned long long okm = __ballot(above + S >= need) & 0xFFFFull;    // lanes 0..15: bins 0..15
        const int bs = need > 0 ? 63 - __builtin_clzll(okm | 1ull) : 0;
        const int s_next = bs < 15 ? __builtin_amdgcn_readlane(S, bs + 1) : 0;
        above += s_next;
        prefix |= (uint32_t)bs << shift;
#pragma unroll
        for (int q = 0; q < LPL; ++q) {
            const int fb = __shfl(F[q], (lane & 48) | bs);
            const int fb1 = __shfl(F[q], (lane & 48) | ((bs + 1) & 15));
            hi[q] = lo[q] + fb;
            lo[q] = lo[q] + (bs < 15 ? fb1 : 0);
        }
    }
    // prefix = T, [lo, hi) = the class's keys equal to T, lo = its keys above T, above = all keys above T (< need)
    if (b == 0) {
#pragma unroll
        for (int q = 0; q < LPL; ++q) {
            const int c = g + 64 * q;
            if (c < ncls) { fs.s_gt[c] = lo[q]; fs.s_eq[c] = hi[q] - lo[q]; }
        }
    }
    __syncthreads();
    const int n = need;
    {
        // ties at T go to the classes in ascending order (lower flat slot first)
        const int eq = tid < ncls ? fs.s_eq[tid] : 0;
        const int eq_before = block256_excl_scan(eq, fs.wtot[0]);
        const int take = min(max(n - above - eq_before, 0), eq);
        const int mine = tid < ncls ? fs.s_gt[tid] + take : 0;
        const int base = block256_excl_scan(mine, fs.wtot[1]);
        if (tid < ncls) { fs.s_gt[tid] = mine; fs.s_base[tid] = base; }
    }
    __syncthreads();
    for (int e = tid; e < ncls * DET_CAP; e += NT) {
        const int c = e / DET_CAP, j = e - c * DET_CAP;
        if (j < fs.s_gt[c]) fs.cand[fs.s_base[c] + j] = ((unsigned long long)lkeys[e] << 32) | (uint32_t)~(uint32_t)(c * TOPK_CAP + j);
    }
    __syncthreads();
    {
        // rank by counting: 8 lanes share an entry (n <= DET_CAP = NT / 8), packed values are distinct
        static_assert(NT / 8 == DET_CAP, "eight lanes per selected entry");
        const int i = tid >> 3, part8 = tid & 7;
        const unsigned long long mine = i < n ? fs.cand[i] : 0ull;
        int before = 0;
        for (int k = part8; k < n; k += 8) before += fs.cand[k] > mine ? 1 : 0;
        before += __shfl_xor(before, 1);
        before += __shfl_xor(before, 2);
        before += __shfl_xor(before, 4);
        if (part8 == 0 && i < n) fs.sel[before] = (int)~(uint32_t)mine;
    }
    __syncthreads();
    YM_NMS_STAMP(w.counters, 3);
    if (tid == 0) o.out_count[0] = n;
    // the two dependent look-ups (slot -> compacted candidate -> anchor) once per detection; the gathers below are then single loads
    if (tid < n) {
        const int k = w.top_idx[fs.sel[tid]];
        fs.s_k[tid] = k;
        fs.s_a[tid] = w.keep_idx[k];
    }
    __syncthreads();
    if (tid < n) {
        const int f = fs.sel[tid];
        const int c = f / TOPK_CAP;
        o.out_ids[tid] = c;
        o.out_scores[tid] = key2f(lkeys[c * DET_CAP + (f - c * TOPK_CAP)]);
        *reinterpret_cast<f32x4*>(o.out_boxes + tid * 4) = *reinterpret_cast<const f32x4*>(w.boxes_k + (size_t)fs.s_k[tid] * 4);
    }
    for (int e = tid; e < n * o.coef_dim; e += NT) {
        const int j = e / o.coef_dim, d = e - j * o.coef_dim;
        o.out_coefs[e] = o.coef[(size_t)fs.s_a[j] * o.coef_dim + d];
    }
    YM_NMS_STAMP(w.counters, 4);
}

template <int LPL>
__global__ __launch_bounds__(NT) void k_final_select(const NmsWs w0, int ncls, int max_det, const FinalOut o0, size_t ws_stride, int N) {
    const NmsWs w = image_ws(w0, ws_stride, blockIdx.y);
    const FinalOut o = image_out(o0, blockIdx.y, max_det, N);
    __shared__ FinalShared fs;
    if (w.counters[0] == 0) {
        if (threadIdx.x == 0) o.out_count[0] = 0;
        return;
    }
    final_stage<LPL>(w, ncls, max_det, o, fs);
}

// ---------------------------------------------------------------------------------------------------
// greedy NMS (cython_nms.pyx:24-74): rank by counting -> sorted order -> sequential suppression
// ---------------------------------------------------------------------------------------------------
// order: score descending, ties by HIGHER index first (argsort()[::-1] of a stable ascending sort).
__device__ __forceinline__ bool greedy_before(float sa, int ia, float sb, int ib) {
    return sa > sb || (sa == sb && ia > ib);
}

// One block handles one list of n detections given as (box[4], score) with element accessor lambdas.
// sorted_box/sorted_id/alive are global scratch of n entries each.
template <typename BoxAt, typename ScoreAt>
__device__ void block_greedy_nms(BoxAt box_at, ScoreAt score_at, int n, float thresh, float scale,
                                 float* sorted_box, int* sorted_id, uint8_t* alive) {
    const int tid = threadIdx.x, nt = blockDim.x;
    for (int i = tid; i < n; i += nt) {
        const float s = score_at(i);
        int rank = 0;
        for (int j = 0; j < n; ++j) rank += greedy_before(score_at(j), j, s, i) ? 1 : 0;
        const f32x4 b = box_at(i);
        f32x4 sb = {b[0] * scale, b[1] * scale, b[2] * scale, b[3] * scale};
        *reinterpret_cast<f32x4*>(sorted_box + (size_t)rank * 4) = sb;
        sorted_id[rank] = i;
        alive[rank] = 1;
    }
    __syncthreads();
    for (int a = 0; a < n; ++a) {
        if (alive[a]) {   // uniform: written before the previous barrier
            const f32x4 bi = *reinterpret_cast<const f32x4*>(sorted_box + (size_t)a * 4);
            const float iarea = (bi[2] - bi[0] + 1.f) * (bi[3] - bi[1] + 1.f);
            for (int b = a + 1 + tid; b < n; b += nt) {
                if (!alive[b]) continue;
                const f32x4 bj = *reinterpret_cast<const f32x4*>(sorted_box + (size_t)b * 4);
                const float xx1 = fmaxf(bi[0], bj[0]), yy1 = fmaxf(bi[1], bj[1]);
                const float xx2 = fminf(bi[2], bj[2]), yy2 = fminf(bi[3], bj[3]);
                float ww = xx2 - xx1 + 1.f, hh = yy2 - yy1 + 1.f;
                ww = ww >= 0.f ? ww : 0.f;
                hh = hh >= 0.f ? hh : 0.f;
                const float inter = ww * hh;
                const float jarea = (bj[2] - bj[0] + 1.f) * (bj[3] - bj[1] + 1.f);
                const float ovr = __fdiv_rn(inter, (iarea + jarea) - inter);
                if (ovr >= thresh) alive[b] = 0;
            }
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(NT) void k_greedy_standalone(const float* __restrict__ dets, int n, float thresh,
                                                           uint8_t* __restrict__ keep_mask, int32_t* __restrict__ out_count,
                                                           float* sorted_box, int* sorted_id, uint8_t* alive) {
    __shared__ int cnt;
    if (threadIdx.x == 0) cnt = 0;
    block_greedy_nms(
        [&](int i) { const float* d = dets + (size_t)i * 5; return f32x4{d[0], d[1], d[2], d[3]}; },
        [&](int i) { return dets[(size_t)i * 5 + 4]; }, n, thresh, 1.f, sorted_box, sorted_id, alive);
    for (int r = threadIdx.x; r < n; r += NT) {
        keep_mask[sorted_id[r]] = alive[r];
        if (alive[r]) atomicAdd(&cnt, 1);
    }
    __syncthreads();
    if (threadIdx.x == 0 && out_count) out_count[0] = cnt;
}

// per class: candidates = kept anchors with score[c] > thre (utils/output_utils.py:93-109)
struct GreedyWs {
    int* cand;          // [C-1][N] compacted-list indices of the class candidates (ascending)
    int* cand_cnt;      // [C-1]
    float* sorted_box;  // [C-1][N][4]
    int* sorted_id;     // [C-1][N]
    uint8_t* alive;     // [C-1][N]
    uint8_t* kept;      // [C-1][N] kept flag by candidate position (ascending index order)
};

__global__ __launch_bounds__(NT) void k_greedy_per_class(const NmsWs w, const GreedyWs g, int N, float score_thre,
                                                          float iou_thre, float img_size) {
    __shared__ int wave_tot[NT / 64];
    __shared__ int running;
    const int K = w.counters[0];
    if (K == 0) return;
    const int c = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const float* srow = w.scores_t + (size_t)c * N;
    int* cand = g.cand + (size_t)c * N;
    if (tid == 0) running = 0;
    __syncthreads();
    for (int base = 0; base < K; base += NT) {
        const int i = base + tid;
        const bool f = i < K && srow[i] > score_thre;
        const unsigned long long bal = __ballot(f);
        const int pre = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wave_tot[wv] = __popcll(bal);
        __syncthreads();
        int off = running;
        for (int x = 0; x < wv; ++x) off += wave_tot[x];
        if (f) cand[off + pre] = i;
        __syncthreads();
        if (tid == 0) { int t = 0; for (int x = 0; x < NT / 64; ++x) t += wave_tot[x]; running += t; }
        __syncthreads();
    }
    const int n = running;
    if (tid == 0) g.cand_cnt[c] = n;
    if (n == 0) return;
    float* sbox = g.sorted_box + (size_t)c * N * 4;
    int* sid = g.sorted_id + (size_t)c * N;
    uint8_t* alive = g.alive + (size_t)c * N;
    block_greedy_nms(
        [&](int i) { return *reinterpret_cast<const f32x4*>(w.boxes_k + (size_t)cand[i] * 4); },
        [&](int i) { return srow[cand[i]]; }, n, iou_thre, img_size, sbox, sid, alive);
    uint8_t* kept = g.kept + (size_t)c * N;
    for (int r = tid; r < n; r += NT) kept[sid[r]] = alive[r];
}

// global top max_det over all kept (class, candidate) pairs; flat order = class-major, ascending index
// inside a class (torch.cat of idx[keep] with keep ascending, utils/output_utils.py:107-115).
__global__ __launch_bounds__(NT) void k_greedy_final(const NmsWs w, const GreedyWs g, int N, int ncls, int max_det,
                                                     float img_size, const float* __restrict__ coef, int coef_dim,
                                                     int32_t* __restrict__ out_count, int64_t* __restrict__ out_ids,
                                                     float* __restrict__ out_scores, float* __restrict__ out_boxes,
                                                     float* __restrict__ out_coefs) {
    __shared__ TopkShared<DET_CAP> sh;
    __shared__ int n_valid;
    const int tid = threadIdx.x;
    const int K = w.counters[0];
    if (K == 0) {
        if (tid == 0) out_count[0] = 0;
        return;
    }
    const long long Lfull = (long long)ncls * N;
    const int L = (int)Lfull;
    auto key_at = [&](int f) -> uint32_t {
        const int c = f / N, p = f - c * N;
        if (p < g.cand_cnt[c] && g.kept[f]) return f2key(w.scores_t[(size_t)c * N + g.cand[f]]);
        return 0u;
    };
    block_topk_sorted<DET_CAP>(key_at, L, max_det, sh);
    if (tid == 0) n_valid = 0;
    __syncthreads();
    if (tid < max_det && sh.keys[tid] != 0u) atomicAdd(&n_valid, 1);
    __syncthreads();
    const int n = n_valid;
    if (tid == 0) out_count[0] = n;
    for (int j = tid; j < n; j += NT) {
        const int f = sh.idx[j];
        const int c = f / N;
        const int k = g.cand[f];
        out_ids[j] = c;
        out_scores[j] = key2f(sh.keys[j]);
        // boxes[idx] / img_size of boxes * img_size (utils/output_utils.py:90,123)
        const f32x4 b = *reinterpret_cast<const f32x4*>(w.boxes_k + (size_t)k * 4);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = __fdiv_rn(b[e] * img_size, img_size);
        *reinterpret_cast<f32x4*>(out_boxes + j * 4) = o;
    }
    for (int e = tid; e < n * coef_dim; e += NT) {
        const int j = e / coef_dim, d = e - j * coef_dim;
        const int a = w.keep_idx[g.cand[sh.idx[j]]];
        out_coefs[e] = coef[(size_t)a * coef_dim + d];
    }
}

size_t greedy_extra_bytes(int N, int C) {
    const size_t cn = (size_t)(C - 1) * N;
    return align_up(cn * 4) + align_up((size_t)(C - 1) * 4) + align_up(cn * 16) + align_up(cn * 4) + align_up(cn) + align_up(cn);
}

GreedyWs carve_greedy(void* base, int N, int C) {
    GreedyWs g;
    size_t off = 0;
    const size_t cn = (size_t)(C - 1) * N;
    auto take = [&](size_t bytes) { char* p = (char*)base + off; off += align_up(bytes); return (void*)p; };
    g.cand = (int*)take(cn * 4);
    g.cand_cnt = (int*)take((size_t)(C - 1) * 4);
    g.sorted_box = (float*)take(cn * 16);
    g.sorted_id = (int*)take(cn * 4);
    g.alive = (uint8_t*)take(cn);
    g.kept = (uint8_t*)take(cn);
    return g;
}

int check_cfg(const ym_nms_cfg* cfg) {
    YM_REQUIRE(cfg, "nms: null cfg");
    YM_REQUIRE(cfg->num_anchors > 0 && cfg->num_classes >= 2 && cfg->num_classes <= 256, "nms: bad N/C");
    YM_REQUIRE(cfg->top_k >= 1 && cfg->top_k <= TOPK_CAP, "nms: top_k must be 1..%d", TOPK_CAP);
    YM_REQUIRE(cfg->max_det >= 1 && cfg->max_det <= DET_CAP, "nms: max_det must be 1..%d", DET_CAP);
    YM_REQUIRE(cfg->coef_dim >= 1, "nms: coef_dim");
    YM_REQUIRE((long long)(cfg->num_classes - 1) * cfg->num_anchors < (1ll << 31), "nms: N*C too large");
    return YM_OK;
}

int run_stage_a(const float* cls, const float* box, const float* anchors, const ym_nms_cfg* cfg, const NmsWs& w,
                hipStream_t st, int B = 1, size_t ws_stride = 0) {
    const int N = cfg->num_anchors, C = cfg->num_classes;
    const int nchunks = ym_cdiv(N, CHUNK);
    hipLaunchKernelGGL(k_score_flag_count, dim3(nchunks, B), dim3(NT), 0, st, cls, N, C, cfg->score_thre, w.flag, w.chunk_cnt, ws_stride);
    hipLaunchKernelGGL(k_compact_decode, dim3(nchunks, B), dim3(256), (size_t)CHUNK * C * sizeof(float), st, cls, box, anchors, N, C,
                       w.flag, w.chunk_cnt, w.keep_idx, w.counters, w.boxes_k, w.scores_t, ws_stride);
    return ym_check_launch("nms stage A");
}

}  // namespace

extern "C" size_t ym_nms_workspace_bytes(const ym_nms_cfg* cfg) {
    if (check_cfg(cfg) != YM_OK) return 0;
    NmsWs w = carve(nullptr, cfg->num_anchors, cfg->num_classes);
    return w.bytes + greedy_extra_bytes(cfg->num_anchors, cfg->num_classes);
}

extern "C" size_t ym_nms_batch_workspace_bytes(const ym_nms_cfg* cfg, int B) {
    if (check_cfg(cfg) != YM_OK || B < 1) return 0;
    return align_up(carve(nullptr, cfg->num_anchors, cfg->num_classes).bytes) * (size_t)B;
}

extern "C" int ym_detect_fast_nms_batch(const float* class_pred, const float* box_pred, const float* coef_pred,
                                        const float* anchors, const ym_nms_cfg* cfg, int B, int32_t* out_count, int64_t* out_ids,
                                        float* out_scores, float* out_boxes, float* out_coefs, void* workspace,
                                        size_t workspace_bytes, ym_stream_t s) {
    int rc = check_cfg(cfg);
    if (rc != YM_OK) return rc;
    YM_REQUIRE(B >= 1 && B <= 65535, "fast_nms: batch must be 1..65535");
    YM_REQUIRE(class_pred && box_pred && coef_pred && anchors && out_count && out_ids && out_scores && out_boxes &&
                   out_coefs && workspace, "fast_nms: null pointer");
    NmsWs w = carve(workspace, cfg->num_anchors, cfg->num_classes);
    const size_t stride = align_up(w.bytes);
    if (stride * (size_t)B > workspace_bytes && !(B == 1 && w.bytes <= workspace_bytes)) {
        ym_set_error("fast_nms: workspace %zu < %zu", workspace_bytes, stride * (size_t)B);
        return YM_ENOSPC;
    }
    hipStream_t st = (hipStream_t)s;
    rc = run_stage_a(class_pred, box_pred, anchors, cfg, w, st, B, stride);
    if (rc != YM_OK) return rc;
    const int ncls = cfg->num_classes - 1;
    hipLaunchKernelGGL(k_class_topk_iou, dim3(ncls, B), dim3(NT), 0, st, w, cfg->num_anchors, cfg->top_k, cfg->iou_thre, stride);
    const size_t merge_lds = (size_t)ncls * DET_CAP * sizeof(uint32_t);          // <= 255 * 128 * 4 = 130 KB
    const int lpl = (ncls + 63) / 64;                    // classes per lane group of the final select (1..4: num_classes <= 256)
    const FinalOut fo = {coef_pred, cfg->coef_dim, out_count, out_ids, out_scores, out_boxes, out_coefs};
#define YM_FINAL(L_)                                                                                                              \
    do {                                                                                                                          \
        static YmLdsAttr set_ = {};                                                                                               \
        if (int rc_ = ym_ensure_dyn_lds(set_, reinterpret_cast<const void*>(k_final_select<L_>), merge_lds, "final_select")) return rc_; \
        hipLaunchKernelGGL(k_final_select<L_>, dim3(1, B), dim3(NT), merge_lds, st, w, ncls, cfg->max_det, fo, stride, cfg->num_anchors); \
    } while (0)
    if (lpl <= 1) YM_FINAL(1); else if (lpl == 2) YM_FINAL(2); else if (lpl == 3) YM_FINAL(3); else YM_FINAL(4);
#undef YM_FINAL
    return ym_check_launch("fast_nms");
}

extern "C" int ym_detect_fast_nms(const float* class_pred, const float* box_pred, const float* coef_pred,
                                  const float* anchors, const ym_nms_cfg* cfg, int32_t* out_count, int64_t* out_ids,
                                  float* out_scores, float* out_boxes, float* out_coefs, void* workspace,
                                  size_t workspace_bytes, ym_stream_t s) {
    return ym_detect_fast_nms_batch(class_pred, box_pred, coef_pred, anchors, cfg, 1, out_count, out_ids, out_scores, out_boxes,
                                    out_coefs, workspace, workspace_bytes, s);
}

extern "C" int ym_detect_greedy_nms(const float* class_pred, const float* box_pred, const float* coef_pred,
                                    const float* anchors, const ym_nms_cfg* cfg, int32_t* out_count, int64_t* out_ids,
                                    float* out_scores, float* out_boxes, float* out_coefs, void* workspace,
                                    size_t workspace_bytes, ym_stream_t s) {
    int rc = check_cfg(cfg);
    if (rc != YM_OK) return rc;
    YM_REQUIRE(class_pred && box_pred && coef_pred && anchors && out_count && out_ids && out_scores && out_boxes &&
                   out_coefs && workspace, "greedy_nms: null pointer");
    const int N = cfg->num_anchors, C = cfg->num_classes;
    NmsWs w = carve(workspace, N, C);
    const size_t need = w.bytes + greedy_extra_bytes(N, C);
    if (need > workspace_bytes) { ym_set_error("greedy_nms: workspace %zu < %zu", workspace_bytes, need); return YM_ENOSPC; }
    GreedyWs g = carve_greedy((char*)workspace + w.bytes, N, C);
    hipStream_t st = (hipStream_t)s;
    rc = run_stage_a(class_pred, box_pred, anchors, cfg, w, st);
    if (rc != YM_OK) return rc;
    hipLaunchKernelGGL(k_greedy_per_class, dim3(C - 1), dim3(NT), 0, st, w, g, N, cfg->score_thre, cfg->iou_thre,
                       cfg->img_size);
    hipLaunchKernelGGL(k_greedy_final, dim3(1), dim3(NT), 0, st, w, g, N, C - 1, cfg->max_det, cfg->img_size, coef_pred,
                       cfg->coef_dim, out_count, out_ids, out_scores, out_boxes, out_coefs);
    return ym_check_launch("greedy_nms");
}

extern "C" int ym_expf_cr(const float* x, float* y, int64_t n, ym_stream_t s) {
    YM_REQUIRE(n >= 0 && (n == 0 || (x && y)), "expf_cr: bad arguments");
    if (n == 0) return YM_OK;
    int grid = (int)((n + 255) / 256);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(k_expf_cr, dim3(grid), dim3(256), 0, (hipStream_t)s, x, y, (long long)n);
    return ym_check_launch("expf_cr");
}

extern "C" size_t ym_greedy_nms_workspace_bytes(int n) {
    if (n <= 0) return 256;
    return align_up((size_t)n * 16) + align_up((size_t)n * 4) + align_up((size_t)n);
}

extern "C" int ym_greedy_nms(const float* dets, int n, float thresh, uint8_t* keep_mask, int32_t* out_count,
                             void* workspace, size_t workspace_bytes, ym_stream_t s) {
    YM_REQUIRE(n >= 0, "greedy_nms: n < 0");
    if (n == 0) {
        if (out_count) (void)hipMemsetAsync(out_count, 0, sizeof(int32_t), (hipStream_t)s);
        return YM_OK;
    }
    YM_REQUIRE(dets && keep_mask && workspace, "greedy_nms: null pointer");
    if (ym_greedy_nms_workspace_bytes(n) > workspace_bytes) { ym_set_error("greedy_nms: workspace too small"); return YM_ENOSPC; }
    char* p = (char*)workspace;
    float* sorted_box = (float*)p; p += align_up((size_t)n * 16);
    int* sorted_id = (int*)p; p += align_up((size_t)n * 4);
    uint8_t* alive = (uint8_t*)p;
    hipLaunchKernelGGL(k_greedy_standalone, dim3(1), dim3(NT), 0, (hipStream_t)s, dets, n, thresh, keep_mask, out_count,
                       sorted_box, sorted_id, alive);
    return ym_check_launch("greedy_nms");
}
