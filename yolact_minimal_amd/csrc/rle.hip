// COCO run-length encoding of the binary detection masks on the device: what
// `pycocotools.mask.encode(np.asfortranarray(mask))` + `.decode('ascii')` produce in the reference's MakeJson.add_mask
// (utils/common_utils.py:88-96, eval.py:64-67).  Algorithm = cocoapi common/maskApi.c rleEncode + rleToString:
// runs over the COLUMN-major pixel order, alternating 0/1 starting with zeros (first count may be 0); each count (from the 4th
// on: its difference to the count two places back) is written as little-endian 5-bit groups + continuation bit, +48.
// One workgroup per mask, one thread per image column (row-major rows are read coalesced across the columns), two passes over
// the mask (count transitions, then place them) = 2*H*W*4 bytes of HBM reads; the result is a few hundred bytes per mask
// instead of a 1.2 MB dense D2H copy.
#include "ym_common.h"

namespace {

constexpr int NT = 1024;
constexpr int MAXW = 4096;
constexpr int UNR = 16;

// exclusive prefix sum of one value per thread over the workgroup; returns the prefix, *total = sum of all
__device__ __forceinline__ int block_exscan(int v, int* total, int* s_wave /*[NT/64 + 1]*/) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(inc, o);
        if (lane >= o) inc += t;
    }
    __syncthreads();
    if (lane == 63) s_wave[wave] = inc;
    __syncthreads();
    if (threadIdx.x == 0) {
        int run = 0;
        for (int w = 0; w < NT / 64; ++w) { const int t = s_wave[w]; s_wave[w] = run; run += t; }
        s_wave[NT / 64] = run;
    }
    __syncthreads();
    *total = s_wave[NT / 64];
    return s_wave[wave] + inc - v;
}

__global__ __launch_bounds__(NT) void k_rle_encode(const float* __restrict__ masks, int H, int W, uint32_t* __restrict__ pos_ws,
                                                   uint32_t* __restrict__ counts, int cap, int32_t* __restrict__ nruns,
                                                   uint8_t* __restrict__ str, int cap_str, int32_t* __restrict__ str_len) {
    __shared__ int s_col[MAXW];
    __shared__ int s_wave[NT / 64 + 1];
    const int tid = threadIdx.x;
    const float* m = masks + (size_t)blockIdx.x * H * W;
    uint32_t* pos = pos_ws + (size_t)blockIdx.x * cap;
    uint32_t* cnt = counts + (size_t)blockIdx.x * cap;
    uint8_t* out = str + (size_t)blockIdx.x * cap_str;
    const unsigned P = (unsigned)H * (unsigned)W;

    // pass 1: transitions per column (the pixel before (0, x) in column-major order is (H-1, x-1); before (0,0): background)
    for (int x = tid; x < W; x += NT) {
        bool prev = x > 0 && m[(size_t)(H - 1) * W + x - 1] != 0.f;
        int c = 0, y = 0;
        for (; y + UNR <= H; y += UNR) {          // UNR independent row loads in flight per lane (the walk is latency bound)
            float v[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u) v[u] = m[(size_t)(y + u) * W + x];
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const bool cur = v[u] != 0.f;
                c += cur != prev;
                prev = cur;
            }
        }
        for (; y < H; ++y) {
            const bool cur = m[(size_t)y * W + x] != 0.f;
            c += cur != prev;
            prev = cur;
        }
        s_col[x] = c;
    }
    __syncthreads();
    // exclusive scan over the columns (chunks of NT with a running carry)
    int carry = 0;
    for (int x0 = 0; x0 < W; x0 += NT) {
        const int x = x0 + tid;
        const int v = x < W ? s_col[x] : 0;
        int total;
        const int pre = block_exscan(v, &total, s_wave);
        if (x < W) s_col[x] = carry + pre;
        carry += total;
        __syncthreads();
    }
    const int T = carry;                 // transitions; runs = T + 1 (the last run ends at P)
    const int R = T + 1;
    if (tid == 0) nruns[blockIdx.x] = R;
    if (R > cap) {                       // caller's buffers are too small: report the size needed, encode nothing
        if (tid == 0) str_len[blockIdx.x] = -1;
        return;
    }
    // pass 2: positions of the transitions, in column-major order
    for (int x = tid; x < W; x += NT) {
        bool prev = x > 0 && m[(size_t)(H - 1) * W + x - 1] != 0.f;
        int o = s_col[x], y = 0;
        for (; y + UNR <= H; y += UNR) {
            float v[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u) v[u] = m[(size_t)(y + u) * W + x];
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const bool cur = v[u] != 0.f;
                if (cur != prev) pos[o++] = (unsigned)x * (unsigned)H + (unsigned)(y + u);
                prev = cur;
            }
        }
        for (; y < H; ++y) {
            const bool cur = m[(size_t)y * W + x] != 0.f;
            if (cur != prev) pos[o++] = (unsigned)x * (unsigned)H + (unsigned)y;
            prev = cur;
        }
    }
    __threadfence_block();
    __syncthreads();
    // counts[j] = pos[j] - pos[j-1]  (pos[-1] = 0, pos[T] = P)
    for (int j = tid; j < R; j += NT) {
        const unsigned hi = j < T ? pos[j] : P, lo = j > 0 ? pos[j - 1] : 0u;
        cnt[j] = hi - lo;
    }
    __threadfence_block();
    __syncthreads();
    // rleToString: x = cnts[j] (j > 2: minus cnts[j-2]); 5 bits per char, bit 5 = "more", sign-aware termination, +48
    int scarry = 0;
    for (int j0 = 0; j0 < R; j0 += NT) {
        const int j = j0 + tid;
        long long x = 0;
        int len = 0;
        uint8_t ch[8];
        if (j < R) {
            x = (long long)cnt[j];
            if (j > 2) x -= (long long)cnt[j - 2];
            bool more = true;
            while (more) {
                int c = (int)(x & 0x1f);
                x >>= 5;
                more = (c & 0x10) ? x != -1 : x != 0;
                if (more) c |= 0x20;
                ch[len++] = (uint8_t)(c + 48);
            }
        }
        int total;
        const int pre = block_exscan(len, &total, s_wave);
        const int o = scarry + pre;
        if (o + len <= cap_str)
            for (int k = 0; k < len; ++k) out[o + k] = ch[k];
        scarry += total;
        __syncthreads();
    }
    if (tid == 0) str_len[blockIdx.x] = scarry <= cap_str ? scarry : -1;
}

}  // namespace

extern "C" int ym_rle_encode(const float* masks, int n, int H, int W, uint32_t* counts, int cap_runs, int32_t* nruns, uint8_t* str,
                             int cap_str, int32_t* str_len, void* workspace, size_t workspace_bytes, ym_stream_t s) {
    YM_REQUIRE(masks && counts && nruns && str && str_len && workspace, "rle_encode: null pointer");
    YM_REQUIRE(n > 0 && H > 0 && W > 0 && W <= MAXW && cap_runs > 0 && cap_str > 0, "rle_encode: need 0 < W <= %d", MAXW);
    YM_REQUIRE((long long)H * W < (1ll << 31), "rle_encode: mask too large");
    if (workspace_bytes < (size_t)n * cap_runs * 4) { ym_set_error("rle_encode: workspace < n*cap_runs*4 bytes"); return YM_ENOSPC; }
    hipLaunchKernelGGL(k_rle_encode, dim3(n), dim3(NT), 0, (hipStream_t)s, masks, H, W, (uint32_t*)workspace, counts, cap_runs, nruns,
                       str, cap_str, str_len);
    return ym_check_launch("rle_encode");
}
