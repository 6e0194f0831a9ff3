// Weight gradient of the fused convolution on the f32 MFMA pipe (gfx950).
//
//   dW[n][kk] = sum_m dY[m][n] * Xg[m][kk]      n = output channel, kk = (kh, kw, ci), m = output pixel (b, oh, ow)
//
// GEMM with the REDUCTION over pixels (M is 10^3..10^6) and a small output ([Cout][KH*KW*Cin]).  Both operands are
// contiguous along the MFMA row/column index (dY rows are Cout floats, X rows are Cin floats of one tap), so the
// 32x32x2 fragments are read from LDS as conflict-free ds_read_b32: lanes 0-31 = 32 consecutive n (or kk) of pixel
// m, lanes 32-63 the same for pixel m+1.  Workgroup tile 128(n) x 128(kk), 4 waves 2x2 of 64x64, pixel step 32,
// register-staged double-buffered LDS.  The pixel range is split across the grid (`msplit`), partial tiles go to
// a caller-provided workspace and a reduce kernel sums them in fixed order and scatters into the OIHW gradient.
#include "conv_common.h"

using namespace ymk;

namespace {

constexpr int TBK = 128;   // kk tile
constexpr int TBM = 32;    // pixels per step (register-staged variants; the DMA variants choose 32 or 16)
constexpr int LP = 132;    // LDS row pitch (floats): 16-byte aligned rows, skewed banks for the float4 writes
constexpr int DP = 128;    // row pitch of the DMA variants: `buffer_load ... lds` places lane l's 16 bytes at base + 16*l, i.e. rows
                           // of 128 floats back to back (the fragments are ds_read_b32 over 32 consecutive floats: any pitch is
                           // conflict-free for them; the skew only ever served the float4 WRITES, which the DMA path does not have)

struct WgradP {
    const float* x;     // NHWC [B][H][W][Cinp]
    const float* dy;    // [M][Cout] (M = B*Ho*Wo)
    float* ws;          // [msplit][Cout][Ktot]
    int B, H, W, Cinp, Cout, KH, KW, stride, pad, Ho, Wo;
    int M, HoWo, Ktot, tiles_n, tiles_k, msplit, m_per_split;
    int incr;          // 1: pixel coordinates advanced incrementally (a 32-pixel step wraps at most two image rows + one image)
    unsigned x_bytes, dy_bytes;
    unsigned mg_howo, sh_howo, mg_wo, sh_wo;   // division by invariant integers (q = (mulhi(m, mg) + m) >> sh)
};

typedef int i32x4_t __attribute__((ext_vector_type(4)));
constexpr unsigned OOB = 0xFFFFFFF0u;
__device__ __forceinline__ f32x4 buf_ld16(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 0));
}
__device__ __forceinline__ unsigned fastdiv(unsigned m, unsigned mg, unsigned sh) { return (__umulhi(m, mg) + m) >> sh; }

// TBN = n tile.  128: waves 2 x 2, each 64(n) x 64(kk).  64 (layers with <= 64 output channels: a 128-wide tile would spend half of
// its MFMAs on zero padding — 35 TFLOP/s on the 64-channel layers of layer1): waves 1 x 4, each 64(n) x 32(kk).
// NB = LDS buffers.  2: the next pixel step is written while the current one is read (one barrier per step, 68 KB: two workgroups
// per CU).  1: one buffer, two barriers per step (34 KB: three workgroups per CU at 152 VGPRs) — the extra barrier is hidden by
// the third resident workgroup; the counters showed this kernel at 1.3 resident waves per SIMD and 57 % MFMA-busy with NB = 2.
// DL: both operand tiles go global -> LDS by DMA (`buffer_load_dwordx4 ... lds`): no staging registers (-32 VGPRs), no ds_write,
// a ring of NB buffers with the loads NB-1 pixel steps ahead and ONE barrier per step (explicit `s_waitcnt vmcnt(n) lgkmcnt(0)` +
// `s_barrier`, the discipline of conv_mfma.hip's DL path: lgkmcnt(0) retires this wave's reads of the buffer the next DMA
// re-stages).  TM = pixels per step: 32 x ring 2 = 64 KB (two workgroups per CU), 16 x ring 3 = 48 KB (three).  TBN = 128 only.
template <int INCR, int TBN, int NB, bool DL = false, int TM = TBM>
__global__ __launch_bounds__(256) void conv_wgrad_f32(const WgradP p) {
    constexpr int KT = TBN == 128 ? 2 : 1;          // 32-wide kk tiles per wave
    constexpr int TMS = TM;                          // pixels per step of this variant
    constexpr int LPK = DL ? DP : LP;                // its LDS row pitch
    static_assert(!DL || TBN == 128, "DMA staging: 128-wide n tile only");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Ys = smem;                       // [NB][TMS][LPK]
    float* Xs = smem + NB * TMS * LPK;       // [NB][TMS][LPK]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = TBN == 128 ? (wave >> 1) : 0, wk = TBN == 128 ? (wave & 1) : wave;

    int id = ym_xcd_remap(blockIdx.x, gridDim.x);
    const int ms = id % p.msplit;
    id /= p.msplit;
    const int tile_n = id / p.tiles_k, tile_k = id - tile_n * p.tiles_k;
    const int n0 = tile_n * TBN, k0 = tile_k * TBK;
    const int m_beg = ms * p.m_per_split, m_end = min(p.M, m_beg + p.m_per_split);

    // staging: thread -> float4 column c4 (0..31) of rows r + 8*i
    const int c4 = tid & 31, r0 = tid >> 5;
    const int ncol = n0 + c4 * 4;
    const bool n_ok = ncol < p.Cout;
    const int kk = k0 + c4 * 4;
    const bool k_ok = kk < p.Ktot;
    const int tap = k_ok ? kk / p.Cinp : 0, ci = kk - tap * p.Cinp;
    const int kh = tap / p.KW, kw = tap - kh * p.KW;

    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc((void*)p.dy, 0, p.dy_bytes, 0x00020000);
    constexpr int NR = TMS / 8;                  // staging rows per thread: r0 + 8*i
    f32x4 ry[DL ? 1 : NR], rx[DL ? 1 : NR];
    // Branch-free operand fetch (raw buffer loads return zeros beyond the descriptor: rows past the slice, padding taps, tile
    // tails) with the per-step integer work kept small: the output-pixel coordinates (b, oh, ow) of each staging row are carried
    // across steps and ADVANCED by the 32-pixel step (a couple of compares) instead of being re-derived with two divisions per
    // row per step; everything that does not depend on the pixel (tap, channel, column masks) is hoisted.  INCR = 0 keeps the
    // division path for maps so small that one step wraps more than two image rows.
    int pb[NR], poh[NR], pow_[NR];
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        const unsigned m = (unsigned)(m_beg + r0 + 8 * i);
        const unsigned b = fastdiv(m, p.mg_howo, p.sh_howo);
        const unsigned rem = m - b * p.HoWo;
        const unsigned oh = fastdiv(rem, p.mg_wo, p.sh_wo);
        pb[i] = (int)b; poh[i] = (int)oh; pow_[i] = (int)(rem - oh * p.Wo);
    }
    const unsigned ymask = n_ok ? 0u : OOB, xmask = k_ok ? 0u : OOB;
    const unsigned ycol = (unsigned)(ncol * 4), xcol = (unsigned)(ci * 4);
    const int dq = TMS / p.Wo, dr = TMS - dq * p.Wo;                  // a 32-pixel step = dq rows + dr columns
    typedef __attribute__((address_space(3))) void* lds_ptr;
    // `buf` (DL only): ring buffer the tile is DMA'd into; the wave's lanes 0-31 / 32-63 are rows 2*wave + 8i / + 1 of the tile, which
    // is where base + 16*lane puts them at a pitch of 128 floats
    auto load = [&](int mt, int buf) {
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const int m = mt + r0 + 8 * i;
            const unsigned dead = m < m_end ? 0u : OOB;
            const unsigned yoff = ((unsigned)(m * p.Cout) * 4u + ycol) | ymask | dead;
            const int ih = poh[i] * p.stride - p.pad + kh, iw = pow_[i] * p.stride - p.pad + kw;
            const unsigned inside = ((unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W) ? 0u : OOB;
            const unsigned xoff = ((unsigned)(((pb[i] * p.H + ih) * p.W + iw) * p.Cinp) * 4u + xcol) | xmask | dead | inside;
            if constexpr (DL) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_y, (lds_ptr)(Ys + (buf * TMS + 2 * wave + 8 * i) * LPK), 16, (int)yoff, 0, 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_ptr)(Xs + (buf * TMS + 2 * wave + 8 * i) * LPK), 16, (int)xoff, 0, 0, 0);
            } else {
                ry[i] = buf_ld16(rs_y, yoff);
                rx[i] = buf_ld16(rs_x, xoff);
            }
            if (INCR) {                                                // advance this row by TMS pixels
                int ow = pow_[i] + dr, oh = poh[i] + dq;
                const bool c = ow >= p.Wo;
                ow -= c ? p.Wo : 0;
                oh += c ? 1 : 0;
                bool c2 = oh >= p.Ho;
                oh -= c2 ? p.Ho : 0;
                pb[i] += c2 ? 1 : 0;
                c2 = oh >= p.Ho;                                       // (a step may wrap two image rows' worth on small maps)
                oh -= c2 ? p.Ho : 0;
                pb[i] += c2 ? 1 : 0;
                pow_[i] = ow; poh[i] = oh;
            } else {
                const unsigned mn = (unsigned)(m + TMS);
                const unsigned b = fastdiv(mn, p.mg_howo, p.sh_howo);
                const unsigned rem = mn - b * p.HoWo;
                const unsigned oh = fastdiv(rem, p.mg_wo, p.sh_wo);
                pb[i] = (int)b; poh[i] = (int)oh; pow_[i] = (int)(rem - oh * p.Wo);
            }
        }
    };
    auto store = [&](int buf) {
#pragma unroll
        for (int i = 0; i < (DL ? 0 : NR); ++i) {
            *reinterpret_cast<f32x4*>(Ys + (buf * TMS + r0 + 8 * i) * LPK + c4 * 4) = ry[i];
            *reinterpret_cast<f32x4*>(Xs + (buf * TMS + r0 + 8 * i) * LPK + c4 * 4) = rx[i];
        }
    };

    f32x16 acc[2][KT];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < KT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int fr = lane & 31, khalf = lane >> 5;
    constexpr int D = NB - 1;                      // DL: prefetch distance in pixel steps
    int cur = 0, nxt = DL ? D : 0;                 // DL: nxt = (cur + D) % NB = the buffer step t-1 was read from
    if constexpr (DL) {
        static_assert(2 * NR * (D > 0 ? D - 1 : 0) < 16, "vmcnt field");
#pragma unroll
        for (int d = 0; d < D; ++d) load(m_beg + d * TMS, d);
        __builtin_amdgcn_s_waitcnt(0x070 | (2 * NR * (D - 1)));     // vmcnt(2*NR*(D-1)) lgkmcnt(0): step 0 has landed
        __builtin_amdgcn_s_barrier();
    } else {
        load(m_beg, 0);
        store(0);
        __syncthreads();
    }
    for (int mt = m_beg; mt < m_end; mt += TMS) {
        load(mt + (DL ? D : 1) * TMS, nxt);
        const float* ya = Ys + cur * TMS * LPK + wn * 64 + fr;
        const float* xb = Xs + cur * TMS * LPK + wk * (32 * KT) + fr;
#pragma unroll
        for (int s = 0; s < TMS / 2; ++s) {
            const int row = 2 * s + khalf;
            const float a0 = ya[row * LPK], a1 = ya[row * LPK + 32];
            const float b0 = xb[row * LPK];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            if constexpr (KT == 2) {
                const float b1 = xb[row * LPK + 32];
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
            }
        }
        if constexpr (DL) {
            // this wave's DMA of step t+1 has landed and its reads of step t are retired (lgkmcnt(0): the DMA issued after the
            // barrier re-stages that buffer) ... and so for every wave after the barrier
            __builtin_amdgcn_s_waitcnt(0x070 | (2 * NR * (D - 1)));
            __builtin_amdgcn_s_barrier();
            cur = cur == NB - 1 ? 0 : cur + 1;
            nxt = nxt == NB - 1 ? 0 : nxt + 1;
        } else if constexpr (NB == 2) {
            store(cur ^ 1);
            __syncthreads();
            cur ^= 1;
        } else {
            __syncthreads();               // every wave is done reading the buffer ...
            store(0);                      // ... before the next step overwrites it
            __syncthreads();
        }
    }

    if constexpr (DL) __builtin_amdgcn_s_waitcnt(0xF70);      // the past-the-end prefetches (zeros) are still landing in LDS
    // D[i = n][j = kk]: col = lane&31 -> kk, row = (r&3) + 8*(r>>2) + 4*khalf -> n
    float* wsb = p.ws + (size_t)ms * p.Cout * p.Ktot;
#pragma unroll
    for (int j = 0; j < KT; ++j) {
        const int kcol = k0 + wk * (32 * KT) + j * 32 + fr;
        if (kcol >= p.Ktot) continue;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + wn * 64 + i * 32 + 4 * khalf + (r & 3) + 8 * (r >> 2);
                if (n < p.Cout) wsb[(size_t)n * p.Ktot + kcol] = acc[i][j][r];
            }
        }
    }
}

// DMA-staged weight gradient (`lds_buffers` 22 / 23 / 24), Cin % 32 == 0, n tile 128.  Same products in the same order as
// conv_wgrad_f32 -- for one `msplit` the same bits -- but the operand stream is organised so that the pixel loop carries almost no
// vector arithmetic: VALU instructions take issue slots from the MFMAs (conv_common.h), and conv_wgrad_f32 spends 160-180 of them
// per 64 MFMAs on the coordinates of FOUR staging pixels per thread (0.57 MFMA-busy).  Here
//   * a DMA instruction (64 lanes x 16 B = 1 KB of LDS) covers 8 pixels x one 32-channel group, so every lane works on ONE pixel of
//     the step and its NR loads per operand differ only in the channel group: the LDS image is [stage][group 0-3][pixel][32];
//   * dY is linear in the pixel index: lane offset fixed for the whole slice, the step in the load's SGPR offset -- no VALU at all;
//   * X: one pixel decomposition per lane and step; the filter tap of a channel group is block-uniform and rides in the SGPR offset
//     (the descriptor's base is moved back by the padding so that the lane offset is never negative);
//     padding taps / rows past the slice / channel groups past K read zeros through an out-of-range lane offset;
//   * the ring is unrolled by its depth: fragment reads and DMA destinations are register + immediate.
// TBN = 64 (layers with <= 64 output channels): two dY groups, waves 1 x 4 of 64(n) x 32(kk) as in conv_wgrad_f32<., 64>.
template <int NB, int TM, int TBN = 128>
__global__ __launch_bounds__(256) void conv_wgrad_dma(const WgradP p) {
    constexpr int PBK = TM / 8;                  // 8-pixel blocks per step
    constexpr int WPB = 4 / PBK;                 // waves per pixel block
    constexpr int NGY = TBN / 32;                // 32-channel groups of the dY tile (the X tile always has 4)
    constexpr int NRY = NGY / WPB, NRX = 4 / WPB;    // DMA instructions per step and wave: dY, X
    constexpr int KT = TBN == 128 ? 2 : 1;       // 32-wide kk blocks per wave
    static_assert(NRY >= 1, "64-wide n tile: 32-pixel steps only");
    constexpr int GS = TM * 32;                  // floats of one (stage, channel group): [pixel][32]
    constexpr int STGY = NGY * GS, STGX = 4 * GS;    // floats per stage
    constexpr int D = NB - 1;                    // prefetch distance in pixel steps
    static_assert((NRY + NRX) * (D > 0 ? D - 1 : 0) < 16, "vmcnt field");
    constexpr int WAIT = 0x070 | ((NRY + NRX) * (D - 1));       // vmcnt((NRY + NRX)(D-1)) lgkmcnt(0)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Ys = smem;                            // [NB][NGY][TM][32]
    float* Xs = smem + NB * STGY;                // [NB][4][TM][32]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = TBN == 128 ? (wave >> 1) : 0, wk = TBN == 128 ? (wave & 1) : wave;

    int id = ym_xcd_remap(blockIdx.x, gridDim.x);
    const int ms = id % p.msplit;
    id /= p.msplit;
    const int tile_n = id / p.tiles_k, tile_k = id - tile_n * p.tiles_k;
    const int n0 = tile_n * TBN, k0 = tile_k * TBK;
    const int m_beg = ms * p.m_per_split, m_end = min(p.M, m_beg + p.m_per_split);

    // this wave's DMA instructions: pixel block `pbk` of the step, dY groups g0y + j (j < NRY), X groups g0x + j (j < NRX)
    const int pbk = wave % PBK, g0y = (wave / PBK) * NRY, g0x = (wave / PBK) * NRX;
    const int prow = 8 * pbk + (lane >> 3), c4 = lane & 7;
    const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc((void*)p.dy, 0, p.dy_bytes, 0x00020000);
    // X: the base is moved back by the padding, the lane offset addresses pixel (oh*s, ow*s), the tap adds (kh*W + kw) pixels
    const long long padoff = (long long)(p.pad * p.W + p.pad) * p.Cinp * 4;
    // (the range check compares lane offset + SGPR offset with the record count: the count grows by what the base moved back)
    const __amdgpu_buffer_rsrc_t rs_x =
        __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.x - padoff), 0, p.x_bytes + (unsigned)padoff, 0x00020000);
    unsigned yv[NRY], kmask[NRX];
    int xs_off[NRX], khp[NRX], kwp[NRX];         // block-uniform per channel group: byte offset of its tap + channel, tap - pad
    bool plain = p.KH * p.KW == 1 && p.pad == 0; // no bounds test, no masks: the lane offset is used as it is
#pragma unroll
    for (int j = 0; j < NRY; ++j) {
        const int ncol = n0 + (g0y + j) * 32 + c4 * 4;
        yv[j] = (unsigned)((m_beg + prow) * p.Cout + ncol) * 4u | (ncol < p.Cout ? 0u : OOB);
    }
#pragma unroll
    for (int j = 0; j < NRX; ++j) {
        const int kk = k0 + (g0x + j) * 32;      // (Cin % 32 == 0: a group never straddles two taps, K is a multiple of 32)
        const bool k_ok = kk < p.Ktot;
        const int tap = k_ok ? kk / p.Cinp : 0, ci = kk - tap * p.Cinp;
        const int kh = tap / p.KW, kw = tap - kh * p.KW;
        xs_off[j] = k_ok ? ((kh * p.W + kw) * p.Cinp + ci) * 4 : 0;
        khp[j] = kh - p.pad; kwp[j] = kw - p.pad;
        kmask[j] = k_ok ? 0u : OOB;
        plain = plain && k_ok;
    }
    typedef __attribute__((address_space(3))) void* lds_ptr;
    // (`stage` is a literal at every call site and the lambdas are inlined)
    auto load = [&](int mt, int stage) __attribute__((always_inline)) {
        float* yd = Ys + ((stage * NGY + g0y) * TM + 8 * pbk) * 32;       // scalar; lane l lands at +16 l bytes = pixel l / 8, slot l % 8
        float* xd = Xs + ((stage * 4 + g0x) * TM + 8 * pbk) * 32;
        const int ystep = (int)((unsigned)((mt - m_beg) * p.Cout) * 4u);
        const unsigned m = (unsigned)(mt + prow);
        const unsigned b = fastdiv(m, p.mg_howo, p.sh_howo);
        const unsigned rem = m - b * p.HoWo;
        const unsigned oh = fastdiv(rem, p.mg_wo, p.sh_wo);
        const unsigned ow = rem - oh * p.Wo;
        const int ih0 = (int)oh * p.stride, iw0 = (int)ow * p.stride;
        const unsigned xb = (unsigned)((((int)b * p.H + ih0) * p.W + iw0) * p.Cinp) * 4u + (unsigned)(c4 * 16);   // (elements < 2^31, bytes < 2^32)
        if (mt + TM <= m_end && plain) {         // block-uniform: the common step of a 1x1 convolution
#pragma unroll
            for (int j = 0; j < NRY; ++j)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_y, (lds_ptr)(yd + j * GS), 16, (int)yv[j], ystep, 0, 0);
#pragma unroll
            for (int j = 0; j < NRX; ++j) {
                const int xo = xs_off[j];        // (a copy: hipcc's host pass silently drops the kernel when an array ELEMENT is passed here)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_ptr)(xd + j * GS), 16, (int)xb, xo, 0, 0);
            }
        } else {
            const unsigned dead = (int)m < m_end ? 0u : OOB;              // rows past the slice (its last step; the run-ahead loads)
            unsigned inside = 0u;
#pragma unroll
            for (int j = 0; j < NRY; ++j)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_y, (lds_ptr)(yd + j * GS), 16, (int)(yv[j] | dead), ystep, 0, 0);
#pragma unroll
            for (int j = 0; j < NRX; ++j) {
                if (j == 0 || khp[j] != khp[j - 1] || kwp[j] != kwp[j - 1]) {      // block-uniform: a new tap
                    const int ih = ih0 + khp[j], iw = iw0 + kwp[j];
                    inside = ((unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W) ? 0u : OOB;
                }
                const int xo = xs_off[j];
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_ptr)(xd + j * GS), 16, (int)(xb | inside | kmask[j] | dead), xo, 0, 0);
            }
        }
    };

    f32x16 acc[2][KT];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < KT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int fr = lane & 31, khalf = lane >> 5;
    const float* ya = Ys + (wn * 2 * TM + khalf) * 32 + fr;             // pixel row 2s + khalf of groups 2 wn, 2 wn + 1
    const float* xf = Xs + (wk * KT * TM + khalf) * 32 + fr;            // ... of groups KT wk (, KT wk + 1)
    auto compute = [&](int stage) __attribute__((always_inline)) {
#pragma unroll
        for (int s = 0; s < TM / 2; ++s) {
            const float a0 = ya[stage * STGY + 2 * s * 32], a1 = ya[stage * STGY + GS + 2 * s * 32];
            const float b0 = xf[stage * STGX + 2 * s * 32];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            if constexpr (KT == 2) {
                const float b1 = xf[stage * STGX + GS + 2 * s * 32];
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
            }
        }
    };

    static_for<0, D>([&](auto SC) __attribute__((always_inline)) { load(m_beg + decltype(SC)::value * TM, decltype(SC)::value); });
    __builtin_amdgcn_s_waitcnt(WAIT);                                   // step 0 has landed
    __builtin_amdgcn_s_barrier();
    auto step = [&](auto SC, int mt) __attribute__((always_inline)) {
        constexpr int cur = decltype(SC)::value;
        load(mt + D * TM, (cur + D) % NB);                              // into the stage step t-1 was read from
        compute(cur);
        // this wave's DMA of step t+1 has landed and its reads of step t are retired (lgkmcnt(0): the DMA issued after the barrier
        // re-stages that buffer) ... and so for every wave after the barrier
        __builtin_amdgcn_s_waitcnt(WAIT);
        __builtin_amdgcn_s_barrier();
    };
    int mt = m_beg;
    for (; mt + (NB - 1) * TM < m_end; mt += NB * TM)
        static_for<0, NB>([&](auto SC) __attribute__((always_inline)) { step(SC, mt + decltype(SC)::value * TM); });
    static_for<0, NB - 1>([&](auto SC) __attribute__((always_inline)) {
        if (mt + decltype(SC)::value * TM < m_end) step(SC, mt + decltype(SC)::value * TM);
    });
    __builtin_amdgcn_s_waitcnt(0xF70);           // the past-the-end prefetches (zeros) are still landing in LDS

    // D[i = n][j = kk]: col = lane&31 -> kk, row = (r&3) + 8*(r>>2) + 4*khalf -> n
    float* wsb = p.ws + (size_t)ms * p.Cout * p.Ktot;
#pragma unroll
    for (int j = 0; j < KT; ++j) {
        const int kcol = k0 + wk * (32 * KT) + j * 32 + fr;
        if (kcol >= p.Ktot) continue;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + wn * 64 + i * 32 + 4 * khalf + (r & 3) + 8 * (r >> 2);
                if (n < p.Cout) wsb[(size_t)n * p.Ktot + kcol] = acc[i][j][r];
            }
        }
    }
}

// sum the msplit slabs in order and scatter [n][tap][ci] -> OIHW [n][ci][kh][kw] (ci < Cin_real); rows may be routed to up to three
// OIHW tensors (ym_wgrad_desc.row_end / dw_seg) and added to what is there (accumulate)
struct WOut { float* dw[3]; int row_end[3]; int accumulate; };
// (32-bit indices and multiply-shift divisions: this kernel runs on the side stream UNDER the data-gradient MFMA kernels, and its
// vector instructions -- two integer divisions per element were ~80 of them -- take issue slots from their MFMAs)
__global__ __launch_bounds__(256) void wgrad_reduce_unpack(const float* __restrict__ ws, const WOut o, int msplit,
                                                           int Cout_real, int Cout, int Ktot, int Cinp, int Cin_real, int KHW,
                                                           FastDiv fd_ktot4, FastDiv fd_cinp4) {
    // one thread = four consecutive input channels of one (output channel, filter tap): 16-byte loads from every slab, four slabs
    // in flight (round 3: one float per thread and a dependent add per slab -- 1.7 ms of launches per step for ~0.5 ms of bytes)
    const unsigned ktot4 = (unsigned)Ktot >> 2, total = (unsigned)Cout_real * ktot4;
    const size_t slab = (size_t)Cout * Ktot;
    for (unsigned q = blockIdx.x * 256u + threadIdx.x; q < total; q += gridDim.x * 256u) {
        unsigned un, ukk4, utap, uci4;
        fd_ktot4.divmod(q, un, ukk4);
        fd_cinp4.divmod(ukk4, utap, uci4);
        const int n = (int)un, tap = (int)utap, ci = (int)uci4 * 4;
        if (ci >= Cin_real) continue;
        const float* src = ws + (size_t)n * Ktot + (size_t)ukk4 * 4;
        f32x4 v = *reinterpret_cast<const f32x4*>(src);
        int s = 1;
        for (; s + 3 < msplit; s += 4) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(src + (size_t)s * slab), b = *reinterpret_cast<const f32x4*>(src + (size_t)(s + 1) * slab);
            const f32x4 c = *reinterpret_cast<const f32x4*>(src + (size_t)(s + 2) * slab), d = *reinterpret_cast<const f32x4*>(src + (size_t)(s + 3) * slab);
            v += a; v += b; v += c; v += d;
        }
        for (; s < msplit; ++s) v += *reinterpret_cast<const f32x4*>(src + (size_t)s * slab);
        const int seg = n < o.row_end[0] ? 0 : (n < o.row_end[1] ? 1 : 2);
        const int n_local = n - (seg == 0 ? 0 : o.row_end[seg - 1]);
        float* dst = o.dw[seg] + ((size_t)n_local * Cin_real + ci) * KHW + tap;
        if (KHW == 1 && (Cin_real & 3) == 0 && ((uintptr_t)dst & 15) == 0) {       // a 1x1 filter: OIHW rows are contiguous in ci
            f32x4 w = v;
            if (o.accumulate) w += *reinterpret_cast<const f32x4*>(dst);
            *reinterpret_cast<f32x4*>(dst) = w;
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (ci + e < Cin_real) {
                    float* d1 = dst + (size_t)e * KHW;
                    *d1 = o.accumulate ? *d1 + v[e] : v[e];
                }
            }
        }
    }
}

// The same reduction for MANY weight gradients in one launch (ym_wgrad_reduce_batch): a res101 step has 104 backbone weight
// gradients, each followed by its own 5-20 us reduce launch; their slabs stay in per-layer scratch instead and one launch per
// gradient bucket reduces them all.  Block b belongs to the item with first_block <= b < first_block + blocks (binary search over
// the table, block-uniform); thread = four input channels of one (output channel, tap), exactly wgrad_reduce_unpack's unit, slabs
// summed in the same order -> the same bits.
struct WRItemDev {
    const float* ws; float* dw;
    unsigned first_block, blocks, quads, slab_hi;      // slab = floats per slab (lo | hi << 32)
    unsigned slab_lo; int msplit, Ktot, Cin_real, KHW, accumulate;
    FastDiv fd_ktot4, fd_cinp4;
};
static_assert(sizeof(WRItemDev) == sizeof(ym_wgrad_reduce_item), "ym_wgrad_reduce_item must mirror WRItemDev");

__global__ __launch_bounds__(256) void wgrad_reduce_batch(const WRItemDev* __restrict__ items, int n_items) {
    int lo = 0, hi = n_items - 1;
    const unsigned b = blockIdx.x;
    while (lo < hi) {                                   // last item with first_block <= b
        const int mid = (lo + hi + 1) >> 1;
        if (items[mid].first_block <= b) lo = mid; else hi = mid - 1;
    }
    const WRItemDev it = items[lo];
    const unsigned q = (b - it.first_block) * 256u + threadIdx.x;
    if (q >= it.quads) return;
    unsigned un, ukk4, utap, uci4;
    it.fd_ktot4.divmod(q, un, ukk4);
    it.fd_cinp4.divmod(ukk4, utap, uci4);
    const int n = (int)un, tap = (int)utap, ci = (int)uci4 * 4;
    if (ci >= it.Cin_real) return;
    const size_t slab = (size_t)it.slab_lo | ((size_t)it.slab_hi << 32);
    const float* src = it.ws + (size_t)n * it.Ktot + (size_t)ukk4 * 4;
    f32x4 v = *reinterpret_cast<const f32x4*>(src);
    int s = 1;
    for (; s + 3 < it.msplit; s += 4) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(src + (size_t)s * slab), b4 = *reinterpret_cast<const f32x4*>(src + (size_t)(s + 1) * slab);
        const f32x4 c = *reinterpret_cast<const f32x4*>(src + (size_t)(s + 2) * slab), d = *reinterpret_cast<const f32x4*>(src + (size_t)(s + 3) * slab);
        v += a; v += b4; v += c; v += d;
    }
    for (; s < it.msplit; ++s) v += *reinterpret_cast<const f32x4*>(src + (size_t)s * slab);
    float* dst = it.dw + ((size_t)n * it.Cin_real + ci) * it.KHW + tap;
    if (it.KHW == 1 && (it.Cin_real & 3) == 0 && ((uintptr_t)dst & 15) == 0) {
        f32x4 w = v;
        if (it.accumulate) w += *reinterpret_cast<const f32x4*>(dst);
        *reinterpret_cast<f32x4*>(dst) = w;
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (ci + e < it.Cin_real) {
                float* d1 = dst + (size_t)e * it.KHW;
                *d1 = it.accumulate ? *d1 + v[e] : v[e];
            }
        }
    }
}

struct WPlan { int M, Ktot, tiles_n, tiles_k, msplit, m_per_split, tbn, nb; };

void fastdiv_make(unsigned d, unsigned* mg, unsigned* sh) {
    unsigned s = 0;
    while ((1ull << s) < d) ++s;
    *sh = s;
    *mg = (unsigned)(((1ull << 32) * ((1ull << s) - d)) / d + 1);
}

int wplan(const ym_wgrad_desc* d, WPlan* pl) {
    YM_REQUIRE(d && d->x && d->dy && d->dw, "wgrad: null pointer");
    YM_REQUIRE(d->Cin == 4 || d->Cin % 4 == 0, "wgrad: Cin (padded) must be a multiple of 4");
    YM_REQUIRE(d->Cout % 4 == 0, "wgrad: dy channel pitch must be a multiple of 4 (pad it)");
    YM_REQUIRE(d->Cin_real >= 1 && d->Cin_real <= d->Cin && d->Cout_real >= 1 && d->Cout_real <= d->Cout, "wgrad: real channel counts");
    YM_REQUIRE(d->Ho == (d->H + 2 * d->pad - d->KH) / d->stride + 1 && d->Wo == (d->W + 2 * d->pad - d->KW) / d->stride + 1,
               "wgrad: Ho/Wo inconsistent");
    const long long M = (long long)d->B * d->Ho * d->Wo;
    YM_REQUIRE(M * d->Cout < (1ll << 31) && (long long)d->B * d->H * d->W * d->Cin < (1ll << 31), "wgrad: tensor too large");
    pl->M = (int)M;
    pl->Ktot = d->KH * d->KW * d->Cin;
    pl->tbn = d->Cout_real <= 64 ? 64 : 128;
    pl->nb = d->lds_buffers == 1 ? 1 : 2;
    if (d->lds_buffers == 22 || d->lds_buffers == 23 || d->lds_buffers == 24) {
        YM_REQUIRE(pl->tbn == 128 || d->lds_buffers == 22, "wgrad: layers with <= 64 output channels have the 32-pixel DMA ring (lds_buffers 22) only");
        // conv_wgrad_dma: whole 32-channel groups per filter tap (and a padding the usual convolutions have); anything else runs
        // on the register-staged double buffer
        const bool dma_ok = d->Cin % 32 == 0 && 2 * d->pad <= d->KH - 1 && 2 * d->pad <= d->KW - 1 &&
                            (unsigned long long)d->B * d->H * d->W * d->Cin * 4ull + (unsigned long long)(d->pad * d->W + d->pad) * d->Cin * 4ull < 0xFFFFFFF0ull;
        pl->nb = dma_ok ? d->lds_buffers : 2;
    }
    pl->tiles_n = ym_cdiv(d->Cout, pl->tbn);
    pl->tiles_k = ym_cdiv(pl->Ktot, TBK);
    int ms = d->msplit;
    if (ms <= 0) {
        const int tiles = pl->tiles_n * pl->tiles_k;
        ms = ym_cdiv(768, tiles);
        const int max_ms = ym_cdiv(pl->M, 8 * TBM);      // at least 8 pixel steps per slice
        if (ms > max_ms) ms = max_ms;
        if (ms > 256) ms = 256;
        if (ms < 1) ms = 1;
    }
    int per = ym_cdiv(pl->M, ms);
    per = ym_cdiv(per, TBM) * TBM;
    pl->m_per_split = per;
    pl->msplit = ym_cdiv(pl->M, per);
    return YM_OK;
}

}  // namespace

extern "C" size_t ym_conv2d_wgrad_workspace_bytes(const ym_wgrad_desc* d) {
    WPlan pl;
    if (wplan(d, &pl) != YM_OK) return 0;
    return (size_t)pl.msplit * d->Cout * pl.Ktot * sizeof(float);
}

namespace {
// first pass: msplit partial gradients [msplit][Cout][Ktot] into the workspace
int wgrad_slabs(const ym_wgrad_desc* d, const WPlan& pl, void* workspace, size_t workspace_bytes, ym_stream_t s) {
    int rc = YM_OK;
    const size_t need = (size_t)pl.msplit * d->Cout * pl.Ktot * sizeof(float);
    if (!workspace || workspace_bytes < need) { ym_set_error("wgrad: workspace %zu B < %zu B", workspace_bytes, need); return YM_ENOSPC; }
    WgradP p;
    p.x = d->x; p.dy = d->dy; p.ws = (float*)workspace;
    p.B = d->B; p.H = d->H; p.W = d->W; p.Cinp = d->Cin; p.Cout = d->Cout; p.KH = d->KH; p.KW = d->KW;
    p.stride = d->stride; p.pad = d->pad; p.Ho = d->Ho; p.Wo = d->Wo;
    p.M = pl.M; p.HoWo = d->Ho * d->Wo; p.Ktot = pl.Ktot; p.tiles_n = pl.tiles_n; p.tiles_k = pl.tiles_k;
    p.msplit = pl.msplit; p.m_per_split = pl.m_per_split;
    {
        const unsigned long long xb = (unsigned long long)d->B * d->H * d->W * d->Cin * 4ull, yb = (unsigned long long)pl.M * d->Cout * 4ull;
        YM_REQUIRE(xb < 0xFFFFFFF0ull && yb < 0xFFFFFFF0ull, "wgrad: x / dy must be < 4 GiB (32-bit buffer offsets)");
        p.x_bytes = (unsigned)xb; p.dy_bytes = (unsigned)yb;
        fastdiv_make((unsigned)p.HoWo, &p.mg_howo, &p.sh_howo);
        fastdiv_make((unsigned)d->Wo, &p.mg_wo, &p.sh_wo);
    }
    hipStream_t st = (hipStream_t)s;
    // 22: DMA, 32 pixels x ring of 2 (64 KB); 23: DMA, 16 pixels x ring of 3 (48 KB); 24: DMA, 16 pixels x ring of 4 (64 KB)
    const size_t lds = pl.nb == 22 ? (size_t)2 * 32 * (pl.tbn + 128) * 4 : pl.nb == 23 ? (size_t)2 * 3 * 16 * DP * 4
                     : pl.nb == 24 ? (size_t)2 * 4 * 16 * DP * 4 : (size_t)2 * pl.nb * TBM * LP * sizeof(float);
    {
        // every instantiation gets the same ceiling, once per device (ym_ensure_dyn_lds)
        const size_t big = (size_t)4 * TBM * LP * sizeof(float);
        static YmLdsAttr attrs[12] = {};
        int ai = 0, rc = YM_OK;
#define YM_WG_ATTR(I, T, N) if (!rc) rc = ym_ensure_dyn_lds(attrs[ai++], reinterpret_cast<const void*>(conv_wgrad_f32<I, T, N>), big, "conv_wgrad_f32")
        YM_WG_ATTR(0, 128, 2); YM_WG_ATTR(1, 128, 2); YM_WG_ATTR(0, 64, 2); YM_WG_ATTR(1, 64, 2);
        YM_WG_ATTR(0, 128, 1); YM_WG_ATTR(1, 128, 1); YM_WG_ATTR(0, 64, 1); YM_WG_ATTR(1, 64, 1);
#undef YM_WG_ATTR
#define YM_WG_ATTR_DL(...) if (!rc) rc = ym_ensure_dyn_lds(attrs[ai++], reinterpret_cast<const void*>(conv_wgrad_dma<__VA_ARGS__>), big, "conv_wgrad_dma")
        YM_WG_ATTR_DL(2, 32); YM_WG_ATTR_DL(3, 16); YM_WG_ATTR_DL(4, 16); YM_WG_ATTR_DL(2, 32, 64);
#undef YM_WG_ATTR_DL
        if (rc) return rc;
    }
    // incremental coordinates need: (rows advanced per step) + 1 < 2 * Ho, so that two conditional wraps suffice
    p.incr = (TBM / d->Wo + 2 <= 2 * d->Ho) ? 1 : 0;
    const dim3 wgrid(pl.tiles_n * pl.tiles_k * pl.msplit);
#define YM_WG_LAUNCH(T, N)                                                                          \
    do {                                                                                            \
        if (p.incr) hipLaunchKernelGGL((conv_wgrad_f32<1, T, N>), wgrid, dim3(256), lds, st, p);   \
        else hipLaunchKernelGGL((conv_wgrad_f32<0, T, N>), wgrid, dim3(256), lds, st, p);          \
    } while (0)
#define YM_WG_LAUNCH_DL(N, T) hipLaunchKernelGGL((conv_wgrad_dma<N, T>), wgrid, dim3(256), lds, st, p)
    if (pl.nb == 22 && pl.tbn == 64) hipLaunchKernelGGL((conv_wgrad_dma<2, 32, 64>), wgrid, dim3(256), lds, st, p);
    else if (pl.nb == 22) YM_WG_LAUNCH_DL(2, 32);
    else if (pl.nb == 23) YM_WG_LAUNCH_DL(3, 16);
    else if (pl.nb == 24) YM_WG_LAUNCH_DL(4, 16);
    else if (pl.tbn == 64) { if (pl.nb == 1) YM_WG_LAUNCH(64, 1); else YM_WG_LAUNCH(64, 2); }
    else { if (pl.nb == 1) YM_WG_LAUNCH(128, 1); else YM_WG_LAUNCH(128, 2); }
#undef YM_WG_LAUNCH
#undef YM_WG_LAUNCH_DL
    rc = ym_check_launch("conv_wgrad_f32");
    return rc;
}
}  // namespace

extern "C" int ym_conv2d_wgrad(const ym_wgrad_desc* d, void* workspace, size_t workspace_bytes, ym_stream_t s) {
    WPlan pl;
    int rc = wplan(d, &pl);
    if (rc != YM_OK) return rc;
    rc = wgrad_slabs(d, pl, workspace, workspace_bytes, s);
    if (rc != YM_OK) return rc;
    hipStream_t st = (hipStream_t)s;
    const size_t total = (size_t)d->Cout_real * pl.Ktot;
    YM_REQUIRE(pl.Ktot % 4 == 0 && d->Cin % 4 == 0 && ((uintptr_t)workspace & 15) == 0, "wgrad: reduce needs K %% 4 == 0 and a 16-byte aligned workspace");
    int grid = (int)((total / 4 + 255) / 256);
    if (grid > 4096) grid = 4096;
    if (grid < 1) grid = 1;
    WOut o;
    o.accumulate = d->accumulate ? 1 : 0;
    if (d->row_end[0] > 0) {
        YM_REQUIRE(d->row_end[0] < d->row_end[1] && d->row_end[1] < d->Cout_real && d->dw_seg[0] && d->dw_seg[1], "wgrad: bad output segments");
        o.dw[0] = d->dw; o.dw[1] = d->dw_seg[0]; o.dw[2] = d->dw_seg[1];
        o.row_end[0] = d->row_end[0]; o.row_end[1] = d->row_end[1]; o.row_end[2] = d->Cout_real;
    } else {
        o.dw[0] = d->dw; o.dw[1] = d->dw; o.dw[2] = d->dw;
        o.row_end[0] = d->Cout_real; o.row_end[1] = d->Cout_real; o.row_end[2] = d->Cout_real;
    }
    YM_REQUIRE(total < (1ull << 31), "wgrad: gradient tensor too large for 32-bit indexing");
    hipLaunchKernelGGL(wgrad_reduce_unpack, dim3(grid), dim3(256), 0, st, (const float*)workspace, o, pl.msplit, d->Cout_real,
                       d->Cout, pl.Ktot, d->Cin, d->Cin_real, d->KH * d->KW, FastDiv::make((unsigned)pl.Ktot / 4), FastDiv::make((unsigned)d->Cin / 4));
    return ym_check_launch("wgrad_reduce_unpack");
}

extern "C" int ym_conv2d_wgrad_slabs(const ym_wgrad_desc* d, void* workspace, size_t workspace_bytes, ym_wgrad_reduce_item* item,
                                     ym_stream_t s) {
    WPlan pl;
    int rc = wplan(d, &pl);
    if (rc != YM_OK) return rc;
    YM_REQUIRE(item, "wgrad(slabs): null item");
    YM_REQUIRE(d->row_end[0] == 0, "wgrad(slabs): a gradient routed to several tensors takes ym_conv2d_wgrad");
    YM_REQUIRE(pl.Ktot % 4 == 0 && d->Cin % 4 == 0 && ((uintptr_t)workspace & 15) == 0, "wgrad: reduce needs K %% 4 == 0 and a 16-byte aligned workspace");
    const size_t quads = (size_t)d->Cout_real * pl.Ktot / 4;
    YM_REQUIRE(quads < (1ull << 31), "wgrad: gradient tensor too large for 32-bit indexing");
    rc = wgrad_slabs(d, pl, workspace, workspace_bytes, s);
    if (rc != YM_OK) return rc;
    WRItemDev it = {};
    it.ws = (const float*)workspace; it.dw = d->dw;
    it.first_block = 0; it.blocks = (unsigned)((quads + 255) / 256); it.quads = (unsigned)quads;
    const size_t slab = (size_t)d->Cout * pl.Ktot;
    it.slab_lo = (unsigned)(slab & 0xFFFFFFFFull); it.slab_hi = (unsigned)(slab >> 32);
    it.msplit = pl.msplit; it.Ktot = pl.Ktot; it.Cin_real = d->Cin_real; it.KHW = d->KH * d->KW; it.accumulate = d->accumulate ? 1 : 0;
    it.fd_ktot4 = FastDiv::make((unsigned)pl.Ktot / 4); it.fd_cinp4 = FastDiv::make((unsigned)d->Cin / 4);
    *reinterpret_cast<WRItemDev*>(item) = it;
    return YM_OK;
}

extern "C" int ym_wgrad_reduce_batch(const ym_wgrad_reduce_item* items_dev, int n_items, unsigned total_blocks, ym_stream_t s) {
    YM_REQUIRE(items_dev && n_items > 0 && total_blocks > 0, "wgrad(reduce batch): empty table");
    hipLaunchKernelGGL(wgrad_reduce_batch, dim3(total_blocks), dim3(256), 0, (hipStream_t)s, (const WRItemDev*)items_dev, n_items);
    return ym_check_launch("wgrad_reduce_batch");
}
