// Fused convolution for gfx950: NHWC fp32 implicit GEMM on the f32 MFMA pipe
// (v_mfma_f32_32x32x2_f32), BN scale/shift + bias + residual + ReLU/tanh in the epilogue, up to three
// output segments so the YOLACT head (conf|bbox|coef) is one launch writing straight into the
// concatenated [B, N, C] tensors.
//
// GEMM view:  C[m][n] = sum_k A[m][k] * W[n][k]
//   m = (b, oh, ow)            M = B*Ho*Wo       (output pixels, NHWC row index)
//   n = output channel         N = Cout
//   k = (kh, kw, cin)          K = KH*KW*Cin     (both operands are K-contiguous in HBM:
//                                                 A row = Cin floats of one input pixel per tap,
//                                                 W row = ym_pack_conv_weight image)
// Workgroup = 256 threads = 4 waves (2x2), tile BM x BN, K step 32 floats (128 B per row).
// LDS image: [rows][36] floats (pitch 144 B = 9 x 16 B -> ds_read_b128 of 16 consecutive rows hits 16
// distinct 16-B slots of the 256-B bank row: conflict-free, cdna guide §2 / G4).
// MFMA operand trick: 32x32x2 wants lane l to supply A[i=l&31][k=l>>5]; since the order of the K
// sum is free, lane half h takes k = 8g+4h+s for step s of group g, so ONE ds_read_b128 per operand row
// feeds four MFMA steps (both operands use the same k assignment, so products pair up correctly).
// Per K-tile a wave issues 4*(TM+TN) ds_read_b128 for 16*TM*TN MFMAs of 64 cycles each: the kernel
// is MFMA-issue bound, LDS and the global->LDS staging (register prefetch, double-buffered) sit
// far below their limits.  Grid: one block per (tile, k-slice), XCD-aware remap so the N-tiles
// sharing an A panel run on the same XCD L2.
#include <stdlib.h>
#include <type_traits>
#include "conv_common.h"

using namespace ymk;

namespace {

constexpr int PITCH = 36;  // floats

// MODE 0: Cin % 32 == 0 (every K tile lies inside one filter tap).  MODE 1: Cin == 4 (stem; one tap per float4).
// MODE 2: data gradient (transposed conv): output pixel (ih,iw) gathers dY[(ih+pad-kh)/s][(iw+pad-kw)/s] where divisible.
// NS: LDS ring depth.  2 = load(t+1) overlaps compute(t).  3 = loads run TWO tiles ahead (small tiles with one workgroup per
// CU measured 2250 cycles per K tile against 1024 cycles of MFMA with NS=2: one L2 round trip was exposed per tile).
// DL ("direct to LDS"): operand tiles go global -> LDS with `buffer_load_dwordx4 ... lds` (no staging registers, no ds_write);
// the LDS image is then [rows][32] floats UNPADDED (the DMA places lane l's 16 bytes at base + 16*l) and bank conflicts are
// avoided by an XOR swizzle of the 16-byte chunks instead: chunk c of row r lives in slot c ^ ((r >> 1) & 7), so the 16 rows a
// ds_read_b128 phase touches cover all 16 distinct (row parity, slot) bank groups.  The ring is NS deep with loads NS-1 tiles
// ahead; the waits are explicit `s_waitcnt vmcnt(n)` + `s_barrier` (a __syncthreads() would drain every outstanding DMA).
// PF (with DL): software-pipelined fragments.  The LDS -> register reads of K group g+1 are issued before the MFMAs of group g
// and the reads of the NEXT tile's first group before the end-of-tile barrier (tile t+1 is required to have landed one barrier
// early: ring of NS >= 3 with NS-2 tiles still in flight), so the dependent MFMA chain of a wave never waits on LDS or on the
// barrier round trip: measured 1500 -> ~1100 cycles per K tile for a workgroup that has its CU to itself (the bs=1 regime).
// RG: pyramid input (ym_conv_desc.nlevels): every staging row carries its own map size.
// SPL ("split bf16", ym_conv_desc.mma = 3 / 6): the products run on the bf16 MFMA (v_mfma_f32_32x32x16_bf16, 16x the f32 pipe's
// rate) with every fp32 operand split, while it is staged into LDS, into SPL bf16 planes x = p0 + p1 (+ p2)
// (p0 = bf16(x), p1 = bf16(x - p0), ...: 16 resp. 24 significant bits) and the cross terms of significance >= 2^-16 resp. 2^-24
// summed in the fp32 accumulator: SPL = 2 -> 3 MFMAs per product ("bf16x3", relative error ~2^-17 per product), SPL = 3 -> 6 MFMAs
// ("bf16x6", fp32-grade).  Tensors in HBM stay fp32; only the register-staged double buffer exists in this mode.
template <int BM, int BN, int MODE, int NS, bool DL = false, bool PF = false, bool RG = false, int SPL = 0>
__global__ __launch_bounds__(256) void conv_igemm_f32(const ConvP p) {
    static_assert(SPL == 0 || (SPL <= 3 && !DL && !PF && (NS == 2 || NS == 3) && MODE != 1), "split-bf16: register staging, Cin % 32 == 0");
    constexpr int LB = SPL ? 2 : NS;            // LDS buffers (split mode: always two; NS = 3 there means two REGISTER sets)
    constexpr int TM = BM / 64, TN = BN / 64;   // 32x32 MFMA tiles per wave (wave tile = BM/2 x BN/2)
    constexpr int AR = BM / 32, BR = BN / 32;   // staging rows per thread
    constexpr int BP = 40;                      // split mode: LDS row pitch of one bf16 plane, in bf16 (80 B: 16 rows of a
                                                // ds_read_b128 phase fall into 16 distinct 16-byte bank groups)
    constexpr int RP = SPL ? (SPL * BP) / 2 : (DL ? 32 : PITCH);   // LDS floats per tile row (split: SPL planes of BP bf16, stored plane-major)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                         // [NS][BM][RP]
    float* Bs = smem + LB * BM * RP;          // [LB][BN][RP]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: the DMA's LDS destination goes through M0
    const int wm = wave >> 1, wn = wave & 1;
    YM_STAMP(0);
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, p.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_ws = __builtin_amdgcn_make_buffer_rsrc((void*)p.ws, 0, p.ws_bytes, 0x00020000);

    int id, ks, nks, ktps;                    // output tile, K slice, slices of THIS tile, K tiles per slice (block-uniform)
    {
        unsigned q, r;
        if ((int)blockIdx.x < p.main_blocks) {
            p.fd_ksplit.divmod((unsigned)ym_xcd_remap(blockIdx.x, p.main_blocks), q, r);
            id = (int)q; ks = (int)r;
            nks = p.ksplit; ktps = p.kt_per_split;
        } else {                              // tail tiles, split finer so that the last partial round of workgroups fills the chip
            p.fd_tail.divmod(blockIdx.x - (unsigned)p.main_blocks, q, r);
            id = p.main_tiles + (int)q; ks = (int)r;
            nks = p.tail_split; ktps = p.tail_ktps;
        }
    }
    int tile_m, tile_n;
    ym_tile_decode(p, id, tile_m, tile_n);
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    // stride-2 data gradient: the parity class of this tile's rows (conv_common.h) -- its first GEMM row, its taps, its K range
    int cls = 0, cls_base = 0, c_kh0 = 0, c_kw0 = 0, c_step = 1, c_nkw = 1;
    bool cls_on = false;
    int kt_beg = ks * ktps, kt_end = min(p.nkt, kt_beg + ktps);
    if constexpr (MODE == 2) {
        if (p.cls) {
            cls_on = true;
            // consecutive M tiles cycle through the four classes (every class is padded to the same number of tiles): an XCD's chunk
            // of the tile space then holds every class in equal shares -- class by class, the two XCDs that owned class 0 did ALL the
            // work of a 1x1 stride-2 data gradient (only that class has a tap) and the launch was no faster than before
            cls = tile_m & 3;
            cls_base = m0 - (tile_m >> 2) * BM;            // local row = m - cls_base = (tile_m >> 2) * BM + (m - m0)
            c_kh0 = p.cls_kh0[cls]; c_kw0 = p.cls_kw0[cls]; c_nkw = p.cls_nkw[cls]; c_step = 2;
            const int per = (int)blockIdx.x < p.main_blocks ? p.cls_ktps[cls] : p.cls_tail_ktps[cls];
            kt_beg = ks * per;
            kt_end = min(p.cls_nkt[cls], kt_beg + per);
        }
    }
    // GEMM row -> dx pixel (b, oh, ow); false for the padding rows of a class / rows past M
    auto row_coords = [&](int m, int& b, int& oh, int& ow) __attribute__((always_inline)) -> bool {
        if (MODE == 2 && cls_on) {
            const int local = m - cls_base;
            if (local >= p.cls_rows[cls]) return false;
            unsigned ub, urem, ui, uj;
            p.cls_fd_hw[cls].divmod((unsigned)local, ub, urem);
            p.cls_fd_w[cls].divmod(urem, ui, uj);
            b = (int)ub; oh = 2 * (int)ui + (cls >> 1); ow = 2 * (int)uj + (cls & 1);
            return true;
        }
        if (m >= p.M) return false;
        unsigned ub, urem, uoh, uow;
        p.fd_howo.divmod((unsigned)m, ub, urem);
        p.fd_wo.divmod(urem, uoh, uow);
        b = (int)ub; oh = (int)uoh; ow = (int)uow;
        return true;
    };
    // tile row -> row of the output / residual / BatchNorm tensors ([pixel][Cout]); -1: a row that produces nothing
    auto out_row = [&](int row) __attribute__((always_inline)) -> int {
        const int m = m0 + row;
        if (MODE == 2 && cls_on) {
            int b, oh, ow;
            return row_coords(m, b, oh, ow) ? (b * p.Ho + oh) * p.Wo + ow : -1;
        }
        return m < p.M ? m : -1;
    };

    // ---- per-thread staging coordinates -------------------------------------------------------
    const int c4 = DL ? ((tid & 7) ^ ((tid >> 4) & 7)) : (tid & 7);   // which float4 of the 32-float K row (DL: swizzled)
    const int rbase = tid >> 3;  // 0..31
    int a_pix[AR];               // (b*H + ih0)*W + iw0   (may be negative; only used when in range); MODE 2: b*H
    int a_ih0[AR], a_iw0[AR];
    int a_h[RG ? AR : 1], a_w[RG ? AR : 1];
#pragma unroll
    for (int i = 0; i < AR; ++i) {
        const int m = m0 + rbase + 32 * i;
        if constexpr (MODE == 2) {
            int b, oh, ow;
            if (row_coords(m, b, oh, ow)) {
                a_ih0[i] = oh + p.pad;
                a_iw0[i] = ow + p.pad;
                a_pix[i] = b * p.H;
            } else {
                a_ih0[i] = -(1 << 20);
                a_iw0[i] = -(1 << 20);
                a_pix[i] = 0;
            }
            continue;
        }
        if (m < p.M) {
            if constexpr (RG) {                          // which level does this GEMM row belong to?
                int base = 0, h = p.lev_h[0], w = p.lev_w[0];
#pragma unroll
                for (int q = 1; q < 5; ++q)
                    if (q < p.nlev && m >= p.lev_m[q]) { base = p.lev_m[q]; h = p.lev_h[q]; w = p.lev_w[q]; }
                const int local = m - base, hw = h * w;
                const int b = local / hw, rem = local - b * hw;
                const int oh = rem / w, ow = rem - oh * w;
                a_ih0[i] = oh - p.pad;
                a_iw0[i] = ow - p.pad;
                a_pix[i] = base + (b * h + a_ih0[i]) * w + a_iw0[i];
                a_h[i] = h; a_w[i] = w;
                continue;
            }
            unsigned ub, urem, uoh, uow;
            p.fd_howo.divmod((unsigned)m, ub, urem);
            p.fd_wo.divmod(urem, uoh, uow);
            const int b = (int)ub, oh = (int)uoh, ow = (int)uow;
            if (MODE == 2) {
                a_ih0[i] = oh + p.pad;
                a_iw0[i] = ow + p.pad;
                a_pix[i] = b * p.H;
            } else {
                a_ih0[i] = oh * p.stride - p.pad;
                a_iw0[i] = ow * p.stride - p.pad;
                a_pix[i] = (b * p.H + a_ih0[i]) * p.W + a_iw0[i];
            }
        } else {
            a_ih0[i] = -(1 << 20);
            a_iw0[i] = -(1 << 20);
            a_pix[i] = 0;
            if constexpr (RG) { a_h[i] = 1; a_w[i] = 1; }
        }
    }
    unsigned wrow[BR];           // byte offset of this thread's float4 in weight row n
#pragma unroll
    for (int i = 0; i < BR; ++i) {
        const int n = n0 + rbase + 32 * i;
        wrow[i] = (n < p.Cout) ? (unsigned)((n * p.Kpad + c4 * 4) * 4) : p.w_bytes;      // past Cout: parked at the buffer's end
    }

    // tap walker for MODE 0 (uniform across the block)
    int kh = 0, kw = 0, c0 = 0;
    if (MODE == 0 || MODE == 2) {
        unsigned tap, uc0, ukh, ukw;
        p.fd_cin.divmod((unsigned)(kt_beg * BK), tap, uc0);
        c0 = (int)uc0;
        if (MODE == 2 && cls_on) {                 // the class's taps: kh0, kh0 + 2, ... x kw0, kw0 + 2, ...
            const int a = (int)tap / c_nkw;
            kh = c_kh0 + 2 * a; kw = c_kw0 + 2 * ((int)tap - a * c_nkw);
        } else {
            p.fd_kw.divmod(tap, ukh, ukw);
            kh = (int)ukh; kw = (int)ukw;
        }
    }

    // One K tile of both operands: `sink_a(i, lane_offset, block_offset)` / `sink_b(...)` receive the source of this lane's 16
    // bytes of staging row i as a per-lane byte offset + a block-uniform one (-> the load's SGPR offset).  There is NO per-tile
    // vector arithmetic: everything that depends on the filter tap (bounds test, pixel offset) is refreshed only when the tap
    // changes (every Cin/32 tiles; never for a 1x1 conv), rows that must read zeros carry an all-ones mask in the lane offset
    // (offset | 0xFFFFFFF0 is beyond any buffer -> the raw buffer load returns 0 whatever the SGPR offset adds: no 32-bit wrap), and
    // weight rows past Cout sit at the END of the buffer.  Tiles past the end of this block's K range are fetched like any other
    // (the counted vmcnt waits need an unconditional load count) and never consumed — a buffer load cannot fault.
    unsigned a_off[AR];
    bool tap_dirty = true;
    int w_koff = 0;              // MODE 2: byte offset of the K tile inside a weight row
    auto gather_tile = [&](int kt, auto&& sink_a, auto&& sink_b) __attribute__((always_inline)) {
        if (MODE == 0 || MODE == 2) {
            if (tap_dirty) {                                  // block-uniform
                tap_dirty = false;
#pragma unroll
                for (int i = 0; i < AR; ++i) {
                    if (MODE == 0) {
                        const int ih = a_ih0[i] + kh, iw = a_iw0[i] + kw;
                        const int Hh = RG ? a_h[RG ? i : 0] : p.H, Ww = RG ? a_w[RG ? i : 0] : p.W;
                        const bool ok = (unsigned)ih < (unsigned)Hh && (unsigned)iw < (unsigned)Ww;
                        a_off[i] = (unsigned)(((a_pix[i] + kh * Ww + kw) * p.Cin + c4 * 4) * 4) | (ok ? 0u : OOB);
                    } else {
                        const int sh = p.stride >> 1, smask = p.stride - 1;     // stride is 1 or 2
                        const int th = a_ih0[i] - kh, tw = a_iw0[i] - kw;
                        const int yh = th >> sh, yw = tw >> sh;
                        const bool ok = th >= 0 && tw >= 0 && ((th | tw) & smask) == 0 && yh < p.H && yw < p.W;
                        a_off[i] = (unsigned)((((a_pix[i] + yh) * p.W + yw) * p.Cin + c4 * 4) * 4) | (ok ? 0u : OOB);
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < AR; ++i) sink_a(i, a_off[i], c0 * 4);
            if constexpr (MODE == 2) w_koff = ((kh * p.KW + kw) * p.Cin + c0) * 4;      // (a class skips taps: the filter row is not walked contiguously)
            c0 += BK;
            if (c0 >= p.Cin) {
                c0 = 0;
                if constexpr (MODE == 2) { kw += c_step; if (kw >= p.KW) { kw = c_kw0; kh += c_step; } }
                else if (++kw == p.KW) { kw = 0; ++kh; }
                tap_dirty = true;
            }
        } else {
            const int tap = kt * 8 + c4;  // Cin == 4: one tap per float4
            const int th = (int)p.fd_kw.div((unsigned)tap), tw = tap - th * p.KW;
            const bool tap_ok = tap < p.KH * p.KW;
#pragma unroll
            for (int i = 0; i < AR; ++i) {
                const int ih = a_ih0[i] + th, iw = a_iw0[i] + tw;
                const bool ok = tap_ok && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
                const unsigned off = (unsigned)((a_pix[i] + th * p.W + tw) * 16);
                sink_a(i, ok ? off : OOB, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < BR; ++i) sink_b(i, wrow[i], MODE == 2 ? w_koff : kt * BK * 4);
    };
    constexpr int NSET = (NS == 3 && !DL) ? 2 : 1;
    f32x4 rA[NSET][DL ? 1 : AR], rB[NSET][DL ? 1 : BR];
    auto load_tile = [&](int kt, auto set_c) __attribute__((always_inline)) {               // register staging
        constexpr int SET = decltype(set_c)::value;
        gather_tile(kt, [&](int i, unsigned off, int soff) { rA[SET][DL ? 0 : i] = buf_ld16_s(rs_in, off, soff); },
                    [&](int i, unsigned off, int soff) { rB[SET][DL ? 0 : i] = buf_ld16_s(rs_w, off, soff); });
    };
    // split mode: 4 floats -> SPL x 4 bf16 (round to nearest even; the residual of each step is exact in fp32)
    auto split_store = [&](unsigned short* row, const f32x4 v, int plane_stride) {
        typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
        typedef float f32x2_t __attribute__((ext_vector_type(2)));
        f32x4 r = v;
#pragma unroll
        for (int pl = 0; pl < (SPL ? SPL : 1); ++pl) {
            const unsigned p01 = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{r[0], r[1]}, bf16x2_t));
            const unsigned p23 = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{r[2], r[3]}, bf16x2_t));
            *reinterpret_cast<uint2*>(row + pl * plane_stride) = uint2{p01, p23};
            if (pl + 1 < SPL) {
                r[0] -= __uint_as_float(p01 << 16);
                r[1] -= __uint_as_float(p01 & 0xFFFF0000u);
                r[2] -= __uint_as_float(p23 << 16);
                r[3] -= __uint_as_float(p23 & 0xFFFF0000u);
            }
        }
    };
    auto store_tile = [&](int buf, auto set_c) __attribute__((always_inline)) {
        constexpr int SET = decltype(set_c)::value;
        float* a = As + buf * BM * RP;
        float* b = Bs + buf * BN * RP;
        if constexpr (SPL > 0) {
#pragma unroll
            for (int i = 0; i < AR; ++i)
                split_store(reinterpret_cast<unsigned short*>(a) + (rbase + 32 * i) * BP + c4 * 4, rA[SET][i], BM * BP);
#pragma unroll
            for (int i = 0; i < BR; ++i)
                split_store(reinterpret_cast<unsigned short*>(b) + (rbase + 32 * i) * BP + c4 * 4, rB[SET][i], BN * BP);
            return;
        }
#pragma unroll
        for (int i = 0; i < AR; ++i)
            *reinterpret_cast<f32x4*>(a + (rbase + 32 * i) * RP + c4 * 4) = rA[SET][DL ? 0 : i];
#pragma unroll
        for (int i = 0; i < BR; ++i)
            *reinterpret_cast<f32x4*>(b + (rbase + 32 * i) * RP + c4 * 4) = rB[SET][DL ? 0 : i];
    };
    auto dma_tile = [&](int kt, int buf) __attribute__((always_inline)) {                   // global -> LDS, asynchronous (tracked by vmcnt)
        typedef __attribute__((address_space(3))) void* lds_ptr;
        float* a = As + (buf * BM + 8 * wave) * RP;          // wave-uniform; lane l lands at +16*l bytes = row l/8, slot l%8
        float* b = Bs + (buf * BN + 8 * wave) * RP;
        gather_tile(kt, [&](int i, unsigned off, int soff) { __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (lds_ptr)(a + 32 * i * RP), 16, (int)off, soff, 0, 0); },
                    [&](int i, unsigned off, int soff) { __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_ptr)(b + 32 * i * RP), 16, (int)off, soff, 0, 0); });
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // A wave with ONE 32x32 accumulator tile (the 64x64 workgroup tile) would issue 16 MFMAs per K tile that all depend on each
    // other through the accumulator: measured ~88 cycles per dependent v_mfma_f32_32x32x2 (64 of work + a forwarding bubble),
    // i.e. 1400 instead of 1024 cycles per K tile when the wave has its SIMD to itself (the bs=1 regime).  Two accumulators that
    // take alternate K steps make consecutive MFMAs independent; they are summed once after the K loop.
    constexpr bool DUAL = TM * TN == 1 && SPL == 0;
    f32x16 acc_odd;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc_odd[r] = 0.f;

    const int frag_row = lane & 31, khalf = lane >> 5;
    const int a_frag_off = (wm * (BM / 2) + frag_row) * RP;
    const int b_frag_off = (wn * (BN / 2) + frag_row) * RP;
    int goff[4];                  // float offset of K group g's float4 (k = 8g + 4*khalf ..) inside a row
#pragma unroll
    for (int g = 0; g < 4; ++g) goff[g] = DL ? (((2 * g + khalf) ^ ((frag_row >> 1) & 7)) * 4) : (g * 8 + khalf * 4);

    auto compute = [&](int buf) __attribute__((always_inline)) {
        if constexpr (SPL > 0) {
            // bf16 32x32x16: lane l supplies A[i = l & 31][k = 8 * (l >> 5) .. + 7] of a 16-deep block: one ds_read_b128 per plane
            typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
            // LDS image (plane-major): [plane][row][BP] bf16, row pitch 80 B
            const unsigned short* a = reinterpret_cast<const unsigned short*>(As + buf * BM * RP) + (wm * (BM / 2) + frag_row) * BP + khalf * 8;
            const unsigned short* b = reinterpret_cast<const unsigned short*>(Bs + buf * BN * RP) + (wn * (BN / 2) + frag_row) * BP + khalf * 8;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                bf16x8_t fa[TM][SPL], fb[TN][SPL];
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int pl = 0; pl < SPL; ++pl)
                        fa[i][pl] = *reinterpret_cast<const bf16x8_t*>(a + i * 32 * BP + pl * BM * BP + kb * 16);
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int pl = 0; pl < SPL; ++pl)
                        fb[j][pl] = *reinterpret_cast<const bf16x8_t*>(b + j * 32 * BP + pl * BN * BP + kb * 16);
                // cross terms by rising significance (pa + pb = SPL-1 ... 0): the small ones enter the accumulator first
#pragma unroll
                for (int sig = SPL - 1; sig >= 0; --sig)
#pragma unroll
                    for (int pa = 0; pa <= sig; ++pa)
#pragma unroll
                        for (int i = 0; i < TM; ++i)
#pragma unroll
                            for (int j = 0; j < TN; ++j)
                                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][pa], fb[j][sig - pa], acc[i][j], 0, 0, 0);
            }
            return;
        }
        const float* a = As + buf * BM * RP + a_frag_off;
        const float* b = Bs + buf * BN * RP + b_frag_off;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 fa[TM], fb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const f32x4*>(a + i * 32 * RP + goff[g]);
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const f32x4*>(b + j * 32 * RP + goff[g]);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                if constexpr (DUAL) {
                    if (s & 1) acc_odd = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[0][s], fb[0][s], acc_odd, 0, 0, 0);
                    else acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[0][s], fb[0][s], acc[0][0], 0, 0, 0);
                } else {
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][s], fb[j][s], acc[i][j], 0, 0, 0);
                }
            }
        }
    };
    // Epilogue operands of the plain path (BN scale/shift, residual rows) are requested BEFORE the K loop: at bs=1 a launch
    // is a ~25 k-cycle latency chain and these loads were a serial ~1.5 k-cycle round trip after the last MFMA.
    constexpr int EC4 = BN / 4, ERPP = 256 / EC4, ENR = BM / ERPP;
    constexpr bool PRE_RES = BM * BN <= 64 * 128;               // the 128x128 tile has no registers to spare
    const bool direct = p.vec && nks == 1;
    f32x4 pre_sc = {1.f, 1.f, 1.f, 1.f}, pre_sh = {0.f, 0.f, 0.f, 0.f}, pre_res[PRE_RES ? ENR : 1];
    if (direct) {
        const int ecol4 = tid % EC4, erow0 = tid / EC4, en = n0 + ecol4 * 4;
        if (en < p.Cout) {
            if (p.scale) pre_sc = *reinterpret_cast<const f32x4*>(p.scale + en);
            if (p.shift) pre_sh = *reinterpret_cast<const f32x4*>(p.shift + en);
        }
        if (PRE_RES) {
            const __amdgpu_buffer_rsrc_t rs_res =
                __builtin_amdgcn_make_buffer_rsrc((void*)p.residual, 0, p.residual ? (unsigned)((size_t)p.M_pix * p.Cout * 4) : 0u, 0x00020000);
#pragma unroll
            for (int k = 0; k < ENR; ++k) {
                const int m = out_row(erow0 + k * ERPP);
                pre_res[k] = buf_ld16(rs_res, (m >= 0 && en < p.Cout) ? (unsigned)(((size_t)m * p.Cout + en) * 4) : OOB);
            }
        }
    }
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, (NS == 3 && !DL) ? 1 : 0>;
    if constexpr (DL && PF) {
        static_assert(NS >= 3, "fragment prefetch needs a ring of 3+");
        constexpr int D = NS - 1;
        static_assert((AR + BR) * (D - 2) < 16, "vmcnt field");
        constexpr int WAIT = 0x070 | ((AR + BR) * (D - 2));   // tiles <= t+1 landed at the barrier that ends tile t-1; lgkmcnt(0): see RING_WAIT
        const int nt = kt_end - kt_beg;
        auto read_frag = [&](int buf, int g, f32x4 (&fa)[TM], f32x4 (&fb)[TN]) __attribute__((always_inline)) {
            const float* a = As + buf * BM * RP + a_frag_off + goff[g];
            const float* b = Bs + buf * BN * RP + b_frag_off + goff[g];
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const f32x4*>(a + i * 32 * RP);
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const f32x4*>(b + j * 32 * RP);
        };
        auto mfma_group = [&](const f32x4 (&fa)[TM], const f32x4 (&fb)[TN]) __attribute__((always_inline)) {
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                if constexpr (DUAL) {
                    if (s4 & 1) acc_odd = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[0][s4], fb[0][s4], acc_odd, 0, 0, 0);
                    else acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[0][s4], fb[0][s4], acc[0][0], 0, 0, 0);
                } else {
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][s4], fb[j][s4], acc[i][j], 0, 0, 0);
                }
            }
        };
#pragma unroll
        for (int d = 0; d < D; ++d) dma_tile(kt_beg + d, d);
        __builtin_amdgcn_s_waitcnt(WAIT);
        __builtin_amdgcn_s_barrier();
        YM_STAMP(1);
        f32x4 fa0[TM], fb0[TN], fa1[TM], fb1[TN];
        read_frag(0, 0, fa0, fb0);
        // The loop is unrolled by the ring depth: tile t lives in stage t % NS, a literal at every call site below, so that every LDS
        // address (fragment reads, DMA destination) is a register + immediate and the loop body has no address VALU (conv_common.h).
        auto tile = [&](auto S, int t) __attribute__((always_inline)) {
            constexpr int buf = decltype(S)::value, buf1 = (buf + 1) % NS, nb = (buf + D) % NS;
            dma_tile(kt_beg + t + D, nb);
            read_frag(buf, 1, fa1, fb1);
            __builtin_amdgcn_sched_barrier(0);
            mfma_group(fa0, fb0);
            __builtin_amdgcn_sched_barrier(0);
            read_frag(buf, 2, fa0, fb0);
            __builtin_amdgcn_sched_barrier(0);
            mfma_group(fa1, fb1);
            __builtin_amdgcn_sched_barrier(0);
            read_frag(buf, 3, fa1, fb1);
            __builtin_amdgcn_sched_barrier(0);
            mfma_group(fa0, fb0);
            __builtin_amdgcn_sched_barrier(0);
            read_frag(buf1, 0, fa0, fb0);                  // first group of tile t+1 (landed one barrier ago)
            __builtin_amdgcn_sched_barrier(0);
            mfma_group(fa1, fb1);
            __builtin_amdgcn_s_waitcnt(WAIT);
            __builtin_amdgcn_s_barrier();
        };
        int t = 0;
        for (; t + NS <= nt; t += NS) static_for<0, NS>([&](auto S) __attribute__((always_inline)) { tile(S, t + decltype(S)::value); });
        static_for<0, NS - 1>([&](auto S) __attribute__((always_inline)) { if (t + decltype(S)::value < nt) tile(S, t + decltype(S)::value); });
        __builtin_amdgcn_s_waitcnt(0xF70);
        __builtin_amdgcn_s_barrier();
    } else if constexpr (DL) {
        constexpr int D = NS - 1;                          // prefetch distance in K tiles
        // s_waitcnt vmcnt((AR+BR)*(D-1)) lgkmcnt(0).  vmcnt: everything but the newest D-1 tiles has landed.  lgkmcnt(0) (RING_WAIT):
        // the DMA issued right after this barrier re-stages the buffer that was read during THIS iteration, and hipcc sinks the
        // last K group's MFMAs (with their lgkmcnt wait) below the raw s_barrier — so a wave could sit at the barrier with
        // ds_reads of that buffer still queued while a faster wave's DMA overwrote it (WAR; cdna guide "restage a buffer 1 phase
        // after its last ds_read only when an lgkmcnt before the barrier retired those reads").  Seen with the ring of 2 as one
        // wrong 16-byte operand chunk (32 wrong outputs) in ~1 % of the launches of a 2610-tile conv.
        static_assert((AR + BR) * (D - 1) < 16, "vmcnt field");
        constexpr int WAIT = 0x070 | ((AR + BR) * (D - 1));
        const int nt = kt_end - kt_beg;
#pragma unroll
        for (int d = 0; d < D; ++d) dma_tile(kt_beg + d, d);
        __builtin_amdgcn_s_waitcnt(WAIT);
        __builtin_amdgcn_s_barrier();
        YM_STAMP(1);
        // unrolled by the ring depth: tile t lives in stage t % NS, a literal at every call site (no address VALU in the loop body)
        auto tile = [&](auto S, int t) __attribute__((always_inline)) {
            constexpr int buf = decltype(S)::value, nb = (buf + D) % NS;     // nb: the buffer tile t-1 was read from
            dma_tile(kt_beg + t + D, nb);
            compute(buf);
            __builtin_amdgcn_s_waitcnt(WAIT);              // tile t+1 of THIS wave has landed ...
            __builtin_amdgcn_s_barrier();                  // ... and of every wave; everyone is done reading tile t
        };
        int t = 0;
        for (; t + NS <= nt; t += NS) static_for<0, NS>([&](auto S) __attribute__((always_inline)) { tile(S, t + decltype(S)::value); });
        static_for<0, NS - 1>([&](auto S) __attribute__((always_inline)) { if (t + decltype(S)::value < nt) tile(S, t + decltype(S)::value); });
        __builtin_amdgcn_s_waitcnt(0xF70);                 // the past-the-end prefetches still write LDS: drain before reuse
        __builtin_amdgcn_s_barrier();
    } else if constexpr (NS == 3) {
        // register sets alternate; tile t+2 is requested while tile t is computed and tile t+1 moves registers -> LDS
        const int nt = kt_end - kt_beg;
        load_tile(kt_beg, I0{});
        store_tile(0, I0{});
        load_tile(kt_beg + 1, I1{});
        __syncthreads();
        YM_STAMP(1);
        int buf = 0;
        // split mode: the conversion VALU of store_tile (its operands arrived an iteration ago) is interleaved with compute's
        // MFMAs — a bf16 MFMA occupies the matrix pipe for 32 cycles and ~5 other instructions issue for free in that shadow
        auto interleave = [&]() {
            if constexpr (SPL > 0) {
                constexpr int NMFMA = TM * TN * 2 * (SPL == 2 ? 3 : 6);
                constexpr int VPER = (SPL == 2 ? 12 : 24) * (AR + BR) / NMFMA + 1;     // conversion VALU per MFMA
#pragma unroll
                for (int g = 0; g < NMFMA; ++g) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // one MFMA
                    __builtin_amdgcn_sched_group_barrier(0x002, VPER, 0);   // a few VALU
                }
            }
        };
        for (int t = 0; t < nt; t += 2) {
            load_tile(kt_beg + t + 2, I0{});
            compute(buf);
            int nb = buf == LB - 1 ? 0 : buf + 1;
            store_tile(nb, I1{});            // past-the-end tiles are zeros: harmless, keeps the body branch-free
            interleave();
            __syncthreads();
            buf = nb;
            if (t + 1 >= nt) break;
            load_tile(kt_beg + t + 3, I1{});
            compute(buf);
            nb = buf == LB - 1 ? 0 : buf + 1;
            store_tile(nb, I0{});
            interleave();
            __syncthreads();
            buf = nb;
        }
    } else {
        load_tile(kt_beg, I0{});
        store_tile(0, I0{});
        __syncthreads();
        YM_STAMP(1);
        auto tile = [&](auto S, int kt) __attribute__((always_inline)) {
            constexpr int cur = decltype(S)::value;
            load_tile(kt + 1, I0{});
            compute(cur);
            store_tile(cur ^ 1, I0{});
            __syncthreads();
        };
        int kt = kt_beg;
        for (; kt + 2 <= kt_end; kt += 2) { tile(IC<0>{}, kt); tile(IC<1>{}, kt + 1); }
        if (kt < kt_end) tile(IC<0>{}, kt);
    }

    if constexpr (DUAL) acc[0][0] += acc_odd;
    YM_STAMP(2);
    // ---- epilogue: C/D layout of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5) --
    if (p.vec) {
        // Stage the BM x BN accumulator tile through LDS (the K loop ended on a barrier, so the operand
        // buffers are dead) and emit it row-major: every lane handles one float4 of a row, so residual loads
        // and output stores are full 16-byte accesses, 512 B (BN=128) / 256 B (BN=64) contiguous per row.
        constexpr int CP = BN + 4;                 // pitch in floats: keeps float4 alignment, skews banks
        constexpr int C4 = BN / 4;                 // float4 per row
        constexpr int RPP = 256 / C4;              // rows per pass
        float* C = smem;
#ifdef YM_TRACE
        // ablation of the trace build (YM_PERS_ABL=9): no staging through LDS, the stores take accumulator registers as they are
        // (WRONG values, same stores): an upper bound for what an epilogue that stores straight from the MFMA layout could save
        const bool epi_abl = p.bnb_relu == 9 && p.bn_sum == nullptr;
        if (!epi_abl) {
#else
        constexpr bool epi_abl = false;
        {
#endif
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        C[(wm * (BM / 2) + i * 32 + 4 * khalf + (r & 3) + 8 * (r >> 2)) * CP + wn * (BN / 2) + j * 32 + frag_row] =
                            acc[i][j][r];
            __syncthreads();
        }
        const int col4 = tid % C4, row0 = tid / C4;
        const int n = n0 + col4 * 4;
        double bsum[4] = {0.0, 0.0, 0.0, 0.0}, bsq[4] = {0.0, 0.0, 0.0, 0.0};   // fp64: var = E[x^2]-E[x]^2 must not cancel in fp32
        // K-slice exchange area.  Uniform split: [slice][M][Cout] (what conv_splitk_reduce reads).  Tail tiles: compact
        // [tail tile][slice][BM][BN], so the workspace does not grow with M.
        const size_t slice = (size_t)p.M * p.Cout;
        const bool in_tail = (int)blockIdx.x >= p.main_blocks;
        const unsigned sb = in_tail ? (unsigned)(BM * BN * 4) : (unsigned)(slice * 4);            // bytes between slices
        const unsigned rs = in_tail ? (unsigned)(BN * 4) : (unsigned)(p.Cout * 4);                 // bytes between rows
        const unsigned ws0 = in_tail ? (unsigned)(id - p.main_tiles) * (unsigned)nks * sb + (unsigned)(col4 * 16)
                                     : (unsigned)m0 * rs + (unsigned)(n * 4);                      // (row 0, this lane's float4), slice 0
        if (nks > 1) {
            // With arrival counters the slices are exchanged between workgroups of ONE launch (on different XCDs, each with
            // its own L2): they are written / read with agent-scope (sc1) accesses, which is all the coherence needed — no
            // L2 write-back / invalidate fences (measured: `__threadfence()` per workgroup doubled the forward time).
            const bool fused = p.counters != nullptr;
            if (n < p.Cout) {
                const unsigned base = ws0 + (unsigned)ks * sb;                        // workspace < 4 GiB (checked on the host)
#pragma unroll 4
                for (int row = row0; row < BM; row += RPP) {
                    const int m = m0 + row;
                    if (m < p.M) {
                        const f32x4 v = *reinterpret_cast<const f32x4*>(C + row * CP + col4 * 4);
                        if (fused) buf_st16_sc1(rs_ws, base + (unsigned)row * rs, v);
                        else *reinterpret_cast<f32x4*>(p.ws + (size_t)ks * slice + n + (size_t)m * p.Cout) = v;
                    }
                }
            }
            if (!fused) return;                    // the host launches conv_splitk_reduce
            // the last workgroup to arrive at this output tile owns the reduction + epilogue
            // the "I am last" flag lives in the (now dead) staging area instead of a second __shared__ object: 16 static bytes on
            // top of 2 x 80 KB of dynamic LDS would cost the split-bf16 128x128 kernel its second workgroup per CU
            int& s_last = *reinterpret_cast<int*>(smem);          // (C was consumed before the barrier below; nobody reads it again here)
            // EVERY writing wave drains its own slice stores before the barrier: __syncthreads() is a workgroup-scope fence and
            // on gfx950 that is `s_waitcnt lgkmcnt(0); s_barrier` only — it does NOT wait for outstanding global stores (vmcnt),
            // so without this line the arrival count below could overtake a slice that is still in flight and the last
            // workgroup would sum a stale slice (seen as run-to-run differences of a few elements in ~1 % of the launches:
            // tools/determinism_check.py; cdna guide §6 G16 "every writing wave drains").
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) {
                int* cnt = p.counters + tile_m * p.tiles_n + tile_n;
                const int old = __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                s_last = old == nks - 1;
                if (s_last) __hip_atomic_store(cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
            }
            __syncthreads();
            if (!s_last) { YM_STAMP(3); return; }
        }
        if (n < p.Cout) {
            f32x4 sc = pre_sc, sh = pre_sh;
            if (!direct) {
                if (p.scale) sc = *reinterpret_cast<const f32x4*>(p.scale + n);
                if (p.shift) sh = *reinterpret_cast<const f32x4*>(p.shift + n);
            }
            const int act = p.seg[0].act;
            float* outb = p.seg[0].out + n;
            const bool fwd_stats = p.bn_sum != nullptr && p.bnb_y == nullptr;
            const float* resb = (p.residual && !(direct && PRE_RES)) ? p.residual + n : nullptr;
#pragma unroll
            for (int rk = 0; rk < BM / RPP; ++rk) {
                const int row = row0 + rk * RPP;
                const int m = out_row(row);                  // row of the output tensors (MODE 2 by parity class: not m0 + row)
                if (m >= 0) {
                    f32x4 v;
                    if (nks > 1) {
                        const unsigned off = ws0 + (unsigned)row * rs;
                        v = buf_ld16_sc1(rs_ws, off);
                        int s2 = 1;
                        for (; s2 + 3 < nks; s2 += 4) {           // four slices in flight, summed in slice order
                            const f32x4 a = buf_ld16_sc1(rs_ws, off + (unsigned)s2 * sb), b = buf_ld16_sc1(rs_ws, off + (unsigned)(s2 + 1) * sb);
                            const f32x4 c = buf_ld16_sc1(rs_ws, off + (unsigned)(s2 + 2) * sb), d = buf_ld16_sc1(rs_ws, off + (unsigned)(s2 + 3) * sb);
                            v += a; v += b; v += c; v += d;
                        }
                        for (; s2 < nks; ++s2) v += buf_ld16_sc1(rs_ws, off + (unsigned)s2 * sb);
                    } else if (epi_abl) {
                        v = f32x4{acc[0][0][(4 * rk) & 15], acc[0][0][(4 * rk + 1) & 15], acc[0][0][(4 * rk + 2) & 15], acc[0][0][(4 * rk + 3) & 15]};
                    } else {
                        v = *reinterpret_cast<const f32x4*>(C + row * CP + col4 * 4);
                    }
                    v = __builtin_elementwise_fma(v, sc, sh);
                    if (direct && PRE_RES) v += pre_res[PRE_RES ? rk : 0];      // zeros when there is no residual
                    else if (resb) v += *reinterpret_cast<const f32x4*>(resb + (size_t)m * p.Cout);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = ym_apply_act(v[e], act);
                    *reinterpret_cast<f32x4*>(outb + (size_t)m * p.Cout) = v;
                    if (fwd_stats) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) { const double dv = v[e]; bsum[e] += dv; bsq[e] += dv * dv; }
                    }
                }
            }
            if (p.bnb_y) {
                // BN-BACKWARD sums of the layer whose output gradient this launch just wrote (ym_conv_desc.bnb_*; the terms of
                // k_col_reduce<1>).  A second, rolled loop that re-reads this lane's own float4s of dout (L2 hits) instead of
                // work inside the unrolled loop above: there the extra loads and constants cost every instantiation of this
                // kernel 16-24 VGPRs (64x64 tile: 124 -> 144 = one resident wave per SIMD less), whether it used them or not.
                const f32x4 b_mu = *reinterpret_cast<const f32x4*>(p.bnb_mean + n), b_is = *reinterpret_cast<const f32x4*>(p.bnb_invstd + n);
                f32x4 b_g = {1.f, 1.f, 1.f, 1.f}, b_bt = {0.f, 0.f, 0.f, 0.f};
                const bool remask = p.bnb_relu && !p.bnb_out;
                if (remask) { b_g = *reinterpret_cast<const f32x4*>(p.bnb_gamma + n); b_bt = *reinterpret_cast<const f32x4*>(p.bnb_beta + n); }
                constexpr int NR = BM / RPP, UR = 2;                     // rows of this lane; UR of them in flight at a time (4: +16 VGPRs)
#pragma unroll 1
                for (int rk0 = 0; rk0 < NR; rk0 += UR) {
                    f32x4 dv[UR], yy[UR], o[UR];
                    int mrow[UR];
#pragma unroll
                    for (int u = 0; u < UR; ++u) {
                        const int m = mrow[u] = out_row(row0 + (rk0 + u) * RPP);
                        const size_t off = (size_t)(m >= 0 ? m : 0) * p.Cout + n;       // (rows that produce nothing: a valid address, value unused)
                        dv[u] = *reinterpret_cast<const f32x4*>(p.seg[0].out + off);
                        yy[u] = *reinterpret_cast<const f32x4*>(p.bnb_y + off);
                        if (p.bnb_out) o[u] = *reinterpret_cast<const f32x4*>(p.bnb_out + off);
                    }
#pragma unroll
                    for (int u = 0; u < UR; ++u) {
                        if (mrow[u] < 0) continue;
                        if (!p.bnb_out) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) o[u][e] = remask ? bn_affine(yy[u][e], b_mu[e], b_is[e], b_g[e], b_bt[e]) : 1.f;
                        }
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const double dd = o[u][e] > 0.f ? dv[u][e] : 0.f;
                            bsum[e] += dd;
                            bsq[e] += dd * (double)((yy[u][e] - b_mu[e]) * b_is[e]);
                        }
                    }
                }
            }
        }
        if (p.bn_sum) {
            // train-mode BatchNorm statistics of THIS conv output, fused: per-workgroup column sums of the tile
            // (fp64), then one fp64 atomic per channel per workgroup.
            __syncthreads();                       // every lane is done reading the staged tile
            double* R = reinterpret_cast<double*>(smem);   // [RPP][BN][2] doubles (<= 16 KB)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                R[((row0 * BN) + col4 * 4 + e) * 2 + 0] = bsum[e];
                R[((row0 * BN) + col4 * 4 + e) * 2 + 1] = bsq[e];
            }
            __syncthreads();
            if (tid < BN && n0 + tid < p.Cout) {
                double S = 0.0, Q = 0.0;
#pragma unroll
                for (int r = 0; r < RPP; ++r) { S += R[(r * BN + tid) * 2]; Q += R[(r * BN + tid) * 2 + 1]; }
                atomicAdd(p.bn_sum + n0 + tid, S);
                atomicAdd(p.bn_sumsq + n0 + tid, Q);
            }
        }
        YM_STAMP(3);
        return;
    }
    // (image, pixel) of each of the tile's BM rows, once per row instead of once per element
    int* rowinfo = reinterpret_cast<int*>(smem);
    __syncthreads();
    if (tid < BM) {
        int b = 0, pix = 0;
        if (m0 + tid < p.M) row_to_image_pixel(p, m0 + tid, b, pix);
        rowinfo[2 * tid] = b;
        rowinfo[2 * tid + 1] = pix;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * (BN / 2) + j * 32 + frag_row;
        const bool n_ok = n < p.Cout;
        // per-lane column state: which segment, its base pointer, BN scale/shift
        float sc = 1.f, sh = 0.f;
        float* optr = nullptr;
        long long obs = 0;
        int opitch = 0, oact = 0;
        if (n_ok) {
            if (p.scale) sc = p.scale[n];
            if (p.shift) sh = p.shift[n];
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                if (s < p.nseg && n >= p.seg[s].n0 && n < p.seg[s].n1) {
                    optr = p.seg[s].out + (n - p.seg[s].n0);
                    obs = p.seg[s].bstride; opitch = p.seg[s].pitch; oact = p.seg[s].act;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int mb = m0 + wm * (BM / 2) + i * 32 + 4 * khalf;
            if (p.ksplit > 1) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = mb + (r & 3) + 8 * (r >> 2);
                    if (n_ok && m < p.M) p.ws[((size_t)ks * p.M + m) * p.Cout + n] = acc[i][j][r];
                }
            } else if (optr) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = mb + (r & 3) + 8 * (r >> 2);
                    if (m < p.M) {
                        float v = __builtin_fmaf(acc[i][j][r], sc, sh);
                        if (p.residual) v += p.residual[(size_t)m * p.Cout + n];
                        const int row = m - m0;
                        optr[(size_t)rowinfo[2 * row] * obs + (size_t)rowinfo[2 * row + 1] * opitch] = ym_apply_act(v, oact);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

__global__ __launch_bounds__(256) void conv_splitk_reduce(const ConvP p) {
    const size_t total = (size_t)p.M * p.Cout;
    if (p.vec) {
        const size_t total4 = total / 4;
        const int act = p.seg[0].act;
        for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < total4; q += (size_t)gridDim.x * 256) {
            const size_t e = q * 4;
            f32x4 v = *reinterpret_cast<const f32x4*>(p.ws + e);
            for (int s = 1; s < p.ksplit; ++s) v += *reinterpret_cast<const f32x4*>(p.ws + (size_t)s * total + e);
            const int n = (int)(e % p.Cout);
            f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
            if (p.scale) sc = *reinterpret_cast<const f32x4*>(p.scale + n);
            if (p.shift) sh = *reinterpret_cast<const f32x4*>(p.shift + n);
            v = __builtin_elementwise_fma(v, sc, sh);
            if (p.residual) v += *reinterpret_cast<const f32x4*>(p.residual + e);
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = ym_apply_act(v[k], act);
            *reinterpret_cast<f32x4*>(p.seg[0].out + e) = v;
        }
        return;
    }
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        float v = 0.f;
        for (int s = 0; s < p.ksplit; ++s) v += p.ws[(size_t)s * total + e];
        const int m = (int)(e / p.Cout), n = (int)(e - (size_t)m * p.Cout);
        epilogue_store(p, m, n, v);
    }
}

// Can the output be written by the vectorised row-major epilogue (one plain NHWC tensor, 16-byte aligned operands)?  The K-slice
// exchange of the fused split-K finish / tail split lives in that epilogue.
bool vec_epilogue(const ym_conv_desc* d) {
    const ym_conv_seg& g = d->seg[0];
    const bool aligned = (((uintptr_t)g.out | (uintptr_t)d->residual | (uintptr_t)d->scale | (uintptr_t)d->shift) & 15) == 0;
    const bool rows_in_order = d->nlevels ? g.batch_stride == 0      // pyramid: the plain output keeps the input's row order
                                          : g.batch_stride == (int64_t)d->Ho * d->Wo * d->Cout;
    return d->nseg == 1 && g.n_begin == 0 && g.n_end == d->Cout && g.pitch == d->Cout && rows_in_order && d->Cout % 4 == 0 && aligned;
}

struct Plan {
    int bm, bn, ksplit, kt_per_split, tiles_m, tiles_n, nkt, M;
    int ws = 0;                                  // 1: the weight-stationary 1x1 kernel (conv_ws.hip) with a ring of `ws_ring` stages
    int ws_ring = 0;
    int tail_tiles, tail_split, tail_ktps;     // 0 = no tail
    // stride-2 data gradient by output-pixel parity class (ConvP::cls): M is then the class-padded row count
    int cls, M_pix, cls_tile0[5], cls_rows[4], cls_w[4], cls_hw[4], cls_kh0[4], cls_kw0[4], cls_nkw[4], cls_nkt[4];
    int slots() const { return tail_tiles > 0 && tail_split > ksplit ? tail_split : ksplit; }
    size_t ws_bytes(int cout) const {          // uniform split: [ksplit][M][Cout]; tail: [tail_tiles][tail_split][bm][bn]
        const size_t u = ksplit > 1 ? (size_t)ksplit * M * cout * sizeof(float) : 0;
        const size_t t = tail_tiles > 0 ? (size_t)tail_tiles * tail_split * bm * bn * sizeof(float) : 0;
        return u > t ? u : t;
    }
    int grid() const { return (tiles_m * tiles_n - tail_tiles) * ksplit + tail_tiles * (tail_tiles > 0 ? tail_split : 0); }
};

int make_plan(const ym_conv_desc* d, Plan* pl, bool allow_cls = true) {
    YM_REQUIRE(d && d->in && d->weight, "conv: null descriptor / pointer");
    YM_REQUIRE(d->B > 0 && d->H > 0 && d->W > 0 && d->Cout > 0, "conv: bad shape");
    YM_REQUIRE(d->Cin == 4 || d->Cin % 32 == 0, "conv: Cin must be 4 (stem) or a multiple of 32, got %d", d->Cin);
    YM_REQUIRE(d->k_pad % BK == 0 && d->k_pad >= d->KH * d->KW * d->Cin, "conv: k_pad %d invalid", d->k_pad);
    if (d->transposed) {
        YM_REQUIRE(d->stride == 1 || d->stride == 2, "conv(dgrad): stride must be 1 or 2");
        YM_REQUIRE(d->Cin % 32 == 0 && d->kwaves == 0, "conv(dgrad): dy channels must be padded to a multiple of 32; workgroup kernel only");
        YM_REQUIRE(d->H == (d->Ho + 2 * d->pad - d->KH) / d->stride + 1 && d->W == (d->Wo + 2 * d->pad - d->KW) / d->stride + 1,
                   "conv(dgrad): H/W (dy) inconsistent with Ho/Wo (dx)");
    } else if (d->nlevels == 0) {
        YM_REQUIRE(d->Ho == (d->H + 2 * d->pad - d->KH) / d->stride + 1 && d->Wo == (d->W + 2 * d->pad - d->KW) / d->stride + 1,
                   "conv: Ho/Wo inconsistent with H/W/K/stride/pad");
    }
    long long M_levels = 0;
    if (d->nlevels != 0) {
        YM_REQUIRE(d->nlevels >= 1 && d->nlevels <= 5, "conv: nlevels must be 0..5");
        YM_REQUIRE(!d->transposed && d->kwaves == 0 && d->Cin % 32 == 0 && d->stride == 1 && d->KH == d->KW && (d->KH & 1) &&
                   d->pad == d->KH / 2, "conv(pyramid): stride 1, odd square filter, pad = K/2, Cin %% 32 == 0, workgroup kernel");
        for (int l = 0; l < d->nlevels; ++l) {
            YM_REQUIRE(d->level_h[l] > 0 && d->level_w[l] > 0, "conv(pyramid): bad level %d", l);
            M_levels += (long long)d->B * d->level_h[l] * d->level_w[l];
        }
    }
    YM_REQUIRE(d->nseg >= 1 && d->nseg <= 3, "conv: nseg must be 1..3");
    for (int s = 0; s < d->nseg; ++s)
        YM_REQUIRE(d->seg[s].out && d->seg[s].n_end > d->seg[s].n_begin && d->seg[s].n_end <= d->Cout,
                   "conv: bad segment %d", s);
    const long long M = d->nlevels ? M_levels : (long long)d->B * d->Ho * d->Wo;
    YM_REQUIRE(M * (long long)d->Cout < (1ll << 31) && (d->nlevels ? M : (long long)d->B * d->H * d->W) * d->Cin < (1ll << 31) * 1ll,
               "conv: tensor too large for 32-bit indexing");
    pl->M = (int)M;
    pl->M_pix = (int)M;
    pl->cls = 0;
    pl->nkt = d->k_pad / BK;
    pl->tail_tiles = 0; pl->tail_split = 0; pl->tail_ktps = 0;
    int bm = d->tile_m, bn = d->tile_n;
    if (bm == 0 || bn == 0) {
        // largest tile that still gives every CU at least ~2 workgroups
        const int cand[3][2] = {{128, 128}, {128, 64}, {64, 64}};
        bm = 64; bn = 64;
        for (int c = 0; c < 3; ++c) {
            const long long wgs = (long long)ym_cdiv(pl->M, cand[c][0]) * ym_cdiv(d->Cout, cand[c][1]);
            if (wgs >= 512) { bm = cand[c][0]; bn = cand[c][1]; break; }
        }
        if (d->Cin == 4) { bm = 128; bn = 64; }
    }
    pl->ws = 0; pl->ws_ring = 0;
    if (d->stages >= 52 && d->stages <= 54) {
        // weight-stationary 1x1 kernel (conv_ws.hip): a plain GEMM with the filter slice resident in LDS.  What it does not cover
        // (a filter with taps, a stride, BatchNorm-backward sums, too much filter for the LDS) runs as a 64x64 direct-to-LDS launch.
        const bool shape_ok = (bm == 64 && bn == 256) || (bm == 128 && bn == 128) || (bm == 256 && bn == 64);
        const int act = d->seg[0].act;
        const bool ok = shape_ok && d->KH == 1 && d->KW == 1 && d->stride == 1 && d->pad == 0 && d->Cin % 32 == 0 && d->k_pad == d->Cin &&
                        d->nlevels == 0 && d->kwaves == 0 && d->mma == 0 && d->bnb_y == nullptr && vec_epilogue(d) &&
                        (act == YM_ACT_NONE || act == YM_ACT_RELU) && (size_t)bn * d->Cin * 4 <= (64u << 10) &&
                        ym_conv_ws_lds_bytes(bm, bn, pl->nkt, d->stages - 50) <= (160u << 10) && (unsigned long long)M * d->Cout * 4ull < 0xFFFFFFF0ull &&
                        (unsigned long long)M * d->Cin * 4ull < 0xFFFFFFF0ull;      // (conv_ws.hip forms A and C offsets in 32 bits)
        if (ok) {
            pl->bm = bm; pl->bn = bn; pl->tiles_m = ym_cdiv(pl->M, bm); pl->tiles_n = ym_cdiv(d->Cout, bn);
            pl->ksplit = 1; pl->kt_per_split = pl->nkt;
            pl->ws = 1; pl->ws_ring = d->stages - 50;
            return YM_OK;
        }
        bm = 64; bn = 64;
    }
    if (d->kwaves > 0) {
        YM_REQUIRE(d->Cin != 4, "conv: the wave-private kernel does not support the stem (Cin == 4)");
        YM_REQUIRE((bm == 32 || bm == 64) && (bn == 32 || bn == 64), "conv(wave): tile must be 32/64, got %dx%d", bm, bn);
        pl->bm = bm; pl->bn = bn; pl->tiles_m = ym_cdiv(pl->M, bm); pl->tiles_n = ym_cdiv(d->Cout, bn);
        pl->ksplit = 1; pl->kt_per_split = pl->nkt;
        // tail split of the wave-private DMA-ring kernel (conv_wave.hip): 32x32 tile with four K waves, plain NHWC output, counters
        if (d->tail_tiles > 0 && d->tail_ksplit > 1 && d->stages >= 22 && d->stages <= 24 && bm == 32 && bn == 32 && d->kwaves == 4 &&
            (d->grid_wgs == 0 || d->grid_wgs == 4) && d->tile_counters && vec_epilogue(d)) {
            YM_REQUIRE(d->tail_tiles <= pl->tiles_m * pl->tiles_n, "conv(wave): tail_tiles %d > %d output tiles", d->tail_tiles, pl->tiles_m * pl->tiles_n);
            int ts = d->tail_ksplit > pl->nkt ? pl->nkt : d->tail_ksplit;
            if (ts > 8) ts = 8;                                    // (the last arriver gathers up to 8 slices at once)
            pl->tail_ktps = ym_cdiv(pl->nkt, ts);
            pl->tail_split = ym_cdiv(pl->nkt, pl->tail_ktps);
            pl->tail_tiles = pl->tail_split > 1 ? d->tail_tiles : 0;
        }
        return YM_OK;
    }
    YM_REQUIRE((bm == 128 || bm == 64) && (bn == 128 || bn == 64), "conv: tile must be 64/128");
    YM_REQUIRE(d->Cin != 4 || (bm == 128 && bn == 64), "conv: stem mode supports the 128x64 tile only");
    pl->bm = bm; pl->bn = bn;
    {
        // Stride-2 data gradient: rows ordered by output-pixel parity class, every class padded to whole M tiles, and only the
        // class's filter taps in its K range (ConvP::cls).  YM_DGRAD_CLASSES=0: the gather over all taps of rounds 1-3 (A/B).
        static int on = -1;
        if (on < 0) { const char* e = getenv("YM_DGRAD_CLASSES"); on = e ? atoi(e) : 1; }
        if (on && allow_cls && d->transposed && d->stride == 2 && d->nlevels == 0 && vec_epilogue(d)) {
            int t0 = 0, nkt_max = 0, nt_max = 0;
            for (int c = 0; c < 4; ++c) {
                const int ph = c >> 1, pw = c & 1;
                const int hc = (d->Ho - ph + 1) / 2, wc = (d->Wo - pw + 1) / 2;          // dx rows / columns of this parity
                const int kh0 = (ph + d->pad) & 1, kw0 = (pw + d->pad) & 1;
                const int nkh = kh0 < d->KH ? (d->KH - kh0 + 1) / 2 : 0, nkw = kw0 < d->KW ? (d->KW - kw0 + 1) / 2 : 0;
                pl->cls_tile0[c] = t0;
                pl->cls_rows[c] = d->B * hc * wc;
                pl->cls_w[c] = wc > 0 ? wc : 1;
                pl->cls_hw[c] = hc * wc > 0 ? hc * wc : 1;
                pl->cls_kh0[c] = kh0; pl->cls_kw0[c] = kw0; pl->cls_nkw[c] = nkw > 0 ? nkw : 1;
                pl->cls_nkt[c] = nkh * nkw * (d->Cin / BK);
                if (pl->cls_nkt[c] > nkt_max) nkt_max = pl->cls_nkt[c];
                t0 += ym_cdiv(pl->cls_rows[c], bm);
                if (ym_cdiv(pl->cls_rows[c], bm) > nt_max) nt_max = ym_cdiv(pl->cls_rows[c], bm);
            }
            pl->cls_tile0[4] = t0;
            pl->cls = 1;
            pl->M = 4 * nt_max * bm;                       // M tile t belongs to class t & 3 (its tile t >> 2): see the kernel
            pl->nkt = nkt_max > 0 ? nkt_max : 1;
        }
    }
    pl->tiles_m = ym_cdiv(pl->M, bm);
    pl->tiles_n = ym_cdiv(d->Cout, bn);
    int ks = d->ksplit;
    if (ks <= 0) {
        ks = 1;
        const int wgs = pl->tiles_m * pl->tiles_n;
        if (wgs < 256) {
            ks = ym_cdiv(512, wgs);
            const int max_ks = pl->nkt / 4 > 0 ? pl->nkt / 4 : 1;   // keep >= 4 K tiles per slice
            if (ks > max_ks) ks = max_ks;
            if (ks > 16) ks = 16;
        }
    }
    if (ks > pl->nkt) ks = pl->nkt;
    if (ks < 1) ks = 1;
    pl->kt_per_split = ym_cdiv(pl->nkt, ks);
    pl->ksplit = ym_cdiv(pl->nkt, pl->kt_per_split);
    if (d->tail_tiles > 0 && d->tail_ksplit > 1 && vec_epilogue(d)) {   // (a segmented / unaligned output ignores the tail knobs)
        YM_REQUIRE(d->tile_counters && pl->ksplit == 1 && d->Cin != 4, "conv: tail_tiles needs tile_counters, ksplit <= 1 and Cin %% 32 == 0");
        YM_REQUIRE(d->tail_tiles <= pl->tiles_m * pl->tiles_n, "conv: tail_tiles %d > %d output tiles", d->tail_tiles, pl->tiles_m * pl->tiles_n);
        int ts = d->tail_ksplit > pl->nkt ? pl->nkt : d->tail_ksplit;
        pl->tail_ktps = ym_cdiv(pl->nkt, ts);
        pl->tail_split = ym_cdiv(pl->nkt, pl->tail_ktps);
        pl->tail_tiles = pl->tail_split > 1 ? d->tail_tiles : 0;
    }
    // The class-ordered rows exist only inside the launch: K slices of such a plan must meet in the fused finish (arrival counters),
    // which maps a tile row back to its dx pixel.  `conv_splitk_reduce` reads the slabs as plain [M][Cout] rows, so without counters
    // (none given, or a workspace past the 32-bit exchange offsets) the plan falls back to the gather over all taps.
    if (pl->cls && pl->slots() > 1 && (!d->tile_counters || pl->ws_bytes(d->Cout) >= 0xFFFFFFF0ull)) return make_plan(d, pl, false);
    return YM_OK;
}

template <int BM, int BN, int MODE, int NS = 2, bool DL = false, bool PF = false, bool RG = false, int SPL = 0>
void launch(const ConvP& p, int grid, hipStream_t st) {
    size_t lds = SPL ? (size_t)2 * (BM + BN) * SPL * 80 : (size_t)NS * (BM + BN) * (DL ? 32 : PITCH) * sizeof(float);
    const size_t epi = (size_t)BM * (BN + 4) * sizeof(float);          // accumulator staging of the vector epilogue
    if (lds < epi) lds = epi;
    static YmLdsAttr attr = {};       // (a refusal leaves its message in ym_last_error; the launch below then fails and is reported)
    (void)ym_ensure_dyn_lds(attr, reinterpret_cast<const void*>(conv_igemm_f32<BM, BN, MODE, NS, DL, PF, RG, SPL>), lds, "conv_igemm_f32");
    hipLaunchKernelGGL((conv_igemm_f32<BM, BN, MODE, NS, DL, PF, RG, SPL>), dim3(grid), dim3(256), lds, st, p);
}

template <int MODE, int SPL, int NS>
void launch_split_ns(const ConvP& p, int bm, int bn, int grid, hipStream_t st) {
    if (bm == 128 && bn == 128) launch<128, 128, MODE, NS, false, false, false, SPL>(p, grid, st);
    else if (bm == 128 && bn == 64) launch<128, 64, MODE, NS, false, false, false, SPL>(p, grid, st);
    else if (bm == 64 && bn == 128) launch<64, 128, MODE, NS, false, false, false, SPL>(p, grid, st);
    else launch<64, 64, MODE, NS, false, false, false, SPL>(p, grid, st);
}
template <int MODE, int SPL>
void launch_split(const ConvP& p, int bm, int bn, int stages, int grid, hipStream_t st) {
    // stages 3: two register sets (the tile converted into LDS during an iteration was loaded a whole iteration earlier)
    if (stages == 3) launch_split_ns<MODE, SPL, 3>(p, bm, bn, grid, st);
    else launch_split_ns<MODE, SPL, 2>(p, bm, bn, grid, st);
}

}  // namespace

extern "C" size_t ym_sizeof_conv_desc(void) { return sizeof(ym_conv_desc); }

extern "C" size_t ym_conv2d_workspace_bytes(const ym_conv_desc* d) {
    Plan pl;
    if (make_plan(d, &pl) != YM_OK) return 0;
    return pl.ws_bytes(d->Cout);
}

extern "C" int ym_conv2d_tile_counters(const ym_conv_desc* d) {
    Plan pl;
    if (make_plan(d, &pl) != YM_OK) return 0;
    return pl.slots() > 1 ? pl.tiles_m * pl.tiles_n : 0;
}

extern "C" int ym_conv2d_fuses_bn_stats(const ym_conv_desc* d) {
    Plan pl;
    if (make_plan(d, &pl) != YM_OK) return 0;
    return (vec_epilogue(d) && (pl.slots() == 1 || d->tile_counters) && d->kwaves == 0) ? 1 : 0;
}

extern "C" int ym_conv2d_fwd(const ym_conv_desc* d, void* workspace, size_t workspace_bytes, ym_stream_t s) {
    Plan pl;
    int rc = make_plan(d, &pl);
    if (rc != YM_OK) return rc;
    const size_t need = pl.ws_bytes(d->Cout);
    if (need > workspace_bytes || (need && !workspace)) {
        ym_set_error("conv: workspace %zu B < %zu B needed (ksplit %d)", workspace_bytes, need, pl.ksplit);
        return YM_ENOSPC;
    }
    ConvP p;
    p.in = d->in; p.w = d->weight; p.scale = d->scale; p.shift = d->shift; p.residual = d->residual;
    p.ws = (float*)workspace;
    p.B = d->B; p.H = d->H; p.W = d->W; p.Cin = d->Cin; p.Cout = d->Cout; p.KH = d->KH; p.KW = d->KW;
    p.stride = d->stride; p.pad = d->pad; p.Ho = d->Ho; p.Wo = d->Wo; p.Kpad = d->k_pad;
    {
        const unsigned long long ib = (d->nlevels ? (unsigned long long)pl.M : (unsigned long long)d->B * d->H * d->W) * d->Cin * 4ull, wb = (unsigned long long)d->Cout * d->k_pad * 4ull;
        YM_REQUIRE(ib < 0xFFFFFFF0ull && wb < 0x7FFFFFF0ull, "conv: input must be < 4 GiB and the packed weight < 2 GiB (32-bit buffer offsets)");
        p.in_bytes = (unsigned)ib; p.w_bytes = (unsigned)wb;
    }
    p.nlev = d->nlevels;
    {
        int m_acc = 0, pix_acc = 0;
        for (int l = 0; l < 5; ++l) {
            const bool on = l < d->nlevels;
            p.lev_h[l] = on ? d->level_h[l] : 1; p.lev_w[l] = on ? d->level_w[l] : 1;
            p.lev_m[l] = m_acc; p.lev_pix[l] = pix_acc;
            if (on) { m_acc += d->B * d->level_h[l] * d->level_w[l]; pix_acc += d->level_h[l] * d->level_w[l]; }
        }
        p.lev_m[5] = m_acc; p.lev_pix[5] = pix_acc;
    }
    p.M = pl.M; p.HoWo = d->Ho * d->Wo; p.nkt = pl.nkt; p.ksplit = pl.ksplit; p.kt_per_split = pl.kt_per_split;
    p.M_pix = pl.M_pix; p.cls = pl.cls;
    for (int c = 0; c < 4; ++c) {
        const bool on = pl.cls != 0;
        p.cls_tile0[c] = on ? pl.cls_tile0[c] : 0; p.cls_rows[c] = on ? pl.cls_rows[c] : 0; p.cls_w[c] = on ? pl.cls_w[c] : 1;
        p.cls_fd_hw[c] = FastDiv::make((unsigned)(on ? pl.cls_hw[c] : 1)); p.cls_fd_w[c] = FastDiv::make((unsigned)(on ? pl.cls_w[c] : 1));
        p.cls_kh0[c] = on ? pl.cls_kh0[c] : 0; p.cls_kw0[c] = on ? pl.cls_kw0[c] : 0; p.cls_nkw[c] = on ? pl.cls_nkw[c] : 1;
        p.cls_nkt[c] = on ? pl.cls_nkt[c] : 0;
        p.cls_ktps[c] = on ? ym_cdiv(pl.cls_nkt[c], pl.ksplit) : 0;
        p.cls_tail_ktps[c] = on && pl.tail_tiles > 0 ? ym_cdiv(pl.cls_nkt[c], pl.tail_split) : 0;
    }
    p.cls_tile0[4] = pl.cls ? pl.cls_tile0[4] : 0;
    p.tiles_m = pl.tiles_m; p.tiles_n = pl.tiles_n; p.nseg = d->nseg;
    for (int i = 0; i < 3; ++i) {
        if (i < d->nseg) {
            p.seg[i].out = d->seg[i].out; p.seg[i].bstride = d->seg[i].batch_stride; p.seg[i].n0 = d->seg[i].n_begin;
            p.seg[i].n1 = d->seg[i].n_end; p.seg[i].pitch = d->seg[i].pitch; p.seg[i].act = d->seg[i].act;
        } else {
            p.seg[i] = SegDev{nullptr, 0, 0, 0, 0, 0};
        }
    }
    {
        YM_REQUIRE(pl.tail_tiles == 0 || ((uintptr_t)workspace & 15) == 0, "conv: tail split needs a 16-byte aligned workspace");
        p.vec = (vec_epilogue(d) && ((uintptr_t)workspace & 15) == 0) ? 1 : 0;
    }
    p.bn_sum = d->bn_sum; p.bn_sumsq = d->bn_sumsq;
    p.bnb_y = d->bnb_y; p.bnb_out = d->bnb_out; p.bnb_mean = d->bnb_mean; p.bnb_invstd = d->bnb_invstd;
    p.bnb_gamma = d->bnb_gamma; p.bnb_beta = d->bnb_beta; p.bnb_relu = d->bnb_relu;
    p.trace = nullptr; p.trace_epoch = nullptr; p.trace_ring = 0; p.trace_stride = 0; p.trace_rt = 0; p.trace_hw = nullptr;
#ifdef YM_TRACE
    if (const char* e = getenv("YM_TRACE_PTR")) p.trace = (long long*)strtoull(e, nullptr, 10);
    if (const char* e = getenv("YM_TRACE_EPOCH_PTR")) p.trace_epoch = (const int*)strtoull(e, nullptr, 10);
    if (const char* e = getenv("YM_TRACE_RING")) p.trace_ring = atoi(e);
    if (const char* e = getenv("YM_TRACE_GRID")) p.trace_stride = atoi(e);
    if (const char* e = getenv("YM_TRACE_REALTIME")) p.trace_rt = atoi(e);
    if (const char* e = getenv("YM_TRACE_HW_PTR")) p.trace_hw = (int*)strtoull(e, nullptr, 10);
    if (p.trace_ring <= 0) p.trace_epoch = nullptr;
    if (const char* e = getenv("YM_PERS_ABL")) { if (!d->bn_sum) p.bnb_relu = atoi(e); }      // conv_persist.hip ablations (trace build only)
#endif
    p.counters = (p.vec && pl.slots() > 1 && (d->kwaves == 0 || pl.tail_tiles > 0) && need < 0xFFFFFFF0ull) ? d->tile_counters : nullptr;
    YM_REQUIRE(!pl.cls || (p.vec && (pl.slots() == 1 || p.counters)), "conv(dgrad, stride 2): the class-ordered plan needs a 16-byte aligned workspace");
    YM_REQUIRE(pl.tail_tiles == 0 || p.counters, "conv: tail_tiles needs a plain NHWC output (vector epilogue) and a workspace < 4 GiB");
    p.main_tiles = pl.tiles_m * pl.tiles_n - pl.tail_tiles;
    p.main_blocks = p.main_tiles * pl.ksplit;
    p.tail_split = pl.tail_tiles > 0 ? pl.tail_split : 1;
    p.tail_ktps = pl.tail_tiles > 0 ? pl.tail_ktps : pl.nkt;
    p.fd_ksplit = FastDiv::make((unsigned)pl.ksplit); p.fd_tail = FastDiv::make((unsigned)p.tail_split);
    p.fd_tiles_n = FastDiv::make((unsigned)pl.tiles_n);
    ym_set_tile_order(p, 0, pl.tail_tiles == 0);          // (a tail keeps the plain order: its tiles are the LAST rows of the output)
    p.fd_howo = FastDiv::make((unsigned)(d->Ho * d->Wo));
    p.fd_wo = FastDiv::make((unsigned)d->Wo); p.fd_cin = FastDiv::make((unsigned)d->Cin); p.fd_kw = FastDiv::make((unsigned)d->KW);
    p.ws_bytes = (unsigned)(need < 0xFFFFFFF0ull ? need : 0);
    if (d->bn_sum) {
        YM_REQUIRE(d->bn_sumsq && p.vec && (pl.slots() == 1 || p.counters) && d->kwaves == 0,
                   "conv: bn_sum given but this configuration cannot fuse the statistics (ask ym_conv2d_fuses_bn_stats)");
    }
    if (d->bnb_y) {
        YM_REQUIRE(d->bn_sum && d->bnb_mean && d->bnb_invstd && (!d->bnb_relu || d->bnb_out || (d->bnb_gamma && d->bnb_beta)),
                   "conv: bnb_y needs bn_sum / bn_sumsq, bnb_mean, bnb_invstd and (with bnb_relu) bnb_out or bnb_gamma + bnb_beta");
        YM_REQUIRE(d->Cout % 4 == 0 && ((uintptr_t)d->bnb_y & 15) == 0 && ((uintptr_t)d->bnb_out & 15) == 0,
                   "conv: bnb_y / bnb_out must be 16-byte aligned [M][Cout] tensors");
    }
    hipStream_t st = (hipStream_t)s;
    if (d->kwaves > 0) {
        const bool dma = d->stages >= 22 && d->stages <= 24;
        YM_REQUIRE(!dma || (d->Cin % 32 == 0 && d->nlevels == 0 && !d->transposed && (size_t)pl.M * d->Cout * 4 < 0xFFFFFFF0ull),
                   "conv(wave, DMA ring): needs Cin %% 32 == 0, one input size, a forward convolution");
        return ym_launch_conv_wave(p, pl.bm, pl.bn, d->kwaves, d->stages, d->grid_wgs, st);
    }
    if (pl.ws) return ym_launch_conv_ws(p, pl.bm, pl.bn, pl.ws_ring, d->grid_wgs, st);
    const int grid = pl.grid();
    p.total_items = grid;
    int stages = d->stages;
    if (stages >= 52 && stages <= 54) stages = 22;       // (a weight-stationary request the kernel does not cover: see make_plan)
    if (stages >= 42 && stages <= 48) {
        // persistent direct-to-LDS kernel (conv_persist.hip); what it does not cover runs on the non-persistent ring of the same depth
        const int ns = stages - 40;
        const int act = d->seg[0].act;
        const bool ok = p.vec && pl.bm == 64 && pl.bn == 64 && d->Cin % 32 == 0 && d->nlevels == 0 && d->mma == 0 && !pl.cls &&
                        (act == YM_ACT_NONE || act == YM_ACT_RELU) && (pl.slots() == 1 || p.counters) && d->bn_sum == nullptr &&
                        (ns == 2 || ns == 3 || ns == 4 || ns == 6 || ns == 8);
        if (ok) {
            static int cus = 0;
            if (cus == 0) {
                int dev = 0, n = 0;
                if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
                cus = n;
            }
            static int defer = -1;                            // YM_PERS_DEFER=0: the synchronous epilogue (A/B experiments)
            if (defer < 0) { const char* e = getenv("YM_PERS_DEFER"); defer = e ? atoi(e) : 1; }
            int per_cu = (int)((160u << 10) / ym_conv_pers_lds_bytes(64, 64, ns, defer != 0));
            if (per_cu > 4) per_cu = 4;                       // 128 VGPRs: four waves per SIMD
            if (per_cu < 1) per_cu = 1;
            int g = d->grid_wgs > 0 ? d->grid_wgs : cus * per_cu;
            if (g > grid) g = grid;
            if (g >= 8 && g < grid) g &= ~7;                  // a workgroup's items then all lie in its own XCD's chunk of the tile space
            return ym_launch_conv_pers(p, 64, 64, d->transposed ? 2 : 0, ns, defer != 0, g, st);
        }
        stages = ns >= 4 && pl.bm == 64 && pl.bn == 64 && !d->transposed ? 24 : (ns == 2 ? 22 : 23);
    }
    // stages: 0/2 register-staged double buffer, 3 register-staged ring of 3 (64-wide tiles), 22/23/24 direct-to-LDS ring of 2/3/4,
    // 33/34 direct-to-LDS ring of 3/4 with software-pipelined fragments (64x64 forward tile)
#define YM_TILE_CASE(BM_, BN_, MODE_, HAS3_)                                                          \
    do {                                                                                              \
        if (stages == 23) launch<BM_, BN_, MODE_, 3, true>(p, grid, st);                           \
        else if (stages == 22) launch<BM_, BN_, MODE_, 2, true>(p, grid, st);                      \
        else if (stages == 3 && HAS3_) launch<BM_, BN_, MODE_, HAS3_ ? 3 : 2>(p, grid, st);        \
        else launch<BM_, BN_, MODE_, 2>(p, grid, st);                                                 \
    } while (0)
    if (d->mma != 0) {                                  // split-bf16 products (see SPL above); tensors stay fp32
        YM_REQUIRE(d->mma == 3 || d->mma == 6, "conv: mma must be 0 (f32 MFMA), 3 (bf16x3) or 6 (bf16x6), got %d", d->mma);
        YM_REQUIRE(d->nlevels == 0 && d->Cin % 32 == 0, "conv: split-bf16 mode needs Cin %% 32 == 0 and a single-size input");
        if (d->transposed) { if (d->mma == 3) launch_split<2, 2>(p, pl.bm, pl.bn, stages, grid, st); else launch_split<2, 3>(p, pl.bm, pl.bn, stages, grid, st); }
        else { if (d->mma == 3) launch_split<0, 2>(p, pl.bm, pl.bn, stages, grid, st); else launch_split<0, 3>(p, pl.bm, pl.bn, stages, grid, st); }
    } else if (d->nlevels > 0) {                        // pyramid input: register-staged double buffer
        if (pl.bm == 128 && pl.bn == 128) launch<128, 128, 0, 2, false, false, true>(p, grid, st);
        else if (pl.bm == 128 && pl.bn == 64) launch<128, 64, 0, 2, false, false, true>(p, grid, st);
        else if (pl.bm == 64 && pl.bn == 128) launch<64, 128, 0, 2, false, false, true>(p, grid, st);
        else launch<64, 64, 0, 2, false, false, true>(p, grid, st);
    } else if (d->transposed) {
        if (pl.bm == 128 && pl.bn == 128) YM_TILE_CASE(128, 128, 2, false);
        else if (pl.bm == 128 && pl.bn == 64) YM_TILE_CASE(128, 64, 2, false);
        else if (pl.bm == 64 && pl.bn == 128) YM_TILE_CASE(64, 128, 2, false);
        else YM_TILE_CASE(64, 64, 2, true);
    } else if (d->Cin == 4) launch<128, 64, 1>(p, grid, st);
    else if (pl.bm == 128 && pl.bn == 128) YM_TILE_CASE(128, 128, 0, false);
    else if (pl.bm == 128 && pl.bn == 64) YM_TILE_CASE(128, 64, 0, true);
    else if (pl.bm == 64 && pl.bn == 128) YM_TILE_CASE(64, 128, 0, true);
    else if (stages == 24) launch<64, 64, 0, 4, true>(p, grid, st);
    else if (stages == 33) launch<64, 64, 0, 3, true, true>(p, grid, st);
    else if (stages == 34) launch<64, 64, 0, 4, true, true>(p, grid, st);
    else YM_TILE_CASE(64, 64, 0, true);
#undef YM_TILE_CASE
    rc = ym_check_launch("conv_igemm_f32");
    if (rc != YM_OK) return rc;
    if (pl.ksplit > 1 && !p.counters) {
        const size_t total = (size_t)pl.M * d->Cout;
        int rgrid = (int)((total / (p.vec ? 4 : 1) + 255) / 256);
        if (rgrid > 2048) rgrid = 2048;
        hipLaunchKernelGGL(conv_splitk_reduce, dim3(rgrid), dim3(256), 0, st, p);
        rc = ym_check_launch("conv_splitk_reduce");
    }
    return rc;
}
