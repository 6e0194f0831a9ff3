// Weight-stationary 1x1 convolution for gfx950: the short-K layers of ResNet layer1 / layer2 (and Swin's stage-1 / 2 linears) at
// batch 8 -- reference modules/resnet.py:20-40 (`conv1` / `conv3` / `downsample` of a Bottleneck), K = Cin = 64 ... 256.
//
// With a 64x64 output tile such a layer has 2 - 8 K tiles per tile: 32 - 128 MFMAs per wave between a cold prologue (address
// set-up + the first global -> LDS round trip) and an epilogue, 9 248 tiles for layer1's 256-channel convs at batch 8 --
// 39 TFLOP/s = 0.25 of the f32 MFMA peak and 1.5 TB/s, bound by neither roofline (tools/train_layer_table.py).  A 1x1 / stride-1
// convolution is a plain GEMM  out[M][N] = A[M][K] W[N][K]^T  with a SMALL W (<= 64 KB per 64 - 256 output channels), so here
//   * a workgroup loads its W slice (BN = 64 WN output channels x all of K) into LDS ONCE and keeps it: no filter traffic and no
//     filter DMA instructions per tile;
//   * it then walks M blocks of BM = 64 WM rows (j, j + gm, j + 2 gm, ...): the A rows stream through ONE ring of K-tile stages
//     (global -> LDS DMA, XOR-swizzled 16-byte chunks, D = NSTG - 1 tiles ahead) that runs on across block boundaries, so the
//     next block's first tiles are in flight during this block's last MFMAs and its epilogue;
//   * every wave owns a 64 x 64 output tile (four 32x32 accumulators): 64 MFMAs per K tile and wave against 4 fragment
//     ds_read_b128 per K group -- 4x the MFMAs per LDS byte and per barrier of the 64x64 workgroup tile;
//   * the epilogue stores straight from the accumulator layout (lane = output channel: per-lane scale / shift, 128-byte row
//     pieces per instruction, buffer stores whose row offset rides in the SGPR operand and whose range check drops rows past M) --
//     no LDS staging, no barrier;
//   * train-mode BatchNorm statistics (ym_conv_desc.bn_sum): a lane's column sums stay in REGISTERS for the workgroup's whole
//     life (fp32 over the 32 rows of a block, fp64 across blocks) and are flushed with ONE fp64 atomic per channel and wave at
//     the end -- the per-tile LDS reduction + 128 atomics per tile of conv_igemm_f32 are gone.
// The data gradient of such a layer is the same GEMM on the dgrad-packed filter ([Cin][Cout_pad] = [N][K]) and takes this kernel
// too (with the residual-gradient add); launches that carry BatchNorm-BACKWARD sums (bnb_*) stay on conv_igemm_f32.
// Same products, fp32 accumulation in K order (lane half h, step s of group g: k = 8g + 4h + s; one accumulator per tile, where the
// 64x64 kernels alternate two): results agree with the other kernels to fp32 summation order, not bit for bit.
#include "conv_common.h"

using namespace ymk;

namespace {

constexpr int waitcnt_imm(int vm, int lgkm) { return (vm & 15) | ((vm >> 4) << 14) | (7 << 4) | ((lgkm & 15) << 8); }

template <int WM, int WN, int NSTG, bool RES>
__global__ __launch_bounds__(256) void conv1x1_ws(const ConvP p, int nslices, int mblocks, int gm) {
    static_assert(WM * WN == 4, "four waves");
    constexpr int BM = 64 * WM, BN = 64 * WN, AR = BM / 32, BR = BN / 32, RP = 32, D = NSTG - 1;
    constexpr int STAGE = BM * RP;                       // floats per ring stage
    static_assert(AR * (D - 1) < 64, "vmcnt field");
    constexpr int WAIT = waitcnt_imm(AR * (D - 1), 0);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int nkt = p.nkt;
    float* Wl = smem;                                    // [nkt][BN][32], rows XOR-swizzled like the ring
    float* ring = smem + nkt * BN * RP;                  // [NSTG][BM][32]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int id = ym_xcd_remap((int)blockIdx.x, (int)gridDim.x);
    const int slice = id % nslices, j0 = id / nslices;
    const int n0 = slice * BN;
    const int N = p.Cout;
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, p.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.w_bytes, 0x00020000);
    const unsigned out_bytes = (unsigned)((size_t)p.M * N * 4);
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc((void*)p.seg[0].out, 0, out_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_res = __builtin_amdgcn_make_buffer_rsrc((void*)p.residual, 0, p.residual ? out_bytes : 0u, 0x00020000);
    typedef __attribute__((address_space(3))) void* lds_ptr;

    const int c4 = (tid & 7) ^ ((tid >> 4) & 7);         // which 16-byte chunk of a 128-byte K row this lane fetches (XOR swizzle)
    const int rbase = tid >> 3;                          // 0..31
    // ---- the W slice: once ------------------------------------------------------------------------------
    {
        unsigned wrow[BR];
#pragma unroll
        for (int i = 0; i < BR; ++i) {
            const int n = n0 + rbase + 32 * i;
            wrow[i] = n < N ? (unsigned)(n * p.Kpad + c4 * 4) * 4u : OOB;
        }
        for (int kt = 0; kt < nkt; ++kt) {
            float* dst = Wl + (kt * BN + 8 * wave) * RP;
            const int so = kt * BK * 4;
#pragma unroll
            for (int i = 0; i < BR; ++i) {
                const unsigned wo = wrow[i];
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_ptr)(dst + 32 * i * RP), 16, (int)wo, so, 0, 0);
            }
        }
    }
    // ---- the A stream: cursor (block, K tile) ---------------------------------------------------------------
    unsigned a_off[AR];
    int ld_j = j0, ld_kt = 0;
    auto setup_loader = [&](int j) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < AR; ++i) {
            const int m = j * BM + rbase + 32 * i;
            a_off[i] = (j < mblocks && m < p.M) ? (unsigned)(m * p.Cin + c4 * 4) * 4u : OOB;      // past the last block: zeros
        }
        ld_kt = 0;
    };
    auto dma_next = [&](int stage) __attribute__((always_inline)) {
        if (ld_kt == nkt) {                              // block-uniform
            ld_j += gm;
            setup_loader(ld_j);
        }
        float* a = ring + stage * STAGE + 8 * wave * RP;
        const int so = ld_kt * BK * 4;
#pragma unroll
        for (int i = 0; i < AR; ++i) {
            const unsigned ao = a_off[i];
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (lds_ptr)(a + 32 * i * RP), 16, (int)ao, so, 0, 0);
        }
        ++ld_kt;
    };

    // ---- fragments ---------------------------------------------------------------------------------------------
    const int frag_row = lane & 31, khalf = lane >> 5;
    typedef const __attribute__((address_space(3))) f32x4* lds_frag_ptr;
    lds_frag_ptr pa[4], pb[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int goff = ((2 * g + khalf) ^ ((frag_row >> 1) & 7)) * 4;
        pa[g] = (lds_frag_ptr)(ring + (wm * 64 + frag_row) * RP + goff);
        pb[g] = (lds_frag_ptr)(Wl + (wn * 64 + frag_row) * RP + goff);
    }
    f32x16 acc[2][2];
    auto read_frag = [&](int aoff4, int boff4, int g, f32x4 (&fa)[2], f32x4 (&fb)[2]) __attribute__((always_inline)) {
        fa[0] = pa[g][aoff4];
        fa[1] = pa[g][aoff4 + 32 * RP / 4];
        fb[0] = pb[g][boff4];
        fb[1] = pb[g][boff4 + 32 * RP / 4];
    };
    auto mfma_group = [&](const f32x4 (&fa)[2], const f32x4 (&fb)[2]) __attribute__((always_inline)) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][s], fb[j][s], acc[i][j], 0, 0, 0);
    };

    // ---- epilogue constants: lane = output channel --------------------------------------------------------------
    const int Nb = N * 4;                                // bytes per output row
    unsigned col_off[2];
    float sc[2], sh[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = n0 + wn * 64 + j * 32 + frag_row;
        col_off[j] = col < N ? (unsigned)(col * 4) : OOB;
        sc[j] = (p.scale && col < N) ? p.scale[col] : 1.f;
        sh[j] = (p.shift && col < N) ? p.shift[col] : 0.f;
    }
    const bool relu = p.seg[0].act == YM_ACT_RELU, stats = p.bn_sum != nullptr;
    // (RES: the residual rows are prefetched into 64 registers -- its own instantiation)
    double ds1[2] = {0.0, 0.0}, ds2[2] = {0.0, 0.0};

    // ---- the stream ------------------------------------------------------------------------------------------------
    setup_loader(j0);
#pragma unroll
    for (int d = 0; d < D; ++d) dma_next(d);
    __builtin_amdgcn_s_waitcnt(WAIT);                    // the W slice (older) and the first A tile have landed
    __builtin_amdgcn_s_barrier();
    int buf = 0, nb = D;
    for (int j = j0; j < mblocks; j += gm) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][jj][r] = 0.f;
        // first row of this lane's 2 x 16 accumulator rows, as a byte offset (the rest of the row index is block-uniform)
        const unsigned row0_off = (unsigned)(j * BM + wm * 64 + 4 * khalf) * (unsigned)Nb;
        unsigned voff[2][2];                             // lane offset of (32-row half i, channel j); an out-of-range channel stays out of range
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
                voff[i][jj] = col_off[jj] == OOB ? OOB : row0_off + (unsigned)(i * 32) * (unsigned)Nb + col_off[jj];
        float res[RES ? 2 : 1][RES ? 2 : 1][RES ? 16 : 1];
        for (int kt = 0; kt < nkt; ++kt) {
            if constexpr (RES) if (kt == nkt - 1) {      // the residual rows return under the block's last 64 MFMAs
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int so = (8 * (r >> 2) + (r & 3)) * Nb;          // 16 scalar row offsets; the 32-row step is in the lane offset
                            res[i][jj][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_res, (int)voff[i][jj], so, 0));
                        }
            }
            dma_next(nb);
            const int aoff4 = buf * (STAGE / 4), boff4 = kt * (BN * RP / 4);
            f32x4 fa0[2], fb0[2], fa1[2], fb1[2];
            read_frag(aoff4, boff4, 0, fa0, fb0);
            read_frag(aoff4, boff4, 1, fa1, fb1);
            __builtin_amdgcn_sched_barrier(0);
            mfma_group(fa0, fb0);
            __builtin_amdgcn_sched_barrier(0);
            read_frag(aoff4, boff4, 2, fa0, fb0);
            __builtin_amdgcn_sched_barrier(0);
            mfma_group(fa1, fb1);
            __builtin_amdgcn_sched_barrier(0);
            read_frag(aoff4, boff4, 3, fa1, fb1);
            __builtin_amdgcn_sched_barrier(0);
            mfma_group(fa0, fb0);
            __builtin_amdgcn_sched_barrier(0);
            mfma_group(fa1, fb1);
            // this wave's reads of stage `buf` are retired and the next tile of the stream has landed for it ... and after the
            // barrier for every wave: stage `buf` may be re-staged by the next dma_next
            __builtin_amdgcn_s_waitcnt(WAIT);
            __builtin_amdgcn_s_barrier();
            nb = buf;
            buf = buf == NSTG - 1 ? 0 : buf + 1;
        }
        if constexpr (RES) __builtin_amdgcn_s_waitcnt(waitcnt_imm(AR, 15));      // everything older than the last tile's DMA: the residual rows
        const bool full = (j + 1) * BM <= p.M;           // block-uniform: every row of the block exists
        float t1[2] = {0.f, 0.f}, t2[2] = {0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rk = i * 32 + 8 * (r >> 2) + (r & 3);
                    float v = __builtin_fmaf(acc[i][jj][r], sc[jj], sh[jj]);
                    if constexpr (RES) v += res[i][jj][r];
                    if (relu) v = v < 0.f ? 0.f : v;     // NaN stays NaN, like torch.relu
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), rs_out, (int)voff[i][jj], (8 * (r >> 2) + (r & 3)) * Nb, 0);
                    if (stats) {
                        const bool live = full || (j * BM + wm * 64 + 4 * khalf + rk) < p.M;
                        const float u = live ? v : 0.f;
                        t1[jj] += u;
                        t2[jj] = __builtin_fmaf(u, u, t2[jj]);
                    }
                }
        if (stats) {
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) { ds1[jj] += (double)t1[jj]; ds2[jj] += (double)t2[jj]; }
        }
    }
    if (stats) {
        // one fp64 atomic per channel and wave: the two lane halves hold the same channels (rows 4h + ...), combine them first
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const double a = ds1[jj] + __shfl_xor(ds1[jj], 32), b = ds2[jj] + __shfl_xor(ds2[jj], 32);
            const int col = n0 + wn * 64 + jj * 32 + frag_row;
            if (khalf == 0 && col < N) {
                atomicAdd(p.bn_sum + col, a);
                atomicAdd(p.bn_sumsq + col, b);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the run-ahead loads of "no work left" still target this workgroup's LDS
}

template <int WM, int WN, int NSTG, bool RES>
int launch_ws_r(const ConvP& p, int nslices, int mblocks, int gm, size_t lds, hipStream_t st) {
    static YmLdsAttr attr = {};
    if (int rc = ym_ensure_dyn_lds(attr, reinterpret_cast<const void*>(conv1x1_ws<WM, WN, NSTG, RES>), lds, "conv1x1_ws")) return rc;
    hipLaunchKernelGGL((conv1x1_ws<WM, WN, NSTG, RES>), dim3(nslices * gm), dim3(256), lds, st, p, nslices, mblocks, gm);
    return ym_check_launch("conv1x1_ws");
}
template <int WM, int WN, int NSTG>
int launch_ws(const ConvP& p, int nslices, int mblocks, int gm, size_t lds, hipStream_t st) {
    return p.residual ? launch_ws_r<WM, WN, NSTG, true>(p, nslices, mblocks, gm, lds, st)
                      : launch_ws_r<WM, WN, NSTG, false>(p, nslices, mblocks, gm, lds, st);
}

}  // namespace

size_t ym_conv_ws_lds_bytes(int bm, int bn, int nkt, int nstg) {
    return ((size_t)nkt * bn * 32 + (size_t)nstg * bm * 32) * sizeof(float);
}

int ym_launch_conv_ws(const ConvP& p, int bm, int bn, int nstg, int grid_wgs, hipStream_t st) {
    const int nslices = ym_cdiv(p.Cout, bn), mblocks = ym_cdiv(p.M, bm);
    const size_t lds = ym_conv_ws_lds_bytes(bm, bn, p.nkt, nstg);
    if (lds > (160u << 10)) { ym_set_error("conv(weight-stationary): %zu B of LDS (tile %dx%d, K %d, ring %d)", lds, bm, bn, p.nkt * 32, nstg); return YM_EINVAL; }
    static int cus = 0;
    if (cus == 0) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cus = n;
    }
    int per_cu = (int)((160u << 10) / lds);
    if (per_cu > 2) per_cu = 2;                          // ~190 VGPRs: two waves per SIMD
    if (per_cu < 1) per_cu = 1;
    int total = grid_wgs > 0 ? grid_wgs : cus * per_cu;
    int gm = total / nslices;
    if (gm < 1) gm = 1;
    if (gm > mblocks) gm = mblocks;
#define YM_WS(WM_, WN_)                                                                       \
    do {                                                                                      \
        if (nstg == 2) return launch_ws<WM_, WN_, 2>(p, nslices, mblocks, gm, lds, st);       \
        if (nstg == 3) return launch_ws<WM_, WN_, 3>(p, nslices, mblocks, gm, lds, st);       \
        if (nstg == 4) return launch_ws<WM_, WN_, 4>(p, nslices, mblocks, gm, lds, st);       \
    } while (0)
    if (bm == 64 && bn == 256) YM_WS(1, 4);
    else if (bm == 128 && bn == 128) YM_WS(2, 2);
    else if (bm == 256 && bn == 64) YM_WS(4, 1);
#undef YM_WS
    ym_set_error("conv(weight-stationary): tile %dx%d / ring %d not built (64x256, 128x128, 256x64; ring 2-4)", bm, bn, nstg);
    return YM_EINVAL;
}
