// Wave-private fused convolution for gfx950 (small-M / latency-bound layers).
//
// Why a second kernel: v_mfma_f32_32x32x2_f32 consumes only two floats per lane per 64 cycles, so a wave
// can feed the matrix pipe straight from global memory (through L1/L2) with fragment-shaped 16-byte
// loads — no LDS staging, no workgroup barrier in the K loop.  That makes the WAVE the scheduling unit:
//   * a wave owns a (32*TM) x (32*TN) output tile and a 1/KW share of the K loop;
//   * the KW waves of a tile sit in the same workgroup and combine their partial sums through LDS in a
//     fixed order (deterministic), so layers like 1156x256x2304 (bs=1, layer3) fill 1000+ SIMDs from ONE
//     launch, without the global split-K workspace round trip and the extra reduce kernel;
//   * register ring of D = 3 K tiles: two tiles of loads are always in flight per wave, which is what a
//     latency-bound chain of ~10 K tiles needs (the LDS kernel exposes one L2 round trip per K tile when
//     a CU holds a single workgroup).
// Operand traffic is 2x the LDS kernel's (no sharing between the waves of a tile pair), which is why the
// large layers stay on conv_igemm_f32; tools/autotune.py picks per shape.
#include "conv_common.h"

using namespace ymk;

namespace {

// s_waitcnt immediate for gfx9 with only vmcnt counted: vmcnt[3:0] = bits 3:0, vmcnt[5:4] = bits 15:14 (expcnt / lgkmcnt: no wait)
constexpr int waitcnt_vm(int vm) { return (vm & 15) | ((vm >> 4) << 14) | (7 << 4) | (15 << 8); }

template <int N>
struct Frag {
    f32x4 v[N][4];   // [tile row-block][k group] : lane holds k = 8g + 4h .. +3 of its row
};

// Stage a wave's partial (32 TM) x (32 TN) tile in LDS, combine the KW K slices of a tile in a fixed order (deterministic), fused
// epilogue.  `slabs`: [WPB][32 TM][32 TN + 4] floats of LDS that no DMA targets any more.
template <int TM, int TN, int KW, int WPB>
__device__ __forceinline__ void wave_tile_finish(const ConvP& p, float* slabs, f32x16 (&acc)[TM][TN], int wave, int lane, bool tile_ok,
                                                 int tile_local, int kslice, int m0, int n0) {
    constexpr int WM = 32 * TM, WN = 32 * TN;
    constexpr int CP = WN + 4;               // LDS pitch of a staged tile (floats)
    const int frag_row = lane & 31, khalf = lane >> 5;
    // the residual rows of this lane are requested before the tile is staged (a wave with a tile of its own makes up to four
    // passes over it: four dependent global loads took 3.3 us of a 14.5 us launch, tools/chain_trace_rt.py)
    constexpr int C4_ = WN / 4, LANES_ = KW * 64, NPASS = (WM * C4_ + LANES_ - 1) / LANES_;
    f32x4 pre_res[NPASS];
    const bool pre = p.vec && p.residual != nullptr && tile_ok;
    if (pre) {
        const int lt_ = kslice * 64 + lane, col4_ = lt_ % C4_, row0_ = lt_ / C4_;
        const __amdgpu_buffer_rsrc_t rs_res = __builtin_amdgcn_make_buffer_rsrc((void*)p.residual, 0, (unsigned)((size_t)p.M * p.Cout * 4), 0x00020000);
#pragma unroll
        for (int k = 0; k < NPASS; ++k) {
            const int m = m0 + row0_ + k * (LANES_ / C4_), n = n0 + col4_ * 4;
            pre_res[k] = buf_ld16(rs_res, (row0_ + k * (LANES_ / C4_) < WM && m < p.M && n < p.Cout) ? (unsigned)(((size_t)m * p.Cout + n) * 4) : OOB);
        }
    }
    float* mine = slabs + (size_t)wave * WM * CP;
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                mine[(i * 32 + 4 * khalf + (r & 3) + 8 * (r >> 2)) * CP + j * 32 + frag_row] = acc[i][j][r];
    __syncthreads();
    if (!tile_ok) return;
    const float* slab0 = slabs + (size_t)(tile_local * KW) * WM * CP;
    constexpr int C4 = WN / 4;                 // float4 per tile row
    constexpr int LANES = KW * 64;             // lanes cooperating on this tile
    const int lt = kslice * 64 + lane;         // 0 .. LANES-1
    if (p.vec) {
        const int col4 = lt % C4, row0 = lt / C4;
        const int n = n0 + col4 * 4;
        if (n < p.Cout) {
            f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
            if (p.scale) sc = *reinterpret_cast<const f32x4*>(p.scale + n);
            if (p.shift) sh = *reinterpret_cast<const f32x4*>(p.shift + n);
            const int act = p.seg[0].act;
#pragma unroll
            for (int k = 0; k < NPASS; ++k) {
                const int row = row0 + k * (LANES / C4);
                const int m = m0 + row;
                if (row >= WM || m >= p.M) break;
                f32x4 v = *reinterpret_cast<const f32x4*>(slab0 + row * CP + col4 * 4);
#pragma unroll
                for (int s = 1; s < KW; ++s) v += *reinterpret_cast<const f32x4*>(slab0 + (size_t)s * WM * CP + row * CP + col4 * 4);
                v = __builtin_elementwise_fma(v, sc, sh);
                if (p.residual) v += pre_res[k];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = ym_apply_act(v[e], act);
                *reinterpret_cast<f32x4*>(p.seg[0].out + (size_t)m * p.Cout + n) = v;
            }
        }
    } else {
        for (int e = lt; e < WM * WN; e += LANES) {
            const int row = e / WN, col = e - row * WN;
            const int m = m0 + row, n = n0 + col;
            if (m < p.M && n < p.Cout) {
                float v = slab0[row * CP + col];
#pragma unroll
                for (int s = 1; s < KW; ++s) v += slab0[(size_t)s * WM * CP + row * CP + col];
                epilogue_store(p, m, n, v);
            }
        }
    }
}

// TM,TN: 32x32 MFMA tiles per wave; KW: waves sharing one output tile (K split); WPB: waves per block
template <int TM, int TN, int KW, int WPB>
__global__ __launch_bounds__(WPB * 64) void conv_wave_f32(const ConvP p) {
    constexpr int TPB = WPB / KW;            // output tiles per block
    constexpr int WM = 32 * TM, WN = 32 * TN;
    constexpr int D = 3;                     // register ring depth
    extern __shared__ __attribute__((aligned(16))) float smem[];   // [WPB][WM][CP]

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tile_local = wave / KW, kslice = wave - tile_local * KW;
    const int frag_row = lane & 31, khalf = lane >> 5;

    const int bid = ym_xcd_remap(blockIdx.x, gridDim.x);
    const int tile = bid * TPB + tile_local;
    const int ntiles = p.tiles_m * p.tiles_n;
    const bool tile_ok = tile < ntiles;
    const int tile_m = tile_ok ? tile / p.tiles_n : 0, tile_n = tile_ok ? tile - tile_m * p.tiles_n : 0;
    const int m0 = tile_m * WM, n0 = tile_n * WN;

    // ---- per-lane operand rows -----------------------------------------------------------------
    int a_pix[TM], a_ih0[TM], a_iw0[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = m0 + i * 32 + frag_row;
        if (tile_ok && m < p.M) {
            const int b = m / p.HoWo, rem = m - b * p.HoWo;
            const int oh = rem / p.Wo, ow = rem - oh * p.Wo;
            a_ih0[i] = oh * p.stride - p.pad;
            a_iw0[i] = ow * p.stride - p.pad;
            a_pix[i] = (b * p.H + a_ih0[i]) * p.W + a_iw0[i];
        } else {
            a_ih0[i] = -(1 << 20); a_iw0[i] = -(1 << 20); a_pix[i] = 0;
        }
    }
    const float* wrow[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + j * 32 + frag_row;
        wrow[j] = (tile_ok && n < p.Cout) ? p.w + (size_t)n * p.Kpad + khalf * 4 : nullptr;
    }

    // this wave's K tiles: kslice, kslice + KW, ...   (tap walker is wave-uniform)
    const int nt = tile_ok ? (p.nkt - kslice + KW - 1) / KW : 0;
    int kh, kw, c0;
    {
        const int k0 = kslice * BK;
        const int tap = k0 / p.Cin;
        c0 = k0 - tap * p.Cin;
        kh = tap / p.KW;
        kw = tap - kh * p.KW;
    }
    int kt_next = kslice;

    Frag<TM> fa[D];
    Frag<TN> fb[D];
    auto load_tile = [&](Frag<TM>& A, Frag<TN>& Bf) {
        const int tap_off = kh * p.W + kw;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int ih = a_ih0[i] + kh, iw = a_iw0[i] + kw;
            const bool ok = (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
            if (ok) {
                const float* src = p.in + (size_t)(a_pix[i] + tap_off) * p.Cin + c0 + khalf * 4;
#pragma unroll
                for (int g = 0; g < 4; ++g) A.v[i][g] = *reinterpret_cast<const f32x4*>(src + g * 8);
            } else {
#pragma unroll
                for (int g = 0; g < 4; ++g) A.v[i][g] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            if (wrow[j]) {
                const float* src = wrow[j] + (size_t)kt_next * BK;
#pragma unroll
                for (int g = 0; g < 4; ++g) Bf.v[j][g] = *reinterpret_cast<const f32x4*>(src + g * 8);
            } else {
#pragma unroll
                for (int g = 0; g < 4; ++g) Bf.v[j][g] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
        kt_next += KW;
        c0 += BK * KW;
        while (c0 >= p.Cin) {
            c0 -= p.Cin;
            if (++kw == p.KW) { kw = 0; ++kh; }
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    auto compute = [&](const Frag<TM>& A, const Frag<TN>& Bf) {
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(A.v[i][g][s], Bf.v[j][g][s], acc[i][j], 0, 0, 0);
    };

    // prologue: D-1 tiles in flight
#pragma unroll
    for (int s = 0; s < D - 1; ++s)
        if (s < nt) load_tile(fa[s], fb[s]);
    for (int t = 0; t < nt; t += D) {
#pragma unroll
        for (int u = 0; u < D; ++u) {
            if (t + u < nt) {
                if (t + u + D - 1 < nt) load_tile(fa[(u + D - 1) % D], fb[(u + D - 1) % D]);
                compute(fa[u], fb[u]);
            }
        }
    }

    wave_tile_finish<TM, TN, KW, WPB>(p, smem, acc, wave, lane, tile_ok, tile_local, kslice, m0, n0);
}

// ---- wave-private DMA rings (round 4) -----------------------------------------------------------------------------------------
// The same decomposition (a wave owns a (32 TM) x (32 TN) tile and a 1/KW share of K; the KW waves of a tile combine through LDS),
// but the operands no longer travel through registers in fragment shape (lane = row: 16 bytes of 32 different rows per load
// instruction, a quarter of every cache line it touches).  Each wave owns a ring of NS K-tile stages in LDS and fills it with
// `buffer_load_dwordx4 ... lds` (64 lanes x 16 B = 8 rows x 128 B per instruction: whole lines, no staging registers), D = NS - 1
// tiles ahead; the fragments are ds_read_b128 out of the XOR-swizzled rows exactly as in conv_igemm_f32<.., DL = true> (slot =
// chunk ^ ((row >> 1) & 7): the 16 rows of a read phase fall into 16 distinct bank groups).  A ring is private to its wave: the K
// loop has NO workgroup barrier -- `s_waitcnt vmcnt(n)` on the wave's own loads is the only synchronisation -- and a stage is
// re-filled one iteration after its last ds_read returned.
// Why (tools/chain_trace_rt.py, batch 1): a layer3 conv split 3-6 ways over workgroups spends 6-7.4 us of every launch in the
// K-slice exchange of its last arriver (publish through the fabric, arrival atomic, read everything back) behind a K loop of
// 8-16 us; with the K split INSIDE the workgroup the exchange is an LDS hand-off.
// WPB: waves per workgroup (>= KW).  Tiles whose waves share nothing (KW == 1) can be launched as one- or two-wave workgroups: the
// dispatcher then balances single tiles over the CUs instead of groups of four (1184 tiles of layer3's conv3 = 296 four-wave
// workgroups = two of them on 40 of the 256 CUs).
template <int TM, int TN, int KW, int NS, int WPB = 4>
__global__ __launch_bounds__(WPB * 64) void conv_wdma_f32(const ConvP p) {
    static_assert(WPB % KW == 0 && WPB <= 4, "the K waves of a tile share a workgroup");
    constexpr int TPB = WPB / KW;            // output tiles per block
    constexpr int WM = 32 * TM, WN = 32 * TN;
    constexpr int AR = 4 * TM, BR = 4 * TN;  // DMA instructions per K tile: 8 rows of 128 B each
    constexpr int RP = 32;                   // LDS floats per tile row (unpadded: the DMA places lane l's 16 bytes at base + 16 l)
    constexpr int STAGE = (WM + WN) * RP;    // floats per ring stage of ONE wave: [WM rows of A | WN rows of W]
    constexpr int D = NS - 1;
    static_assert(NS >= 2 && NS <= 4 && (AR + BR) * D < 64, "ring depth / vmcnt field");
    static_assert((size_t)STAGE >= (size_t)WM * (WN + 4), "the finished tile is staged in the wave's first ring stage");
    extern __shared__ __attribute__((aligned(16))) float smem[];   // [WPB][NS][STAGE]

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tile_local = wave / KW, kslice = wave - tile_local * KW;
    const int frag_row = lane & 31, khalf = lane >> 5;
    YM_STAMP(0);
    // Tail split (ym_conv_desc.tail_tiles, KW == 4 and a 32x32 tile only): 296 tiles on 256 CUs leave 40 CUs with two workgroups, and
    // the launch lasts as long as those (tools/chain_trace_rt.py: layer3's 3x3 at batch 1, median workgroup exit 14.7 us, last 22.7).
    // The LAST tail_tiles tiles are therefore computed as tail_split K slices each, by workgroups that fill the second slot of
    // every CU with a fraction of a tile; their partial tiles meet through the workspace (sc1 stores, arrival counter, the last
    // arriver sums in slice order and runs the epilogue: conv_igemm_f32's exchange) while the whole tiles are still in their K loops.
    constexpr bool TAIL = TM * TN == 1 && KW == 4 && WPB == 4;
    const bool in_tail = TAIL && (int)blockIdx.x >= p.main_blocks;
    int tile, tslice = 0, tsplit = 1;
    if (in_tail) {
        unsigned q, r;
        p.fd_tail.divmod(blockIdx.x - (unsigned)p.main_blocks, q, r);
        tile = p.main_tiles + (int)q; tslice = (int)r; tsplit = p.tail_split;
    } else {
        tile = ym_xcd_remap(blockIdx.x, TAIL ? p.main_blocks : (int)gridDim.x) * TPB + tile_local;
    }
    const bool tile_ok = tile < p.tiles_m * p.tiles_n;
    int tile_m = 0, tile_n = 0;
    if (tile_ok) ym_tile_decode(p, tile, tile_m, tile_n);
    const int m0 = tile_m * WM, n0 = tile_n * WN;
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, p.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.w_bytes, 0x00020000);

    // ---- DMA coordinates: instruction i of an operand covers its rows 8 i .. 8 i + 7, lane l fills slot l & 7 of row 8 i + (l >> 3),
    // i.e. fetches chunk (l & 7) ^ ((row >> 1) & 7) = (l & 7) ^ ((4 i + (l >> 4)) & 7) -------------------------------------------------
    const int drow = lane >> 3;
    int a_pix[AR], a_ih0[AR], a_iw0[AR];
    unsigned a_off[AR], wrow[BR];
#pragma unroll
    for (int i = 0; i < AR; ++i) {
        const int m = m0 + 8 * i + drow;
        if (tile_ok && m < p.M) {
            unsigned ub, urem, uoh, uow;
            p.fd_howo.divmod((unsigned)m, ub, urem);
            p.fd_wo.divmod(urem, uoh, uow);
            a_ih0[i] = (int)uoh * p.stride - p.pad;
            a_iw0[i] = (int)uow * p.stride - p.pad;
            a_pix[i] = ((int)ub * p.H + a_ih0[i]) * p.W + a_iw0[i];
        } else {
            a_ih0[i] = -(1 << 20); a_iw0[i] = -(1 << 20); a_pix[i] = 0;
        }
    }
#pragma unroll
    for (int j = 0; j < BR; ++j) {
        const int n = n0 + 8 * j + drow;
        const int c4 = (lane & 7) ^ ((4 * j + (lane >> 4)) & 7);
        wrow[j] = (tile_ok && n < p.Cout) ? (unsigned)((n * p.Kpad + c4 * 4) * 4) : p.w_bytes;       // past Cout: parked at the buffer's end
    }

    // this wave's K tiles: a contiguous range (the filter-tap walker is wave-uniform)
    // (a tail workgroup owns K slice `tslice` of its tile, [tslice * tail_ktps, ...), and splits THAT among its waves)
    const int k_lo = in_tail ? tslice * p.tail_ktps : 0, k_hi = in_tail ? min(p.nkt, k_lo + p.tail_ktps) : p.nkt;
    const int per = (k_hi - k_lo + KW - 1) / KW;
    const int kt_beg = k_lo + kslice * per, kt_end = tile_ok ? min(k_hi, kt_beg + per) : kt_beg;
    const int nt = max(0, kt_end - kt_beg);
    int kh, kw, c0;
    {
        unsigned tap, uc0, ukh, ukw;
        p.fd_cin.divmod((unsigned)(kt_beg * BK), tap, uc0);
        p.fd_kw.divmod(tap, ukh, ukw);
        c0 = (int)uc0; kh = (int)ukh; kw = (int)ukw;
    }
    bool tap_dirty = true;
    int ld_kt = kt_beg;
    float* ring = smem + (size_t)wave * NS * STAGE;

    // next K tile of this wave's stream -> ring stage `stage` (asynchronous, AR + BR loads per lane, always: tiles past the range
    // are fetched like any other and never consumed, so the counted vmcnt waits stay valid)
    auto dma_next = [&](int stage) __attribute__((always_inline)) {
        typedef __attribute__((address_space(3))) void* lds_ptr;
        if (tap_dirty) {                           // wave-uniform: the filter tap changed (never inside a 1x1 conv)
            tap_dirty = false;
#pragma unroll
            for (int i = 0; i < AR; ++i) {
                const int ih = a_ih0[i] + kh, iw = a_iw0[i] + kw;
                const bool ok = (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
                const int c4 = (lane & 7) ^ ((4 * i + (lane >> 4)) & 7);
                a_off[i] = (unsigned)(((a_pix[i] + kh * p.W + kw) * p.Cin + c4 * 4) * 4) | (ok ? 0u : OOB);
            }
        }
        float* a = ring + stage * STAGE;           // wave-uniform; lane l lands at + 16 l bytes of each 1 KB block
        float* b = a + WM * RP;
        // (tiles past this wave's K range are fetched like any other and never consumed: the load count per step must not change.  An
        //  SGPR offset beyond the buffer was tried to make them return at once -- 17.6 -> 21.1 us per launch in the layer3 chain: such
        //  an access is not an early-out.)
        const int soff_a = c0 * 4, soff_b = ld_kt * BK * 4;
#pragma unroll
        for (int i = 0; i < AR; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (lds_ptr)(a + 8 * i * RP), 16, (int)a_off[i], soff_a, 0, 0);
#pragma unroll
        for (int j = 0; j < BR; ++j) {
            const unsigned wo = wrow[j];           // (a local copy: hipcc drops the kernel's host stub for an array element here)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_ptr)(b + 8 * j * RP), 16, (int)wo, soff_b, 0, 0);
        }
        ++ld_kt;
        c0 += BK;
        if (c0 >= p.Cin) {
            c0 = 0;
            if (++kw == p.KW) { kw = 0; ++kh; }
            tap_dirty = true;
        }
    };

    // fragments: lane half h takes k = 8 g + 4 h + s of K group g (one ds_read_b128 per operand row feeds four MFMA steps)
    typedef const __attribute__((address_space(3))) f32x4* lds_frag_ptr;
    lds_frag_ptr pa[4], pb[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int goff = ((2 * g + khalf) ^ ((frag_row >> 1) & 7)) * 4;
        pa[g] = (lds_frag_ptr)(ring + frag_row * RP + goff);
        pb[g] = (lds_frag_ptr)(ring + WM * RP + frag_row * RP + goff);
    }
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    constexpr bool DUAL = TM * TN == 1;      // a lone accumulator would chain every MFMA to the previous one (~88 instead of 64 cycles)
    f32x16 acc_odd;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc_odd[r] = 0.f;

    auto compute = [&](auto S) __attribute__((always_inline)) {
        constexpr int ST = decltype(S)::value;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 fa[TM], fb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[i] = pa[g][(ST * STAGE + i * 32 * RP) / 4];
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[j] = pb[g][(ST * STAGE + j * 32 * RP) / 4];
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
        }
    };

    // ---- the stream: tile t lives in stage t % NS (a literal below: the loop is unrolled by the ring depth) ---------------------------
    // vmcnt((AR + BR) D): everything but the newest D tiles has landed = tile t.  lgkmcnt(0): the ds_reads of the previous tile have
    // returned before the DMA issued below re-fills their stage (they fed MFMAs that were issued, but hipcc may sink the wait).
    constexpr int WAIT = waitcnt_vm((AR + BR) * D) & ~(15 << 8);       // + lgkmcnt(0)
#pragma unroll
    for (int d = 0; d < D; ++d) dma_next(d);
#ifdef YM_TRACE
    if (p.trace) { __builtin_amdgcn_s_waitcnt(waitcnt_vm((AR + BR) * (D - 1))); YM_STAMP(1); }      // (first tile landed)
#endif
#ifdef YM_TRACE
    // ablations of the trace build (YM_PERS_ABL -> p.bnb_relu, tools/wave_ablation.py): 1 = operand stream only (DMA + its counted
    // wait, no LDS reads, no MFMAs), 2 = LDS reads + MFMAs only (whatever the ring holds), 3 = MFMAs only (register operands)
    const int abl = p.bnb_relu;
#endif
    auto tile_step = [&](auto S) __attribute__((always_inline)) {
        constexpr int ST = decltype(S)::value;
        __builtin_amdgcn_sched_barrier(0);         // the re-fill of the stage read last stays behind the MFMAs that consumed its fragments
#ifdef YM_TRACE
        if (abl >= 1 && abl <= 3) {
            if (abl == 1) {
                dma_next((ST + D) % NS);
                __builtin_amdgcn_s_waitcnt(WAIT);
            } else if (abl == 2) {
                compute(S);
            } else {
                const f32x4 ra = {1.f, 2.f, 3.f, 4.f}, rb = {.5f, .25f, .125f, 1.f};
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int s4 = 0; s4 < 4; ++s4) {
                        if constexpr (DUAL) {
                            if (s4 & 1) acc_odd = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[s4], rb[s4], acc_odd, 0, 0, 0);
                            else acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[s4], rb[s4], acc[0][0], 0, 0, 0);
                        } else {
#pragma unroll
                            for (int i = 0; i < TM; ++i)
#pragma unroll
                                for (int j = 0; j < TN; ++j)
                                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[s4], rb[s4], acc[i][j], 0, 0, 0);
                        }
                    }
            }
        } else
#endif
        {
            dma_next((ST + D) % NS);
            __builtin_amdgcn_s_waitcnt(WAIT);
            compute(S);
        }
    };
    int t = 0;
    for (; t + NS <= nt; t += NS) static_for<0, NS>([&](auto S) __attribute__((always_inline)) { tile_step(S); });
    static_for<0, NS - 1>([&](auto S) __attribute__((always_inline)) { if (t + decltype(S)::value < nt) tile_step(S); });
    if constexpr (DUAL) acc[0][0] += acc_odd;
    YM_STAMP(2);
    // the run-ahead loads still target this wave's ring: drain them before the first stage becomes the tile's staging slab
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    // slabs: one per wave, WM x (WN + 4) floats, at the start of each wave's ring (ring pitch NS * STAGE >= slab size)
    // -> wave_tile_finish expects them back to back: use a compact region at the start of the block's LDS instead (every ring is dead)
    if constexpr (TAIL) {
        if (in_tail) {
            // 32x32 tile, 256 lanes: every lane owns ONE float4 of the tile.  Combine the four K waves through LDS, publish the partial
            // tile of this K slice, arrive; the last arriver of the tile sums the slices in slice order and runs the epilogue.
            constexpr int CP = WN + 4;
            float* mine = smem + (size_t)wave * WM * CP;
#pragma unroll
            for (int r = 0; r < 16; ++r) mine[(4 * khalf + (r & 3) + 8 * (r >> 2)) * CP + frag_row] = acc[0][0][r];
            __syncthreads();
            const int lt = threadIdx.x, col4 = lt & 7, row = lt >> 3;
            const int n = n0 + col4 * 4, m = m0 + row;
            f32x4 v = *reinterpret_cast<const f32x4*>(smem + row * CP + col4 * 4);
#pragma unroll
            for (int w2 = 1; w2 < KW; ++w2) v += *reinterpret_cast<const f32x4*>(smem + (size_t)w2 * WM * CP + row * CP + col4 * 4);
            const __amdgpu_buffer_rsrc_t rs_ws = __builtin_amdgcn_make_buffer_rsrc((void*)p.ws, 0, p.ws_bytes, 0x00020000);
            const unsigned sb = WM * WN * 4;                                              // bytes between the slices of a tile
            const unsigned ws0 = (unsigned)(tile - p.main_tiles) * (unsigned)tsplit * sb + (unsigned)(row * WN + col4 * 4) * 4;
            buf_st16_sc1(rs_ws, ws0 + (unsigned)tslice * sb, v);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // EVERY writing wave drains its slice stores before the arrival
            __syncthreads();
            int* s_last = reinterpret_cast<int*>(smem);            // (the slabs were consumed before the barrier above)
            if (lt == 0) {
                int* cnt = p.counters + tile;
                const int old = __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                *s_last = old == tsplit - 1;
                if (old == tsplit - 1) __hip_atomic_store(cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
            }
            __syncthreads();
            if (*s_last && m < p.M && n < p.Cout) {
                f32x4 part[8];
#pragma unroll
                for (int s2 = 0; s2 < 8; ++s2) part[s2] = buf_ld16_sc1(rs_ws, s2 < tsplit ? ws0 + (unsigned)s2 * sb : OOB);
                f32x4 sum = part[0];
#pragma unroll
                for (int s2 = 1; s2 < 8; ++s2) if (s2 < tsplit) sum += part[s2];
                f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
                if (p.scale) sc = *reinterpret_cast<const f32x4*>(p.scale + n);
                if (p.shift) sh = *reinterpret_cast<const f32x4*>(p.shift + n);
                sum = __builtin_elementwise_fma(sum, sc, sh);
                if (p.residual) sum += *reinterpret_cast<const f32x4*>(p.residual + (size_t)m * p.Cout + n);
                const int act = p.seg[0].act;
#pragma unroll
                for (int e = 0; e < 4; ++e) sum[e] = ym_apply_act(sum[e], act);
                *reinterpret_cast<f32x4*>(p.seg[0].out + (size_t)m * p.Cout + n) = sum;
            }
            YM_STAMP(3);
            return;
        }
    }
    wave_tile_finish<TM, TN, KW, WPB>(p, smem, acc, wave, lane, tile_ok, tile_local, kslice, m0, n0);
    YM_STAMP(3);
}

template <int TM, int TN, int KW, int WPB>
int launch(ConvP p, hipStream_t st) {
    constexpr int TPB = WPB / KW;
    p.tiles_m = ym_cdiv(p.M, 32 * TM);
    p.tiles_n = ym_cdiv(p.Cout, 32 * TN);
    p.ksplit = 1;
    const int grid = ym_cdiv(p.tiles_m * p.tiles_n, TPB);
    const size_t lds = (size_t)WPB * 32 * TM * (32 * TN + 4) * sizeof(float);
    static YmLdsAttr attr = {};
    if (int rc = ym_ensure_dyn_lds(attr, reinterpret_cast<const void*>(conv_wave_f32<TM, TN, KW, WPB>), lds, "conv_wave_f32")) return rc;
    hipLaunchKernelGGL((conv_wave_f32<TM, TN, KW, WPB>), dim3(grid), dim3(WPB * 64), lds, st, p);
    return ym_check_launch("conv_wave_f32");
}

template <int TM, int TN>
int dispatch_kw(const ConvP& p, int kwaves, hipStream_t st) {
    switch (kwaves) {
        case 1: return launch<TM, TN, 1, 4>(p, st);
        case 2: return launch<TM, TN, 2, 4>(p, st);
        case 4: return launch<TM, TN, 4, 4>(p, st);
        case 8:
            if constexpr (TM * TN < 4) return launch<TM, TN, 8, 8>(p, st);   // 64x64 x 8 waves would spill (512-thread block)
            [[fallthrough]];
        default: ym_set_error("conv(wave): unsupported kwaves %d for a %dx%d wave tile", kwaves, 32 * TM, 32 * TN); return YM_EINVAL;
    }
}

template <int TM, int TN, int KW, int NS, int WPB>
int launch_dma(ConvP p, hipStream_t st) {
    constexpr int TPB = WPB / KW;
    p.tiles_m = ym_cdiv(p.M, 32 * TM);
    p.tiles_n = ym_cdiv(p.Cout, 32 * TN);
    p.fd_tiles_n = FastDiv::make((unsigned)p.tiles_n);
    ym_set_tile_order(p, 0, true);           // (the wave tiles differ from the planner's: the order is chosen for THIS tiling)
    p.ksplit = 1;
    int grid = ym_cdiv(p.tiles_m * p.tiles_n, TPB);
    if (TM * TN == 1 && KW == 4 && WPB == 4) {           // (the caller planned the tail: main_tiles / tail_split / tail_ktps / counters)
        p.main_blocks = p.main_tiles;
        grid = p.main_tiles + (p.tiles_m * p.tiles_n - p.main_tiles) * p.tail_split;
    }
    const size_t lds = (size_t)WPB * NS * (32 * TM + 32 * TN) * 32 * sizeof(float);
    static YmLdsAttr attr = {};
    if (int rc = ym_ensure_dyn_lds(attr, reinterpret_cast<const void*>(conv_wdma_f32<TM, TN, KW, NS, WPB>), lds, "conv_wdma_f32")) return rc;
    hipLaunchKernelGGL((conv_wdma_f32<TM, TN, KW, NS, WPB>), dim3(grid), dim3(WPB * 64), lds, st, p);
    return ym_check_launch("conv_wdma_f32");
}

template <int TM, int TN>
int dispatch_dma(const ConvP& p, int kwaves, int ns, int wpb, hipStream_t st) {
    if (wpb <= 0) wpb = 4;
#define YM_DMA_CASE(KW_, NS_, WPB_) if (kwaves == KW_ && ns == NS_ && wpb == WPB_) return launch_dma<TM, TN, KW_, NS_, WPB_>(p, st)
    YM_DMA_CASE(1, 2, 4); YM_DMA_CASE(1, 3, 4); YM_DMA_CASE(2, 2, 4); YM_DMA_CASE(2, 3, 4); YM_DMA_CASE(4, 2, 4); YM_DMA_CASE(4, 3, 4);
    YM_DMA_CASE(1, 2, 1); YM_DMA_CASE(1, 3, 1); YM_DMA_CASE(1, 2, 2); YM_DMA_CASE(1, 3, 2); YM_DMA_CASE(2, 2, 2); YM_DMA_CASE(2, 3, 2);
    if constexpr (TM * TN == 1) { YM_DMA_CASE(1, 4, 4); YM_DMA_CASE(2, 4, 4); YM_DMA_CASE(4, 4, 4); YM_DMA_CASE(1, 4, 1); YM_DMA_CASE(1, 4, 2); }
#undef YM_DMA_CASE
    ym_set_error("conv(wave, DMA ring): kwaves %d, ring of %d, %d waves per workgroup is not built for a %dx%d wave tile", kwaves, ns, wpb, 32 * TM, 32 * TN);
    return YM_EINVAL;
}

}  // namespace

int ym_launch_conv_wave(const ConvP& p, int tm, int tn, int kwaves, int stages, int wpb, hipStream_t st) {
    if (stages >= 22 && stages <= 24) {          // wave-private DMA rings of 2 / 3 / 4 K tiles (Cin % 32 == 0, one input size)
        if (tm == 32 && tn == 32) return dispatch_dma<1, 1>(p, kwaves, stages - 20, wpb, st);
        if (tm == 64 && tn == 32) return dispatch_dma<2, 1>(p, kwaves, stages - 20, wpb, st);
        if (tm == 32 && tn == 64) return dispatch_dma<1, 2>(p, kwaves, stages - 20, wpb, st);
        ym_set_error("conv(wave, DMA ring): tile must be 32x32, 64x32 or 32x64, got %dx%d", tm, tn);
        return YM_EINVAL;
    }
    if (tm == 32 && tn == 32) return dispatch_kw<1, 1>(p, kwaves, st);
    if (tm == 64 && tn == 32) return dispatch_kw<2, 1>(p, kwaves, st);
    if (tm == 32 && tn == 64) return dispatch_kw<1, 2>(p, kwaves, st);
    if (tm == 64 && tn == 64) return dispatch_kw<2, 2>(p, kwaves, st);
    ym_set_error("conv(wave): tile must be 32/64 x 32/64, got %dx%d", tm, tn);
    return YM_EINVAL;
}
