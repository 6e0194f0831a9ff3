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

template <int N>
struct Frag {
    f32x4 v[N][4];   // [tile row-block][k group] : lane holds k = 8g + 4h .. +3 of its row
};

// TM,TN: 32x32 MFMA tiles per wave; KW: waves sharing one output tile (K split); WPB: waves per block
template <int TM, int TN, int KW, int WPB>
__global__ __launch_bounds__(WPB * 64) void conv_wave_f32(const ConvP p) {
    constexpr int TPB = WPB / KW;            // output tiles per block
    constexpr int WM = 32 * TM, WN = 32 * TN;
    constexpr int CP = WN + 4;               // LDS pitch of a staged tile (floats)
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

    // ---- stage partial tiles in LDS, combine the KW slices in fixed order, fused epilogue ------------
    float* mine = smem + (size_t)wave * WM * CP;
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                mine[(i * 32 + 4 * khalf + (r & 3) + 8 * (r >> 2)) * CP + j * 32 + frag_row] = acc[i][j][r];
    __syncthreads();
    if (!tile_ok) return;
    const float* slab0 = smem + (size_t)(tile_local * KW) * WM * CP;
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
            for (int row = row0; row < WM; row += LANES / C4) {
                const int m = m0 + row;
                if (m >= p.M) break;
                f32x4 v = *reinterpret_cast<const f32x4*>(slab0 + row * CP + col4 * 4);
#pragma unroll
                for (int s = 1; s < KW; ++s) v += *reinterpret_cast<const f32x4*>(slab0 + (size_t)s * WM * CP + row * CP + col4 * 4);
                v = __builtin_elementwise_fma(v, sc, sh);
                if (p.residual) v += *reinterpret_cast<const f32x4*>(p.residual + (size_t)m * p.Cout + n);
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

template <int TM, int TN, int KW, int WPB>
int launch(ConvP p, hipStream_t st) {
    constexpr int TPB = WPB / KW;
    p.tiles_m = ym_cdiv(p.M, 32 * TM);
    p.tiles_n = ym_cdiv(p.Cout, 32 * TN);
    p.ksplit = 1;
    const int grid = ym_cdiv(p.tiles_m * p.tiles_n, TPB);
    const size_t lds = (size_t)WPB * 32 * TM * (32 * TN + 4) * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wave_f32<TM, TN, KW, WPB>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
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

}  // namespace

int ym_launch_conv_wave(const ConvP& p, int tm, int tn, int kwaves, hipStream_t st) {
    if (tm == 32 && tn == 32) return dispatch_kw<1, 1>(p, kwaves, st);
    if (tm == 64 && tn == 32) return dispatch_kw<2, 1>(p, kwaves, st);
    if (tm == 32 && tn == 64) return dispatch_kw<1, 2>(p, kwaves, st);
    if (tm == 64 && tn == 64) return dispatch_kw<2, 2>(p, kwaves, st);
    ym_set_error("conv(wave): tile must be 32/64 x 32/64, got %dx%d", tm, tn);
    return YM_EINVAL;
}
