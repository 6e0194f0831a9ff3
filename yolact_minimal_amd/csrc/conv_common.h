// Shared between the two convolution kernels (LDS-tiled workgroup kernel, wave-private kernel).
#pragma once
#include <cstdlib>
#include "ym_common.h"

namespace ymk {

constexpr int BK = 32;   // K step in floats (one 128-byte row of either operand)

// Division by a launch-invariant positive integer: q = (mulhi(x, mg) + x) >> sh (x < 2^31).  Integer division is ~40 VALU
// instructions on gfx950; the conv prologue needs ~8 of them before its first load can issue.
struct FastDiv {
    unsigned mg, sh, d;
    __host__ static FastDiv make(unsigned d) {
        if (d == 0) d = 1;
        unsigned s = 0;
        while ((1ull << s) < d) ++s;
        return FastDiv{(unsigned)(((1ull << 32) * ((1ull << s) - d)) / d + 1), s, d};
    }
    __device__ __forceinline__ unsigned div(unsigned x) const { return (__umulhi(x, mg) + x) >> sh; }
    __device__ __forceinline__ void divmod(unsigned x, unsigned& q, unsigned& r) const { q = div(x); r = x - q * d; }
};

// Branch-free operand fetch: raw buffer loads return 0 for offsets beyond the descriptor's range, so padding taps,
// rows past M / Cout and the K tail need no control flow (and hipcc can keep COUNTED vmcnt waits across the K loop —
// with `if (ok) load` it drained vmcnt(0) before every LDS write, defeating any prefetch depth > 1).
typedef int i32x4_t __attribute__((ext_vector_type(4)));
constexpr unsigned OOB = 0xFFFFFFF0u;
__device__ __forceinline__ f32x4 buf_ld16(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 0));
}
// The same with the block-uniform part of the offset in the instruction's SGPR operand (`soffset`).  The range check compares
// lane offset + SGPR offset with the record count WITHOUT wrapping at 32 bits (measured: conv_wgrad_dma), so a lane whose
// `byte_off` is the out-of-range marker still reads 0 whatever the SGPR offset is.  Every VALU instruction of a K loop takes an issue
// slot from the MFMAs (tools/micro/mfma_lds.hip: 48 extra VALU instructions per 16 MFMAs = -17 % of the matrix pipe; the
// persistent 64x64 kernel gained 5 % from the 14 it lost this way), so the K loops keep their per-tile offset arithmetic scalar.
__device__ __forceinline__ f32x4 buf_ld16_s(__amdgpu_buffer_rsrc_t r, unsigned byte_off, int soffset) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, soffset, 0));
}
// compile-time ring stages: a K loop written per stage (unrolled by the ring depth) has every LDS address as base + immediate
template <int I> struct IC { static constexpr int value = I; };
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) { f(IC<I>{}); static_for<I + 1, N>(f); }
}
// agent-scope (sc1) accesses: coherent across the 8 XCD L2s without cache maintenance (aux bit 4 = sc1 on gfx94x/gfx950)
constexpr int AUX_SC1 = 16;
__device__ __forceinline__ f32x4 buf_ld16_sc1(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, AUX_SC1));
}
__device__ __forceinline__ void buf_st16_sc1(__amdgpu_buffer_rsrc_t r, unsigned byte_off, f32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4_t, v), r, (int)byte_off, 0, AUX_SC1);
}

struct SegDev {
    float* out;
    long long bstride;
    int n0, n1, pitch, act;
};

struct ConvP {
    const float* in;
    const float* w;
    const float* scale;
    const float* shift;
    const float* residual;
    float* ws;  // split-K partials [ksplit][M][Cout] (only when ksplit > 1)
    double* bn_sum;    // optional fused per-channel sum / sum of squares of the OUTPUT (train-mode BatchNorm)
    double* bn_sumsq;
    const float* bnb_y;       // optional (with bn_sum): BN-backward sums of the layer whose output gradient this launch writes
    const float* bnb_out;     // (ym_conv_desc.bnb_*): bn_sum += dz, bn_sumsq += dz * xhat
    const float* bnb_mean;
    const float* bnb_invstd;
    const float* bnb_gamma;
    const float* bnb_beta;
    int bnb_relu;
    int* counters;     // optional per-output-tile arrival counters: fused split-K finish (see ym_conv_desc.tile_counters)
    int B, H, W, Cin, Cout, KH, KW, stride, pad, Ho, Wo, Kpad;
    int M, HoWo, nkt, ksplit, kt_per_split, tiles_m, tiles_n;
    int main_blocks, main_tiles, tail_split, tail_ktps;   // blocks >= main_blocks: tile main_tiles + t / tail_split, slice t % tail_split
    FastDiv fd_ksplit, fd_tail, fd_tiles_n, fd_howo, fd_wo, fd_cin, fd_kw;
    int nlev;          // pyramid input (ym_conv_desc.nlevels): per level its size, first GEMM row, first output pixel of an image
    int lev_h[5], lev_w[5], lev_m[6], lev_pix[6];
    // Stride-2 DATA GRADIENT by output-pixel parity class (conv_mfma.hip, MODE 2): dx pixel (ih, iw) only receives the filter taps with
    // kh = (ih + pad) mod 2 (mod 2), kw likewise; gathering all KH x KW taps for every pixel multiplies zeros in 3 of 4 MFMAs.  The
    // GEMM rows are therefore grouped by class (c = 2 (ih & 1) + (iw & 1); M tile t holds rows of class t & 3, every class padded to
    // the same number of whole tiles, M = the padded total) and a tile walks its class's taps only: 1 + 2 + 2 + 4 taps instead of 4 x 9 for a 3x3, 1 + 0 + 0 + 0 instead of 4 for 1x1.
    int cls;                       // 1 = class mode
    int M_pix;                     // real number of dx pixels (p.M counts the padded class rows)
    int cls_tile0[5];              // (tiles of class c if the classes were laid out one after the other; informational)
    int cls_rows[4];               // dx pixels of class c = B * cls_h[c] * cls_w[c]
    int cls_w[4];                  // pixels per image row of the class
    FastDiv cls_fd_hw[4], cls_fd_w[4];
    int cls_kh0[4], cls_kw0[4], cls_nkw[4];   // first tap (then every second one), taps per filter row
    int cls_nkt[4], cls_ktps[4], cls_tail_ktps[4];   // K tiles of the class, per K slice of a main / tail tile
    // Tile order.  The XCD remap (ym_xcd_remap) hands every XCD a contiguous chunk of the tile ids; ids run n-fastest inside GROUPS of
    // `fd_gn.d` tile columns, group after group (ym_tile_decode).  One group = the plain n-fastest order: an XCD's chunk is a band of
    // tile rows across ALL columns, so each of the 8 L2s fetches its eighth of the input and the WHOLE filter (A + 8 W bytes leave
    // the fabric); groups of one column = m-fastest: the L2s partition the filter and each reads the whole input (8 A + W); g
    // groups in between give every XCD a (rows / (8 / g)) x (columns / g) block: g A + (8 / g) W.  ym_set_tile_order picks g.
    int n_groups;
    FastDiv fd_grp, fd_gn, fd_gn_last;   // tiles per group (tiles_m x group width), group width, width of the last group
    long long* trace;  // debug builds (-DYM_TRACE, tools/conv_trace.py): per-workgroup s_memtime stamps [grid][4]
    const int* trace_epoch;  // debug builds: stamps go to region (*trace_epoch % trace_ring) of `trace` ([ring][grid][4]) so that the last
    int trace_ring;          // `ring` replays of a captured launch stay readable (tools/overlap_trace.py); null / 0: one region
    int trace_stride;        // workgroup slots per region (0: gridDim.x)
    int trace_rt;            // stamp s_memrealtime (100 MHz, one counter for the chip) instead of s_memtime (shader clock, per-CU-group base)
    int* trace_hw;           // debug builds: HW_REG_HW_ID of each workgroup's first wave ([ring][grid]: CU / SE / pipe / queue ids)
    unsigned in_bytes, w_bytes, ws_bytes;   // sizes of `in` / `w` for the raw-buffer descriptors (out-of-range reads return 0)
    int total_items;   // persistent kernel (conv_persist.hip): (tile, K slice) work items = main_blocks + tail tiles * tail_split
    int nseg;
    int vec;   // 1: single segment, plain NHWC [M][Cout], Cout % 4 == 0, 16-byte aligned -> vectorised epilogue
    SegDev seg[3];
};

// id -> (tile_m, tile_n) under the group order above (block-uniform: two multiply-shift divisions)
__device__ __forceinline__ void ym_tile_decode(const ConvP& p, int id, int& tile_m, int& tile_n) {
    const unsigned g = p.fd_grp.div((unsigned)id);
    const unsigned r = (unsigned)id - g * p.fd_grp.d;
    const bool last = (int)g == p.n_groups - 1;
    const unsigned w = last ? p.fd_gn_last.d : p.fd_gn.d;
    const unsigned tm = last ? p.fd_gn_last.div(r) : p.fd_gn.div(r);
    tile_m = (int)tm;
    tile_n = (int)(g * p.fd_gn.d + (r - tm * w));
}

// Host side: choose the number of column groups from the bytes of the two operands (p.tiles_m / tiles_n / in_bytes / w_bytes set).
// `groups` > 0 forces it (1 = n-fastest, >= tiles_n = m-fastest); 0 = the traffic model: g in {1, 2, 4, 8} minimising
// g A + (8 / g) W, a change of order only for a gain of >= 5 %.  YM_TILE_GROUPS overrides (experiments; "legacy" = rounds 1-4:
// m-fastest iff the filter is the larger operand).
inline void ym_set_tile_order(ConvP& p, int groups, bool allow_model) {
    const int tn = p.tiles_n > 0 ? p.tiles_n : 1, tm = p.tiles_m > 0 ? p.tiles_m : 1;
    static const char* env = getenv("YM_TILE_GROUPS");
    if (groups <= 0) {
        const double a = (double)p.in_bytes, w = (double)p.w_bytes;
        if (env && env[0] == 'l') groups = (allow_model && w > a) ? tn : 1;
        else if (env && atoi(env) > 0) groups = allow_model ? atoi(env) : 1;
        else {
            groups = 1;
            double best = a + 8.0 * w;
            // (only where both operands are small enough to sit in the L2s next to the output -- the batch-1 regime, where every tile
            //  of the launch is in flight at once; larger launches keep the order they were tuned with: measured neutral to -0.7 %)
            const bool small = a + w <= 8.0 * 1024 * 1024;
            if (allow_model && !small && w > a) groups = tn;
            for (int g = 2; g <= 8 && allow_model && small; g *= 2) {
                if (tn < g) break;
                const double t = g * a + (8.0 / g) * w;
                if (t < 0.95 * best) { best = t; groups = g; }
            }
        }
    }
    if (groups > tn) groups = tn;
    const int gw = (tn + groups - 1) / groups;            // group width in tile columns
    p.n_groups = (tn + gw - 1) / gw;
    p.fd_gn = FastDiv::make((unsigned)gw);
    p.fd_gn_last = FastDiv::make((unsigned)(tn - (p.n_groups - 1) * gw));
    p.fd_grp = FastDiv::make((unsigned)(tm * gw));
}

// Phase stamps for tools/conv_trace.py (only in the -DYM_TRACE debug build: `make -C yolact_minimal_amd/csrc trace`).
#ifdef YM_TRACE
// stamp 0 carries the XCC id (HW_REG_XCC_ID[3:0]) in bits 60..63: every XCD has its own counter base, and in a chain of launches
// block b is NOT always on XCD b % 8.  With `trace_epoch` the stamps of replay e of a captured launch go to region e % trace_ring
// (tools/overlap_trace.py: the last few replays of every request slot stay readable); `trace_hw` receives HW_REG_HW_ID (CU,
// shader engine, compute pipe and queue of the workgroup's first wave).
// In that mode the stamps are s_memrealtime (the constant-rate counter every CU shares): s_memtime, the shader-clock counter of
// the phase stamps, has a different base on every shader engine / CU group (measured: up to 16 ms apart inside one XCD), so it
// orders nothing across CUs.
// (the epoch word is read with an agent-scope load by the stamping lane: between the kernel nodes of a captured graph the scalar
//  cache is not invalidated, and a plain `*p.trace_epoch` -- a scalar load -- returned epochs of earlier replays on some CUs)
__device__ __forceinline__ size_t ym_trace_region(const ConvP& p) {
    if (!p.trace_epoch) return 0;
    const unsigned e = (unsigned)__hip_atomic_load(p.trace_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return (size_t)(e % (unsigned)p.trace_ring) * (p.trace_stride ? (unsigned)p.trace_stride : gridDim.x);
}
#define YM_STAMP(i) do { if ((i) == 3) __builtin_amdgcn_s_waitcnt(0); if (p.trace && threadIdx.x == 0) { const size_t reg_ = ym_trace_region(p); p.trace[(reg_ + blockIdx.x) * 4 + (i)] = (long long)((p.trace_epoch || p.trace_rt) ? __builtin_amdgcn_s_memrealtime() : __builtin_amdgcn_s_memtime()) | ((i) == 0 ? (long long)(__builtin_amdgcn_s_getreg(0x1814) & 15) << 60 : 0ll); if ((i) == 0 && p.trace_hw) p.trace_hw[reg_ + blockIdx.x] = (int)__builtin_amdgcn_s_getreg(0xF804); } } while (0)
#else
#define YM_STAMP(i) do { } while (0)
#endif


// GEMM row m -> (image b, output pixel inside the image); pyramid inputs place level l's pixels after those of levels < l
__device__ __forceinline__ void row_to_image_pixel(const ConvP& p, int m, int& b, int& pix) {
    if (p.nlev > 0) {
        int base = 0, hw = p.lev_h[0] * p.lev_w[0], pix0 = 0;
#pragma unroll
        for (int q = 1; q < 5; ++q)
            if (q < p.nlev && m >= p.lev_m[q]) { base = p.lev_m[q]; hw = p.lev_h[q] * p.lev_w[q]; pix0 = p.lev_pix[q]; }
        const int local = m - base;
        b = local / hw;
        pix = pix0 + (local - b * hw);
    } else {
        b = (int)p.fd_howo.div((unsigned)m);
        pix = m - b * p.HoWo;
    }
}

__device__ __forceinline__ void epilogue_store(const ConvP& p, int m, int n, float acc) {
    float v = acc;
    v = __builtin_fmaf(v, p.scale ? p.scale[n] : 1.f, p.shift ? p.shift[n] : 0.f);
    if (p.residual) v += p.residual[(size_t)m * p.Cout + n];
    int b, pix;
    row_to_image_pixel(p, m, b, pix);
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        if (s < p.nseg && n >= p.seg[s].n0 && n < p.seg[s].n1) {
            const SegDev& g = p.seg[s];
            g.out[(size_t)b * g.bstride + (size_t)pix * g.pitch + (n - g.n0)] = ym_apply_act(v, g.act);
        }
    }
}


}  // namespace ymk

// conv_wave.hip: wave-private kernels (stages 22 / 23 / 24: the DMA-ring variant); returns YM_OK / YM_EINVAL (unsupported variant)
int ym_launch_conv_wave(const ymk::ConvP& p, int tm, int tn, int kwaves, int stages, int waves_per_block, hipStream_t st);
// conv_persist.hip: persistent direct-to-LDS kernel (ring of `ns` K tiles, `grid` workgroups walk p.total_items work items);
// mode 0 = convolution, 2 = data gradient; launches with fused BatchNorm sums are not covered.  YM_EINVAL: no such variant.
// defer: the un-split item's stores are issued under the next item's MFMAs (costs a dedicated 16 KB accumulator tile in LDS).
int ym_launch_conv_pers(const ymk::ConvP& p, int bm, int bn, int mode, int ns, bool defer, int grid, hipStream_t st);
size_t ym_conv_pers_lds_bytes(int bm, int bn, int ns, bool defer);
// conv_ws.hip: weight-stationary 1x1 / stride-1 convolution (stages 52 / 53 / 54 = ring of 2 / 3 / 4; tile 64x256, 128x128 or
// 256x64: a workgroup keeps `bn` output channels x all of K in LDS and walks M blocks of `bm` rows).  grid_wgs: workgroups (0 = as
// many as the CUs hold).  YM_EINVAL: no such variant / too much LDS.
int ym_launch_conv_ws(const ymk::ConvP& p, int bm, int bn, int nstg, int grid_wgs, hipStream_t st);
size_t ym_conv_ws_lds_bytes(int bm, int bn, int nkt, int nstg);
