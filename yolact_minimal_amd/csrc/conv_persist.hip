// Persistent fused convolution for gfx950 (the 64x64-tile regime of conv_mfma.hip: ResNet layer2-4 at every batch size).
//
// Same implicit GEMM, same MFMA order and same epilogue arithmetic as conv_igemm_f32<64,64,MODE,NS,DL=true> -- bit-identical
// results -- but the launch is `grid` workgroups that WALK the (output tile, K slice) work items  w, w + grid, w + 2 grid, ...
// and the operand stream never stops at an item boundary:
//
//   * ONE ring of NS K-tile stages per workgroup, filled by `buffer_load ... lds` (global -> LDS DMA, no staging registers) D =
//     NS-1 tiles ahead of the MFMAs.  The loader has its own cursor (item, K tile): when it has issued an item's last K tile it
//     decodes the workgroup's NEXT item and keeps going, so the first D tiles of item i+1 are in flight while item i's last K
//     tiles are multiplied and while its epilogue runs.  In conv_igemm_f32 those slots fetched past-the-end garbage, and every
//     item (= workgroup there) paid address set-up + a cold L2/MALL round trip (~1.7 us) before its first MFMA and an epilogue
//     (~2-3 us) with nothing in flight: at M ~ 9 k (batch 8) that was 6.9 k + 8.2 k cycles around 8.2 k cycles of MFMA per tile.
//   * The accumulator tile goes through LDS for the row-major epilogue (16-byte stores).  Un-split items (DEFER, the default): a
//     dedicated 16 KB tile next to the ring, so that the stores can be issued later (below); ring of 2 + tile = 48 KB = 3
//     workgroups per CU.  K-split items (and DEFER = false): the ring stage that was consumed LAST, the only stage no DMA targets
//     until the next item's first iteration -- 64x64 fp32 is exactly one 16 KB stage.
//   * Lean code (10 KB against 36 KB for conv_igemm_f32<64,64,...>): ReLU / identity only and NO fused BatchNorm statistics.  A
//     launch that carries them (ym_conv_desc.bn_sum: the training forward and most data gradients) stays on the per-item kernel:
//     an instantiation with the statistics epilogue was built and measured 3x SLOWER than that kernel (150 vs 50 us for the
//     layer3 data gradients, MFMA-busy 0.27): the column sums re-read dout / y / out from L2 and add fp64 atomics per item, and
//     a persistent workgroup sits through that latency itself (3 resident workgroups per CU, phases correlated), where the
//     per-item kernel has a fourth workgroup and freshly dispatched ones to cover it.
//
// Work decomposition, K-slice exchange (agent-scope `sc1` stores + arrival counter, last arriver sums in slice order), tail split
// and XCD-aware tile order are those of conv_mfma.hip: item b here is blockIdx.x == b there.  With grid % 8 == 0 a workgroup's
// items all map to its own XCD's chunk of the tile space.
//
// vmcnt discipline: the ring waits are `s_waitcnt vmcnt((AR+BR)(D-1)) lgkmcnt(0)` + `s_barrier`.  Loads return in order, so a
// counted wait is safe as long as at least that many LOADS are younger than the tile it waits for; the epilogue's own loads and
// stores in between only make it wait longer (a store that completes early lowers the count of a wave whose outstanding loads
// are still the youngest ones).
#include "conv_common.h"

using namespace ymk;

namespace {

// s_waitcnt immediate for gfx9: vmcnt[3:0] = bits 3:0, expcnt = bits 6:4, lgkmcnt = bits 11:8, vmcnt[5:4] = bits 15:14
constexpr int waitcnt_imm(int vm, int lgkm) { return (vm & 15) | ((vm >> 4) << 14) | (7 << 4) | ((lgkm & 15) << 8); }

// workgroup barrier that does NOT drain outstanding global loads / DMA (a __syncthreads() may): LDS traffic only
__device__ __forceinline__ void wg_sync_lds() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// DEFER: the un-split item's epilogue is split in two.  Part 1, at the end of its K loop, only moves the accumulators into the
// dedicated LDS tile and requests the residual / scale / shift operands; part 2 (read the tile row-major, affine + residual + ReLU,
// 16-byte stores) runs inside the NEXT item's K loop, right after the barrier that ends its first K tile -- the barrier the tile
// hand-off needs anyway.  The MFMAs of item i+1 start immediately after item i's last one.  Measured: +2 % (49.6 vs 50.8 us on the
// 2320-tile layer3 conv at batch 8): co-resident workgroups cover each other's epilogues better than their phase-locked start
// suggests.
// ALIGNED: every work item has a multiple of NS K tiles (checked on the host), so every item starts in ring stage 0 and the K loop
// is unrolled by the ring depth with literal stages: fragment reads and DMA destinations are register + immediate, no address VALU
// in the loop (the run-time-stage loop below needs 10 per K tile: VALU instructions take MFMA issue slots, conv_common.h).
template <int BM, int BN, int MODE, int NS, bool DEFER, bool ALIGNED>
__global__ __launch_bounds__(256) void conv_igemm_pers(const ConvP p) {
    static_assert(BM == 64 && BN == 64, "the accumulator staging aliases one ring stage: (BM + BN) * 128 B == BM * BN * 4 B");
    static_assert(MODE == 0 || MODE == 2, "Cin % 32 == 0 convolution / data gradient");
    static_assert(NS >= 2 && NS <= 12, "ring depth");
    constexpr int AR = BM / 32, BR = BN / 32;   // staging rows per thread
    constexpr int RP = 32;                      // LDS floats per tile row (unpadded: the DMA places lane l's 16 bytes at base + 16 l)
    constexpr int D = NS - 1;                   // prefetch distance in K tiles
    constexpr int STAGE = (BM + BN) * RP;       // floats per ring stage: [BM rows of A | BN rows of W]
    // PF: software-pipelined fragments (ring of 3+).  The LDS -> register reads of K group g+1 are issued before the MFMAs of group
    // g and those of the NEXT tile's first group before the end-of-tile barrier, so a wave that has its SIMD to itself never waits
    // on LDS between MFMAs; the next tile must then have landed one barrier early: D-2 tiles may still be in flight at a barrier.
    constexpr bool PF = NS >= 3;
    static_assert((AR + BR) * (D - 1) < 64, "vmcnt field");
    constexpr int WAIT = waitcnt_imm((AR + BR) * (PF ? D - 2 : D - 1), 0);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* C2 = smem + NS * STAGE;                                   // DEFER: the finished item's accumulator tile (BM x BN fp32)
    int* s_flag = reinterpret_cast<int*>(smem + NS * STAGE + (DEFER ? BM * BN : 0));

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: the DMA's LDS destination goes through M0
    const int wm = wave >> 1, wn = wave & 1;
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, p.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_ws = __builtin_amdgcn_make_buffer_rsrc((void*)p.ws, 0, p.ws_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_res =
        __builtin_amdgcn_make_buffer_rsrc((void*)p.residual, 0, p.residual ? (unsigned)((size_t)p.M * p.Cout * 4) : 0u, 0x00020000);

    // ---- work items (block-uniform) ---------------------------------------------------------------
    struct Item { int id, ks, nks, kt_beg, kt_end, tile_m, tile_n; };
    auto decode = [&](int b) __attribute__((always_inline)) {
        Item it;
        unsigned q, r;
        int ktps;
        if (b < p.main_blocks) {
            p.fd_ksplit.divmod((unsigned)ym_xcd_remap(b, p.main_blocks), q, r);
            it.id = (int)q; it.ks = (int)r; it.nks = p.ksplit; ktps = p.kt_per_split;
        } else {                                  // tail tiles, split finer (ym_conv_desc.tail_tiles)
            p.fd_tail.divmod((unsigned)(b - p.main_blocks), q, r);
            it.id = p.main_tiles + (int)q; it.ks = (int)r; it.nks = p.tail_split; ktps = p.tail_ktps;
        }
        ym_tile_decode(p, it.id, it.tile_m, it.tile_n);
        it.kt_beg = it.ks * ktps;
        it.kt_end = min(p.nkt, it.kt_beg + ktps);
        return it;
    };

    // ---- the loader: cursor (item, K tile) + this thread's staging coordinates for that item ----------
    const int c4 = (tid & 7) ^ ((tid >> 4) & 7);   // which float4 of the 32-float K row this lane fetches (XOR swizzle, see conv_mfma.hip)
    const int rbase = tid >> 3;                    // 0..31
    int a_pix[AR], a_ih0[AR], a_iw0[AR];
    unsigned wrow[BR], a_off[AR];              // per-lane byte offsets; the K position inside the tap / the filter row goes in `soffset`
    int kh = 0, kw = 0, c0 = 0;
    bool tap_dirty = true;
    int ld_item = (int)blockIdx.x, ld_kt = 0, ld_end = 0;

    auto setup_loader = [&](int b) __attribute__((always_inline)) {
        tap_dirty = true;
        // past the last item the ring keeps its load count and the loads read zeros: an item whose rows all lie beyond M / Cout
        const bool live = b < p.total_items;
        const Item it = decode(live ? b : 0);
        const int m0 = it.tile_m * BM, n0 = it.tile_n * BN;
#pragma unroll
        for (int i = 0; i < AR; ++i) {
            const int m = m0 + rbase + 32 * i;
            if (live && m < p.M) {
                unsigned ub, urem, uoh, uow;
                p.fd_howo.divmod((unsigned)m, ub, urem);
                p.fd_wo.divmod(urem, uoh, uow);
                const int b_ = (int)ub, oh = (int)uoh, ow = (int)uow;
                if (MODE == 2) {
                    a_ih0[i] = oh + p.pad;
                    a_iw0[i] = ow + p.pad;
                    a_pix[i] = b_ * p.H;
                } else {
                    a_ih0[i] = oh * p.stride - p.pad;
                    a_iw0[i] = ow * p.stride - p.pad;
                    a_pix[i] = (b_ * p.H + a_ih0[i]) * p.W + a_iw0[i];
                }
            } else {
                a_ih0[i] = -(1 << 20);
                a_iw0[i] = -(1 << 20);
                a_pix[i] = 0;
            }
        }
#pragma unroll
        for (int i = 0; i < BR; ++i) {
            const int n = n0 + rbase + 32 * i;
            wrow[i] = (live && n < p.Cout) ? (unsigned)((n * p.Kpad + c4 * 4) * 4) : p.w_bytes;      // past Cout: parked at the buffer's end
        }
        unsigned tap, uc0, ukh, ukw;
        p.fd_cin.divmod((unsigned)(it.kt_beg * BK), tap, uc0);
        p.fd_kw.divmod(tap, ukh, ukw);
        c0 = (int)uc0; kh = (int)ukh; kw = (int)ukw;
        ld_kt = it.kt_beg; ld_end = live ? it.kt_end : 0x7FFFFFFF;
    };

    // next K tile of the stream -> ring stage `stage` (asynchronous, tracked by vmcnt: AR + BR loads per lane, always).  No VALU
    // work per tile (VALU instructions take MFMA issue slots, tools/micro/mfma_lds.hip): the lane offsets change only with the filter
    // tap, the block-uniform K offsets ride in the instruction's SGPR offset (the range check does not wrap at 32 bits, so a masked
    // lane stays out of range), and the LDS destination is scalar.
    auto dma_next = [&](int stage) __attribute__((always_inline)) {
        typedef __attribute__((address_space(3))) void* lds_ptr;
        if (ld_kt == ld_end) {                     // block-uniform
            ld_item += (int)gridDim.x;
            setup_loader(ld_item);
        }
        if (tap_dirty) {                           // block-uniform: the filter tap changed (never inside a 1x1 conv's item)
            tap_dirty = false;
#pragma unroll
            for (int i = 0; i < AR; ++i) {
                if (MODE == 0) {
                    const int ih = a_ih0[i] + kh, iw = a_iw0[i] + kw;
                    const bool ok = (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
                    a_off[i] = (unsigned)(((a_pix[i] + kh * p.W + kw) * p.Cin + c4 * 4) * 4) | (ok ? 0u : OOB);
                } else {
                    const int sh = p.stride >> 1, smask = p.stride - 1;     // stride is 1 or 2
                    const int th = a_ih0[i] - kh, tw = a_iw0[i] - kw;
                    const int yh = th >> sh, yw = tw >> sh;
                    const bool ok = th >= 0 && tw >= 0 && ((th | tw) & smask) == 0 && yh < p.H && yw < p.W;
                    a_off[i] = (unsigned)((((a_pix[i] + yh) * p.W + yw) * p.Cin + c4 * 4) * 4) | (ok ? 0u : OOB);
                }
            }
        }
        float* a = smem + stage * STAGE + 8 * wave * RP;        // wave-uniform; lane l lands at +16 l bytes = row l / 8, slot l % 8
        float* b = a + BM * RP;
#pragma unroll
        for (int i = 0; i < AR; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (lds_ptr)(a + 32 * i * RP), 16, (int)a_off[i], c0 * 4, 0, 0);
#pragma unroll
        for (int i = 0; i < BR; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_ptr)(b + 32 * i * RP), 16, (int)wrow[i], ld_kt * BK * 4, 0, 0);
        ++ld_kt;
        c0 += BK;
        if (c0 >= p.Cin) {
            c0 = 0;
            if (++kw == p.KW) { kw = 0; ++kh; }
            tap_dirty = true;
        }
    };

    // ---- fragments: lane half h takes k = 8g + 4h + s of K group g (one ds_read_b128 per operand row feeds four MFMA steps) ----
    const int frag_row = lane & 31, khalf = lane >> 5;
    const int a_frag_off = (wm * 32 + frag_row) * RP;
    const int b_frag_off = BM * RP + (wn * 32 + frag_row) * RP;
    // this lane's fragment of K group g in stage 0: 8 LDS address registers; stage s = + s * STAGE, a literal in the ALIGNED loops
    // (explicit LDS pointers: through generic pointers kept in an array hipcc fell back to flat loads)
    typedef const __attribute__((address_space(3))) f32x4* lds_frag_ptr;
    lds_frag_ptr pa[4], pb[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int goff = ((2 * g + khalf) ^ ((frag_row >> 1) & 7)) * 4;
        pa[g] = (lds_frag_ptr)(smem + a_frag_off + goff);
        pb[g] = (lds_frag_ptr)(smem + b_frag_off + goff);
    }
    f32x16 acc, acc_odd;         // two accumulators take alternate K steps: consecutive MFMAs of the lone wave are independent

    auto compute = [&](int stage) __attribute__((always_inline)) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 fa = pa[g][stage * (STAGE / 4)];
            const f32x4 fb = pb[g][stage * (STAGE / 4)];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[0], fb[0], acc, 0, 0, 0);
            acc_odd = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[1], fb[1], acc_odd, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[2], fb[2], acc, 0, 0, 0);
            acc_odd = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[3], fb[3], acc_odd, 0, 0, 0);
        }
    };

    auto read_frag = [&](int stage, int g, f32x4& fa, f32x4& fb) __attribute__((always_inline)) {
        fa = pa[g][stage * (STAGE / 4)];
        fb = pb[g][stage * (STAGE / 4)];
    };
    auto mfma_group = [&](const f32x4& fa, const f32x4& fb) __attribute__((always_inline)) {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[0], fb[0], acc, 0, 0, 0);
        acc_odd = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[1], fb[1], acc_odd, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[2], fb[2], acc, 0, 0, 0);
        acc_odd = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[3], fb[3], acc_odd, 0, 0, 0);
    };

    // ---- epilogue geometry (row-major: one float4 of a row per lane) ---------------------------------
    constexpr int C4 = BN / 4;                 // float4 per row
    constexpr int RPP = 256 / C4;              // rows per pass
    constexpr int ENR = BM / RPP;              // rows per lane
    const int col4 = tid % C4, row0 = tid / C4;
    const size_t slice = (size_t)p.M * p.Cout;
    const int act_relu = p.seg[0].act == YM_ACT_RELU;

    // ---- deferred epilogue (DEFER): the finished item whose stores are still to be issued ----------------
    bool pend = false, c_dirty = false;        // c_dirty: a wave may still be reading C2 (no barrier since part 2)
    int pend_m0 = 0, pend_n0 = 0;
    f32x4 pend_sc = {1.f, 1.f, 1.f, 1.f}, pend_sh = {0.f, 0.f, 0.f, 0.f}, pend_res[ENR];
    auto flush_pending = [&]() __attribute__((always_inline)) {
        const int pn = pend_n0 + col4 * 4;
        if (pn < p.Cout) {
            float* outb = p.seg[0].out + pn;
#pragma unroll
            for (int rk = 0; rk < ENR; ++rk) {
                const int row = row0 + rk * RPP;
                const int m = pend_m0 + row;
                if (m < p.M) {
                    f32x4 v = *reinterpret_cast<const f32x4*>(C2 + row * BN + col4 * 4);
                    v = __builtin_elementwise_fma(v, pend_sc, pend_sh);
                    v += pend_res[rk];                                   // zeros when there is no residual
                    if (act_relu) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = v[e] < 0.f ? 0.f : v[e];
                    }
                    *reinterpret_cast<f32x4*>(outb + (size_t)m * p.Cout) = v;
                }
            }
        }
        pend = false;
        c_dirty = true;
    };

    // ---- the stream ---------------------------------------------------------------------------------
    setup_loader(ld_item);
#pragma unroll
    for (int d = 0; d < D; ++d) dma_next(d);
    __builtin_amdgcn_s_waitcnt(WAIT);
    __builtin_amdgcn_s_barrier();
    int buf = 0, nb = D;                       // nb = (buf + D) % NS = the stage consumed one iteration ago

    for (int bid = (int)blockIdx.x; bid < p.total_items; bid += (int)gridDim.x) {
        const Item it = decode(bid);
        const int m0 = it.tile_m * BM, n0 = it.tile_n * BN, nks = it.nks, ks = it.ks;
        const int n = n0 + col4 * 4;
        // epilogue operands of the un-split item are requested BEFORE its K loop (they return behind the D tiles already in flight)
        const bool direct = nks == 1;
        f32x4 pre_sc = {1.f, 1.f, 1.f, 1.f}, pre_sh = {0.f, 0.f, 0.f, 0.f}, pre_res[ENR];
        if (direct && !DEFER) {
            if (n < p.Cout) {
                if (p.scale) pre_sc = *reinterpret_cast<const f32x4*>(p.scale + n);
                if (p.shift) pre_sh = *reinterpret_cast<const f32x4*>(p.shift + n);
            }
#pragma unroll
            for (int k = 0; k < ENR; ++k) {
                const int m = m0 + row0 + k * RPP;
                pre_res[k] = buf_ld16(rs_res, (m < p.M && n < p.Cout) ? (unsigned)(((size_t)m * p.Cout + n) * 4) : OOB);
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[r] = 0.f; acc_odd[r] = 0.f; }
        const int nt = it.kt_end - it.kt_beg;
#ifdef YM_TRACE
        // ablations of the trace build (YM_PERS_ABL -> p.bnb_relu; tools/pers_ablation.py): 1 = operand stream only, 2 = LDS reads +
        // MFMAs only, 3 = MFMAs only (register operands), 5 = MFMAs only and NO per-tile barrier
        if (p.bnb_relu >= 1 && p.bnb_relu <= 5) {
            const f32x4 ra = {1.f, 2.f, 3.f, 4.f}, rb = {.5f, .25f, .125f, 1.f};
            for (int t = 0; t < nt; ++t) {
                if (p.bnb_relu == 1) dma_next(nb);
                else if (p.bnb_relu == 2) compute(buf);
                else {
#pragma unroll
                    for (int g = 0; g < 4; ++g) mfma_group(ra, rb);
                }
                if (p.bnb_relu != 5) {
                    __builtin_amdgcn_s_waitcnt(WAIT);
                    __builtin_amdgcn_s_barrier();
                }
                nb = buf;
                buf = buf == NS - 1 ? 0 : buf + 1;
                if constexpr (DEFER) {
                    if (t == 0) { if (pend) flush_pending(); } else c_dirty = false;
                }
            }
        } else
#endif
        if constexpr (ALIGNED && PF) {
            f32x4 fa0, fb0, fa1, fb1;
            read_frag(0, 0, fa0, fb0);             // (this tile landed at least one barrier ago)
            // (the deferred stores of the previous item are issued between the first and the second tile, OUTSIDE the generic lambda:
            // with flush_pending() inside it hipcc kept the closure in scratch memory)
            auto tile = [&](auto SC, int t) __attribute__((always_inline)) {
                    constexpr int B0 = decltype(SC)::value, B1 = (B0 + 1) % NS, BP = (B0 + NS - 1) % NS;
                    dma_next(BP);
                    read_frag(B0, 1, fa1, fb1);
                    __builtin_amdgcn_sched_barrier(0);
                    mfma_group(fa0, fb0);
                    __builtin_amdgcn_sched_barrier(0);
                    read_frag(B0, 2, fa0, fb0);
                    __builtin_amdgcn_sched_barrier(0);
                    mfma_group(fa1, fb1);
                    __builtin_amdgcn_sched_barrier(0);
                    read_frag(B0, 3, fa1, fb1);
                    __builtin_amdgcn_sched_barrier(0);
                    mfma_group(fa0, fb0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (B0 + 1 < NS || t + NS < nt) read_frag(B1, 0, fa0, fb0);      // first group of this item's next tile
                    __builtin_amdgcn_sched_barrier(0);
                    mfma_group(fa1, fb1);
                    __builtin_amdgcn_s_waitcnt(WAIT);
                    __builtin_amdgcn_s_barrier();
            };
            for (int t = 0; t < nt; t += NS) {
                tile(IC<0>{}, t);
                if constexpr (DEFER) {             // every wave's copy of the previous item's tile is in C2 now: issue its stores
                    if (t == 0) { if (pend) flush_pending(); } else c_dirty = false;
                }
                static_for<1, NS>([&](auto SC) __attribute__((always_inline)) { tile(SC, t); });
                if constexpr (DEFER) c_dirty = false;
            }
        } else if constexpr (ALIGNED) {
            auto tile = [&](auto SC) __attribute__((always_inline)) {
                constexpr int B0 = decltype(SC)::value;
                dma_next((B0 + NS - 1) % NS);
                compute(B0);
                __builtin_amdgcn_s_waitcnt(WAIT);
                __builtin_amdgcn_s_barrier();
            };
            for (int t = 0; t < nt; t += NS) {
                tile(IC<0>{});
                if constexpr (DEFER) {
                    if (t == 0) { if (pend) flush_pending(); } else c_dirty = false;
                }
                static_for<1, NS>([&](auto SC) __attribute__((always_inline)) { tile(SC); });
                if constexpr (DEFER) c_dirty = false;
            }
        } else if constexpr (PF) {
            f32x4 fa0, fb0, fa1, fb1;
            read_frag(buf, 0, fa0, fb0);           // (this tile landed at least one barrier ago)
            for (int t = 0; t < nt; ++t) {
                const int buf1 = buf == NS - 1 ? 0 : buf + 1;
                dma_next(nb);
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
                if (t + 1 < nt) read_frag(buf1, 0, fa0, fb0);      // first group of this item's next tile
                __builtin_amdgcn_sched_barrier(0);
                mfma_group(fa1, fb1);
                __builtin_amdgcn_s_waitcnt(WAIT);
                __builtin_amdgcn_s_barrier();
                nb = buf;
                buf = buf1;
                if constexpr (DEFER) {             // every wave's copy of the previous item's tile is in C2 now: issue its stores
                    if (t == 0) { if (pend) flush_pending(); } else c_dirty = false;
                }
            }
        } else {
            for (int t = 0; t < nt; ++t) {
                dma_next(nb);
                compute(buf);
                __builtin_amdgcn_s_waitcnt(WAIT);      // the next tile of the stream has landed for THIS wave, and this wave's ds_reads retired ...
                __builtin_amdgcn_s_barrier();          // ... and for every wave: stage `buf` may be re-staged by the next dma_next
                nb = buf;
                buf = buf == NS - 1 ? 0 : buf + 1;
                if constexpr (DEFER) {
                    if (t == 0) { if (pend) flush_pending(); } else c_dirty = false;
                }
            }
        }
        acc += acc_odd;
        if (DEFER && direct) {
            // part 1: park the tile in C2 and request the epilogue operands; the stores follow under the next item's MFMAs
            if (c_dirty) wg_sync_lds();            // (a one-K-tile item: somebody may still be reading the previous tile)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                C2[(wm * 32 + 4 * khalf + (r & 3) + 8 * (r >> 2)) * BN + wn * 32 + frag_row] = acc[r];
            pend_sc = f32x4{1.f, 1.f, 1.f, 1.f};
            pend_sh = f32x4{0.f, 0.f, 0.f, 0.f};
            if (n < p.Cout) {
                if (p.scale) pend_sc = *reinterpret_cast<const f32x4*>(p.scale + n);
                if (p.shift) pend_sh = *reinterpret_cast<const f32x4*>(p.shift + n);
            }
#pragma unroll
            for (int k = 0; k < ENR; ++k) {
                const int m = m0 + row0 + k * RPP;
                pend_res[k] = buf_ld16(rs_res, (m < p.M && n < p.Cout) ? (unsigned)(((size_t)m * p.Cout + n) * 4) : OOB);
            }
            pend_m0 = m0; pend_n0 = n0;
            pend = true;
            c_dirty = false;
            continue;
        }

        // ---- epilogue: C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) ----
        float* C = smem + nb * STAGE;              // the stage consumed last: no DMA targets it before the next item's first iteration
#pragma unroll
        for (int r = 0; r < 16; ++r)
            C[(wm * 32 + 4 * khalf + (r & 3) + 8 * (r >> 2)) * BN + wn * 32 + frag_row] = acc[r];
        wg_sync_lds();
        const bool in_tail = bid >= p.main_blocks;
        const unsigned sb = in_tail ? (unsigned)(BM * BN * 4) : (unsigned)(slice * 4);            // bytes between slices
        const unsigned rs = in_tail ? (unsigned)(BN * 4) : (unsigned)(p.Cout * 4);                 // bytes between rows
        const unsigned ws0 = in_tail ? (unsigned)(it.id - p.main_tiles) * (unsigned)nks * sb + (unsigned)(col4 * 16)
                                     : (unsigned)m0 * rs + (unsigned)(n * 4);                      // (row 0, this lane's float4), slice 0
        bool finish = true;
        if (nks > 1) {
            // K-slice exchange between workgroups (on different XCDs): sc1 stores / loads + an arrival counter (conv_mfma.hip)
            if (n < p.Cout) {
                const unsigned base = ws0 + (unsigned)ks * sb;
#pragma unroll
                for (int k = 0; k < ENR; ++k) {
                    const int row = row0 + k * RPP;
                    if (m0 + row < p.M) buf_st16_sc1(rs_ws, base + (unsigned)row * rs, *reinterpret_cast<const f32x4*>(C + row * BN + col4 * 4));
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // EVERY writing wave drains its slice stores before the arrival
            wg_sync_lds();
            if (tid == 0) {
                int* cnt = p.counters + it.tile_m * p.tiles_n + it.tile_n;
                const int old = __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const int last = old == nks - 1;
                *s_flag = last;
                if (last) __hip_atomic_store(cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
            }
            wg_sync_lds();
            finish = *s_flag != 0;
        }
        if (finish && n < p.Cout) {
            f32x4 sc = pre_sc, sh = pre_sh;
            if (!direct) {
                if (p.scale) sc = *reinterpret_cast<const f32x4*>(p.scale + n);
                if (p.shift) sh = *reinterpret_cast<const f32x4*>(p.shift + n);
            }
            float* outb = p.seg[0].out + n;
#pragma unroll
            for (int rk = 0; rk < ENR; ++rk) {
                const int row = row0 + rk * RPP;
                const int m = m0 + row;
                if (m < p.M) {
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
                    } else {
                        v = *reinterpret_cast<const f32x4*>(C + row * BN + col4 * 4);
                    }
                    v = __builtin_elementwise_fma(v, sc, sh);
                    if (direct) v += pre_res[rk];                       // zeros when there is no residual
                    else if (p.residual) v += *reinterpret_cast<const f32x4*>(p.residual + (size_t)m * p.Cout + n);
                    if (act_relu) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = v[e] < 0.f ? 0.f : v[e];   // NaN stays NaN, like torch.relu
                    }
                    *reinterpret_cast<f32x4*>(outb + (size_t)m * p.Cout) = v;
                }
            }
        }
        wg_sync_lds();                             // stage `nb` is free again: the next item's first dma_next re-stages it
    }
    if constexpr (DEFER) {
        if (pend) {                                // the last item's stores
            wg_sync_lds();
            flush_pending();
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the run-ahead loads of "no work left" still target this workgroup's LDS
}

template <int BM, int BN, int MODE, int NS, bool DEFER, bool ALIGNED>
int launch_pers(const ConvP& p, int grid, hipStream_t st) {
    const size_t lds = ym_conv_pers_lds_bytes(BM, BN, NS, DEFER);
    static YmLdsAttr attr = {};
    if (int rc = ym_ensure_dyn_lds(attr, reinterpret_cast<const void*>(conv_igemm_pers<BM, BN, MODE, NS, DEFER, ALIGNED>), lds, "conv_igemm_pers")) return rc;
    hipLaunchKernelGGL((conv_igemm_pers<BM, BN, MODE, NS, DEFER, ALIGNED>), dim3(grid), dim3(256), lds, st, p);
    return ym_check_launch("conv_igemm_pers");
}

template <int MODE, bool DEFER>
int launch_pers_ns(const ConvP& p, int ns, int grid, hipStream_t st) {
    // every item a multiple of `ns` K tiles?  (main slices: kt_per_split each, the last one the rest; tail slices: tail_ktps)
    const bool main_ok = p.main_blocks == 0 || (p.nkt % ns == 0 && (p.ksplit == 1 || p.kt_per_split % ns == 0));
    const bool tail_ok = p.total_items == p.main_blocks || (p.nkt % ns == 0 && p.tail_ktps % ns == 0);
    const bool al = main_ok && tail_ok && ns <= 4;
    switch (ns) {
        case 2: return al ? launch_pers<64, 64, MODE, 2, DEFER, true>(p, grid, st) : launch_pers<64, 64, MODE, 2, DEFER, false>(p, grid, st);
        case 3: return al ? launch_pers<64, 64, MODE, 3, DEFER, true>(p, grid, st) : launch_pers<64, 64, MODE, 3, DEFER, false>(p, grid, st);
        case 4: return al ? launch_pers<64, 64, MODE, 4, DEFER, true>(p, grid, st) : launch_pers<64, 64, MODE, 4, DEFER, false>(p, grid, st);
        case 6: return launch_pers<64, 64, MODE, 6, DEFER, false>(p, grid, st);
        case 8: return launch_pers<64, 64, MODE, 8, DEFER, false>(p, grid, st);
        default: ym_set_error("conv(persistent): ring depth %d not built (2, 3, 4, 6, 8)", ns); return YM_EINVAL;
    }
}

}  // namespace

size_t ym_conv_pers_lds_bytes(int bm, int bn, int ns, bool defer) {
    return (size_t)ns * (bm + bn) * 32 * sizeof(float) + (defer ? (size_t)bm * bn * sizeof(float) : 0) + 16;
}

int ym_launch_conv_pers(const ConvP& p, int bm, int bn, int mode, int ns, bool defer, int grid, hipStream_t st) {
    if (bm != 64 || bn != 64 || (mode != 0 && mode != 2)) {
        ym_set_error("conv(persistent): 64x64 tile, convolution or data gradient only (got %dx%d mode %d)", bm, bn, mode);
        return YM_EINVAL;
    }
    if (defer) return mode == 0 ? launch_pers_ns<0, true>(p, ns, grid, st) : launch_pers_ns<2, true>(p, ns, grid, st);
    return mode == 0 ? launch_pers_ns<0, false>(p, ns, grid, st) : launch_pers_ns<2, false>(p, ns, grid, st);
}
