// EXPERIMENT (round 3): fp32-grade GEMM on the bf16 MFMA with PRE-SPLIT operands.
// Every fp32 operand x is stored as three bf16 planes x = p0 + p1 + p2 (24 significant bits), laid out
//     [row][K/32][plane][32 bf16]            (192 contiguous bytes per row and 32-deep K block)
// so BOTH operands go global -> LDS by DMA (buffer_load ... lds) with no conversion VALU, no staging registers and no ds_write,
// and the six cross products of significance >= 2^-24 run on v_mfma_f32_32x32x16_bf16 (6 x 32 cycles per 16 k against
// 8 x 64 cycles of v_mfma_f32_32x32x2_f32: 2.67x less matrix-pipe time).  128x128 workgroup tile, four waves of 64x64.
//     pg_split:  fp32 [M][K] -> planes            pg_gemm:  out[m][n] = act(sum_k A[m][k] W[n][k] * scale[n] + shift[n])
// Build: hipcc -O3 --offload-arch=gfx950 -shared -fPIC tools/micro/planes_gemm.hip -o tools/micro/libplanes_gemm.so
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

namespace {

constexpr int waitcnt_imm(int vm, int lgkm) { return (vm & 15) | ((vm >> 4) << 14) | (7 << 4) | ((lgkm & 15) << 8); }

__global__ __launch_bounds__(256) void k_split(const float* __restrict__ x, uint16_t* __restrict__ out, size_t rows, int K) {
    const int kb = K / 32;
    const size_t total4 = rows * (size_t)K / 4;
    for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < total4; q += (size_t)gridDim.x * 256) {
        const size_t e = q * 4;
        const size_t r = e / K;
        const int k = (int)(e - r * K);
        f32x4 v = *reinterpret_cast<const f32x4*>(x + e);
        uint16_t* dst = out + ((r * kb + k / 32) * 3) * 32 + (k & 31);
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
            const unsigned p01 = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{v[0], v[1]}, bf16x2_t));
            const unsigned p23 = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{v[2], v[3]}, bf16x2_t));
            *reinterpret_cast<uint2*>(dst + pl * 32) = uint2{p01, p23};
            v[0] -= __uint_as_float(p01 << 16);
            v[1] -= __uint_as_float(p01 & 0xFFFF0000u);
            v[2] -= __uint_as_float(p23 << 16);
            v[3] -= __uint_as_float(p23 & 0xFFFF0000u);
        }
    }
}

// NS: ring depth in K tiles of 16 (one v_mfma_f32_32x32x16_bf16 k block): a stage is (128 + 128) rows x 3 planes x 32 B = 24 KB, so
// NS = 6 keeps 5 tiles (120 KB, ~3800 MFMA cycles) in flight per workgroup -- with K tiles of 32 (48 KB) the LDS holds a ring of
// 3 only, and two tiles of look-ahead (3000 cycles) are less than one L2-miss round trip: 40 % MFMA utilisation measured.
// PF: the fragments of the next tile are read before the end-of-tile barrier (needs that tile landed one barrier early).
template <int NS, bool PF>
__global__ __launch_bounds__(256) void k_gemm(const uint16_t* __restrict__ A, const uint16_t* __restrict__ W, float* __restrict__ out,
                                              int M, int N, int K, const float* __restrict__ scale, const float* __restrict__ shift, int relu,
                                              int tiles_n) {
    constexpr int BM = 128, BN = 128;
    constexpr int PLANE = 128 * 32;             // bytes of one plane of one operand in a stage
    constexpr int OPND = 3 * PLANE;             // 12288
    constexpr int STAGE = 2 * OPND;             // 24576
    constexpr int D = NS - 1;
    constexpr int LPS = 6;                      // DMA loads per thread per stage
    constexpr int WAIT = waitcnt_imm(LPS * (PF ? D - 2 : D - 1), 0);
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int tile_m = blockIdx.x / tiles_n, tile_n = blockIdx.x - tile_m * tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int KB = K / 32, KT = K / 16;
    const unsigned a_bytes = (unsigned)((size_t)M * KB * 192), w_bytes = (unsigned)((size_t)N * KB * 192);
    const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)W, 0, w_bytes, 0x00020000);

    // ---- DMA coordinates: per operand and stage 12 instructions (plane q / 4, rows 32 (q % 4) .. +31, one 32-byte line each);
    // wave w issues q = 3w .. 3w+2.  lane l -> row + l / 2, slot l % 2, which holds chunk slot ^ ((row >> 3) & 1) ----
    unsigned a_off[3], w_off[3], a_mask[3], w_mask[3];
    int lds_off[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int q = wave * 3 + i, plane = q >> 2, row = (q & 3) * 32 + (lane >> 1), slot = lane & 1;
        const int chunk = slot ^ ((row >> 3) & 1);
        const int m = m0 + row, n = n0 + row;
        a_off[i] = (unsigned)((size_t)(m < M ? m : 0) * KB * 192 + plane * 64 + chunk * 16);
        w_off[i] = (unsigned)((size_t)(n < N ? n : 0) * KB * 192 + plane * 64 + chunk * 16);
        a_mask[i] = m < M ? 0u : 0xFFFFFFF0u;      // offset | 0xFFFFFFF0 is beyond the buffer: the load returns zeros
        w_mask[i] = n < N ? 0u : 0xFFFFFFF0u;
        lds_off[i] = plane * PLANE + (q & 3) * 32 * 32;
    }
    auto dma_tile = [&](int kt, int stage) __attribute__((always_inline)) {
        typedef __attribute__((address_space(3))) void* lds_ptr;
        char* base = smem + stage * STAGE;
        const unsigned kb = (unsigned)((kt >> 1) * 192 + (kt & 1) * 32), tmask = kt < KT ? 0u : 0xFFFFFFF0u;      // past the end: zeros
#pragma unroll
        for (int i = 0; i < 3; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (lds_ptr)(base + lds_off[i]), 16, (int)((a_off[i] + kb) | a_mask[i] | tmask), 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 3; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_ptr)(base + OPND + lds_off[i]), 16, (int)((w_off[i] + kb) | w_mask[i] | tmask), 0, 0, 0);
    };

    // ---- fragments: lane l supplies row l & 31, k = 8 (l >> 5) .. + 7 of the 16-deep block ----
    const int r32 = lane & 31, kh = lane >> 5;
    const int coff = (kh ^ ((r32 >> 3) & 1)) * 16;
    int a_row_off[2], b_row_off[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        a_row_off[i] = (wm * 64 + i * 32 + r32) * 32 + coff;
        b_row_off[i] = OPND + (wn * 64 + i * 32 + r32) * 32 + coff;
    }

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    auto read_frags = [&](int stage, bf16x8 (&fa)[2][3], bf16x8 (&fb)[2][3]) __attribute__((always_inline)) {
        const char* base = smem + stage * STAGE;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                fa[i][p] = *reinterpret_cast<const bf16x8*>(base + p * PLANE + a_row_off[i]);
                fb[i][p] = *reinterpret_cast<const bf16x8*>(base + p * PLANE + b_row_off[i]);
            }
    };
    auto mfma_block = [&](const bf16x8 (&fa)[2][3], const bf16x8 (&fb)[2][3]) __attribute__((always_inline)) {
        // cross terms by rising significance (pa + pb = 2, 1, 0): the small ones enter the accumulator first
#pragma unroll
        for (int sig = 2; sig >= 0; --sig)
#pragma unroll
            for (int pa = 0; pa <= sig; ++pa)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][pa], fb[j][sig - pa], acc[i][j], 0, 0, 0);
    };

#pragma unroll
    for (int d = 0; d < D; ++d) dma_tile(d, d);
    __builtin_amdgcn_s_waitcnt(WAIT);
    __builtin_amdgcn_s_barrier();
    int buf = 0, nb = D;
    if constexpr (PF) {
        bf16x8 fa0[2][3], fb0[2][3], fa1[2][3], fb1[2][3];
        read_frags(0, fa0, fb0);
        for (int t = 0; t < KT; t += 2) {          // two tiles per trip: the fragment sets alternate without copies
            int buf1 = buf == NS - 1 ? 0 : buf + 1;
            dma_tile(t + D, nb);
            read_frags(buf1, fa1, fb1);            // tile t+1 landed one barrier ago
            __builtin_amdgcn_sched_barrier(0);
            mfma_block(fa0, fb0);
            __builtin_amdgcn_s_waitcnt(WAIT);
            __builtin_amdgcn_s_barrier();
            nb = buf; buf = buf1;
            if (t + 1 >= KT) break;
            buf1 = buf == NS - 1 ? 0 : buf + 1;
            dma_tile(t + 1 + D, nb);
            read_frags(buf1, fa0, fb0);
            __builtin_amdgcn_sched_barrier(0);
            mfma_block(fa1, fb1);
            __builtin_amdgcn_s_waitcnt(WAIT);
            __builtin_amdgcn_s_barrier();
            nb = buf; buf = buf1;
        }
    } else {
        const int abl = relu >> 4;
        for (int t = 0; t < KT; ++t) {
            if (abl != 2) dma_tile(t + D, nb);
            if (abl != 1) {
                bf16x8 fa[2][3], fb[2][3];
                read_frags(buf, fa, fb);
                mfma_block(fa, fb);
            }
            __builtin_amdgcn_s_waitcnt(WAIT);
            __builtin_amdgcn_s_barrier();
            nb = buf;
            buf = buf == NS - 1 ? 0 : buf + 1;
        }
    }
    __builtin_amdgcn_s_waitcnt(waitcnt_imm(0, 0));          // past-the-end DMAs still target LDS
    __builtin_amdgcn_s_barrier();

    // ---- epilogue: stage the 128x128 tile through LDS, row-major float4 stores ----
    constexpr int CP = BN + 4;
    static_assert(NS * STAGE >= BM * CP * 4, "the accumulator staging needs 66 KB of the ring");
    float* C = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                C[(wm * 64 + i * 32 + 4 * kh + (r & 3) + 8 * (r >> 2)) * CP + wn * 64 + j * 32 + r32] = acc[i][j][r];
    __syncthreads();
    const int col4 = tid & 31, row0 = tid >> 5;
    const int n = n0 + col4 * 4;
    if (n < N) {
        f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
        if (scale) sc = *reinterpret_cast<const f32x4*>(scale + n);
        if (shift) sh = *reinterpret_cast<const f32x4*>(shift + n);
#pragma unroll
        for (int rk = 0; rk < 16; ++rk) {
            const int row = row0 + rk * 8, m = m0 + row;
            if (m < M) {
                f32x4 v = *reinterpret_cast<const f32x4*>(C + row * CP + col4 * 4);
                v = __builtin_elementwise_fma(v, sc, sh);
                if (relu & 1) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = v[e] < 0.f ? 0.f : v[e];
                }
                *reinterpret_cast<f32x4*>(out + (size_t)m * N + n) = v;
            }
        }
    }
}

// k_gemm2: the same GEMM with a K loop that issues NO vector instruction besides MFMAs (VALU instructions
// take MFMA issue slots -- k_gemm spends 32-96 of them per 24 MFMAs on offset adds, v_readfirstlane of the DMA destination,
// fragment addresses and, worst, 64 v_accvgpr_mov per pair of tiles that shuffle the accumulators between two register sets).
// Lane offsets are fixed for the whole launch and the K position rides in the load's SGPR offset; the wave index is scalar; the
// ring (NS even) is unrolled by its depth so that every LDS address is register + immediate and the two fragment sets alternate
// statically.  Past-the-end tiles read whatever follows (never consumed; beyond the buffer the load returns zeros).
template <int I> struct IC { static constexpr int value = I; };
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) { f(IC<I>{}); static_for<I + 1, N>(f); }
}

template <int NS>
__global__ __launch_bounds__(256) void k_gemm2(const uint16_t* __restrict__ A, const uint16_t* __restrict__ W, float* __restrict__ out,
                                               int M, int N, int K, const float* __restrict__ scale, const float* __restrict__ shift, int relu,
                                               int tiles_n) {
    static_assert(NS % 2 == 0 && NS >= 4, "even ring: the fragment sets alternate statically");
    constexpr int BM = 128, BN = 128;
    constexpr int PLANE = 128 * 32, OPND = 3 * PLANE, STAGE = 2 * OPND;
    constexpr int D = NS - 1, LPS = 6;
    constexpr int WAIT = waitcnt_imm(LPS * (D - 2), 0);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int tile_m = blockIdx.x / tiles_n, tile_n = blockIdx.x - tile_m * tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int KB = K / 32, KT = K / 16;
    const unsigned a_bytes = (unsigned)((size_t)M * KB * 192), w_bytes = (unsigned)((size_t)N * KB * 192);
    const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)W, 0, w_bytes, 0x00020000);
    unsigned a_off[3], w_off[3];
    int lds_off[3];                              // scalar
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int q = wave * 3 + i, plane = q >> 2, row = (q & 3) * 32 + (lane >> 1), slot = lane & 1;
        const int chunk = slot ^ ((row >> 3) & 1);
        const int m = m0 + row, n = n0 + row;
        a_off[i] = m < M ? (unsigned)((size_t)m * KB * 192 + plane * 64 + chunk * 16) : 0xFFFFFFF0u;
        w_off[i] = n < N ? (unsigned)((size_t)n * KB * 192 + plane * 64 + chunk * 16) : 0xFFFFFFF0u;
        lds_off[i] = plane * PLANE + (q & 3) * 32 * 32;
    }
    typedef __attribute__((address_space(3))) void* lds_ptr;
    auto dma_tile = [&](int kt, int stage) __attribute__((always_inline)) {
        char* base = smem + stage * STAGE;
        const int kb = (kt >> 1) * 192 + (kt & 1) * 32;
#pragma unroll
        for (int i = 0; i < 3; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (lds_ptr)(base + lds_off[i]), 16, (int)a_off[i], kb, 0, 0);
#pragma unroll
        for (int i = 0; i < 3; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_ptr)(base + OPND + lds_off[i]), 16, (int)w_off[i], kb, 0, 0);
    };
    const int r32 = lane & 31, kh = lane >> 5;
    const int coff = (kh ^ ((r32 >> 3) & 1)) * 16;
    typedef const __attribute__((address_space(3))) bf16x8* lds_frag_ptr;
    lds_frag_ptr pa[2], pb[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        pa[i] = (lds_frag_ptr)(smem + (wm * 64 + i * 32 + r32) * 32 + coff);
        pb[i] = (lds_frag_ptr)(smem + OPND + (wn * 64 + i * 32 + r32) * 32 + coff);
    }
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    auto read_frags = [&](int stage, bf16x8 (&fa)[2][3], bf16x8 (&fb)[2][3]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                fa[i][p] = pa[i][(stage * STAGE + p * PLANE) / 16];
                fb[i][p] = pb[i][(stage * STAGE + p * PLANE) / 16];
            }
    };
    auto mfma_block = [&](const bf16x8 (&fa)[2][3], const bf16x8 (&fb)[2][3]) __attribute__((always_inline)) {
#pragma unroll
        for (int sig = 2; sig >= 0; --sig)
#pragma unroll
            for (int pa_ = 0; pa_ <= sig; ++pa_)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][pa_], fb[j][sig - pa_], acc[i][j], 0, 0, 0);
    };
#pragma unroll
    for (int d = 0; d < D; ++d) dma_tile(d, d);
    __builtin_amdgcn_s_waitcnt(WAIT);
    __builtin_amdgcn_s_barrier();
    bf16x8 fa0[2][3], fb0[2][3], fa1[2][3], fb1[2][3];
    read_frags(0, fa0, fb0);
    auto tile = [&](auto SC, int t) __attribute__((always_inline)) {
        constexpr int S = decltype(SC)::value, S1 = (S + 1) % NS, SP = (S + NS - 1) % NS;
        dma_tile(t + D, SP);
        if constexpr (S % 2 == 0) {
            read_frags(S1, fa1, fb1);              // tile t+1 landed one barrier ago
            __builtin_amdgcn_sched_barrier(0);
            mfma_block(fa0, fb0);
        } else {
            read_frags(S1, fa0, fb0);
            __builtin_amdgcn_sched_barrier(0);
            mfma_block(fa1, fb1);
        }
        __builtin_amdgcn_s_waitcnt(WAIT);
        __builtin_amdgcn_s_barrier();
    };
    int t = 0;
    for (; t + NS <= KT; t += NS) static_for<0, NS>([&](auto SC) __attribute__((always_inline)) { tile(SC, t + decltype(SC)::value); });
    static_for<0, NS - 1>([&](auto SC) __attribute__((always_inline)) { if (t + decltype(SC)::value < KT) tile(SC, t + decltype(SC)::value); });
    __builtin_amdgcn_s_waitcnt(waitcnt_imm(0, 0));          // past-the-end DMAs still target LDS
    __builtin_amdgcn_s_barrier();

    constexpr int CP = BN + 4;
    static_assert(NS * STAGE >= BM * CP * 4, "the accumulator staging needs 66 KB of the ring");
    float* C = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                C[(wm * 64 + i * 32 + 4 * kh + (r & 3) + 8 * (r >> 2)) * CP + wn * 64 + j * 32 + r32] = acc[i][j][r];
    __syncthreads();
    const int col4 = tid & 31, row0 = tid >> 5;
    const int n = n0 + col4 * 4;
    if (n < N) {
        f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
        if (scale) sc = *reinterpret_cast<const f32x4*>(scale + n);
        if (shift) sh = *reinterpret_cast<const f32x4*>(shift + n);
#pragma unroll
        for (int rk = 0; rk < 16; ++rk) {
            const int row = row0 + rk * 8, m = m0 + row;
            if (m < M) {
                f32x4 v = *reinterpret_cast<const f32x4*>(C + row * CP + col4 * 4);
                v = __builtin_elementwise_fma(v, sc, sh);
                if (relu & 1) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = v[e] < 0.f ? 0.f : v[e];
                }
                *reinterpret_cast<f32x4*>(out + (size_t)m * N + n) = v;
            }
        }
    }
}

template <int NS>
int launch2(const uint16_t* A, const uint16_t* W, float* out, int M, int N, int K, const float* scale, const float* shift, int relu,
            hipStream_t st) {
    const size_t lds = (size_t)NS * 24576;
    static bool set = false;
    if (!set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm2<NS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        set = true;
    }
    const int tiles_m = (M + 127) / 128, tiles_n = (N + 127) / 128;
    hipLaunchKernelGGL((k_gemm2<NS>), dim3(tiles_m * tiles_n), dim3(256), lds, st, A, W, out, M, N, K, scale, shift, relu, tiles_n);
    return (int)hipGetLastError();
}

template <int NS, bool PF>
int launch(const uint16_t* A, const uint16_t* W, float* out, int M, int N, int K, const float* scale, const float* shift, int relu,
           hipStream_t st) {
    const size_t lds = (size_t)NS * 24576;
    static bool set = false;
    if (!set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm<NS, PF>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        set = true;
    }
    const int tiles_m = (M + 127) / 128, tiles_n = (N + 127) / 128;
    hipLaunchKernelGGL((k_gemm<NS, PF>), dim3(tiles_m * tiles_n), dim3(256), lds, st, A, W, out, M, N, K, scale, shift, relu, tiles_n);
    return (int)hipGetLastError();
}

}  // namespace

extern "C" int pg_split(const float* x, uint16_t* planes, long rows, int K, void* stream) {
    if (K % 32) return -1;
    size_t total4 = (size_t)rows * K / 4;
    int grid = (int)((total4 + 255) / 256);
    if (grid > 65536) grid = 65536;
    hipLaunchKernelGGL(k_split, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, planes, (size_t)rows, K);
    return (int)hipGetLastError();
}

extern "C" int pg_gemm(const uint16_t* A, const uint16_t* W, float* out, int M, int N, int K, const float* scale, const float* shift,
                       int relu, int ns, int pf, void* stream) {
    if (K % 32 || N % 4) return -1;
    hipStream_t st = (hipStream_t)stream;
    if (pf == 2) return ns == 4 ? launch2<4>(A, W, out, M, N, K, scale, shift, relu, st) : launch2<6>(A, W, out, M, N, K, scale, shift, relu, st);
    if (ns == 3) return pf ? launch<3, true>(A, W, out, M, N, K, scale, shift, relu, st) : launch<3, false>(A, W, out, M, N, K, scale, shift, relu, st);
    if (ns == 4) return pf ? launch<4, true>(A, W, out, M, N, K, scale, shift, relu, st) : launch<4, false>(A, W, out, M, N, K, scale, shift, relu, st);
    return pf ? launch<6, true>(A, W, out, M, N, K, scale, shift, relu, st) : launch<6, false>(A, W, out, M, N, K, scale, shift, relu, st);
}
