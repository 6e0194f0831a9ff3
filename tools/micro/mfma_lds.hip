// Steady-state rate of the conv kernels' inner pattern, in isolation: per "K tile" a wave does 4 x [2 ds_read_b128 (conflict-free,
// XOR-swizzled rows as in conv_igemm_pers), s_waitcnt lgkmcnt(0), 4 v_mfma_f32_32x32x2_f32 on two alternating accumulators], with
// or without an s_barrier per K tile, no global traffic.  W waves per SIMD (W workgroups of 4 waves per CU).
//   hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_lds.hip -o tools/micro/mfma_lds
#include <hip/hip_runtime.h>
#include <stdio.h>
// -DFASTADDR: wrap the stream offsets with one v_and instead of an integer modulo (~12 VALU instructions each): the modulo alone
// made the "operand stream" cost 48 VALU instructions per lane and K tile, and VALU issue displaces MFMA issue
#ifdef FASTADDR
#define WRAP & ~-(int)
#else
#define WRAP %
#endif
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define RD(fa_, fb_, base, g) fa_ = *reinterpret_cast<const f32x4*>(base + a_off + goff[g]); fb_ = *reinterpret_cast<const f32x4*>(base + b_off + goff[g]);
#define MM(fa_, fb_) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa_[0], fb_[0], acc, 0, 0, 0); acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa_[1], fb_[1], acc2, 0, 0, 0); \
acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa_[2], fb_[2], acc, 0, 0, 0); acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa_[3], fb_[3], acc2, 0, 0, 0);


// 4: mode 2 + the operand stream: 4 x buffer_load_dwordx4 ... lds per lane and K tile into the other stage (ring of 2, vmcnt(0) at the
//    barrier), from an L2-resident buffer.  5: the same with the loads going to VGPRs and 4 ds_write_b128 (register staging).
// 6: mode 4 with a ring of 3 (the tile issued ONE iteration earlier is awaited: vmcnt(4)).
template <int MODE>   // 0: MFMA only (2 accumulators)  1: + LDS reads  2: + barrier per tile  3: LDS reads software-pipelined + barrier
__global__ __launch_bounds__(256) void k(float* out, int tiles, const float* src, unsigned span) {
    extern __shared__ __attribute__((aligned(16))) float smem[];       // 2 (3) stages x (64 + 64) rows x 32 floats
    typedef __attribute__((address_space(3))) void* lds_ptr;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, span, 0x00020000);
    unsigned goffb = (unsigned)((blockIdx.x * 16384u + threadIdx.x * 16u) WRAP span);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    for (int i = tid; i < 2 * 128 * 32; i += 256) smem[i] = (float)((i * 2654435761u) >> 20) * 1e-6f;
    __syncthreads();
    const int r = lane & 31, kh = lane >> 5;
    const int a_off = (wm * 32 + r) * 32, b_off = 64 * 32 + (wn * 32 + r) * 32;
    int goff[4];
    for (int g = 0; g < 4; ++g) goff[g] = ((2 * g + kh) ^ ((r >> 1) & 7)) * 4;
    f32x16 acc, acc2;
    for (int i = 0; i < 16; ++i) { acc[i] = 0.f; acc2[i] = 0.f; }
    f32x4 fa = {1.f, 2.f, 3.f, 4.f}, fb = {.5f, .25f, .125f, 1.f};
    int buf = 0;
    if (MODE == 3) {
        f32x4 fa0, fb0, fa1, fb1;
        fa0 = *reinterpret_cast<const f32x4*>(smem + a_off + goff[0]);
        fb0 = *reinterpret_cast<const f32x4*>(smem + b_off + goff[0]);
        for (int t = 0; t < tiles; ++t) {
            const float* s = smem + buf * 128 * 32;
            const float* s1 = smem + (buf ^ 1) * 128 * 32;
            RD(fa1, fb1, s, 1) __builtin_amdgcn_sched_barrier(0); MM(fa0, fb0) __builtin_amdgcn_sched_barrier(0);
            RD(fa0, fb0, s, 2) __builtin_amdgcn_sched_barrier(0); MM(fa1, fb1) __builtin_amdgcn_sched_barrier(0);
            RD(fa1, fb1, s, 3) __builtin_amdgcn_sched_barrier(0); MM(fa0, fb0) __builtin_amdgcn_sched_barrier(0);
            RD(fa0, fb0, s1, 0) __builtin_amdgcn_sched_barrier(0); MM(fa1, fb1)
            __builtin_amdgcn_s_waitcnt(0xC07F);
            __builtin_amdgcn_s_barrier();
            buf ^= 1;
        }
    } else if (MODE == 11) {
        int nb = 1;
        f32x4 b0[4], b1[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) b0[g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)((goffb + g * 4096u) WRAP span), 0, 0));
        for (int t = 0; t < tiles; t += 2) {
#define TILE11(bcur, bnext)                                                                                                      \
            {                                                                                                                    \
                const float* s = smem + buf * 128 * 32;                                                                          \
                float* dst = smem + nb * 128 * 32 + (tid >> 6) * 8 * 32;                                                         \
                goffb = (goffb + 16384u) WRAP span;                                                                                 \
                _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                                    \
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(dst + i * 32 * 32), 16, (int)((goffb + i * 4096u) WRAP span), 0, 0, 0); \
                _Pragma("unroll") for (int g = 0; g < 4; ++g)                                                                    \
                    bnext[g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)((goffb + 8192u + g * 4096u) WRAP span), 0, 0)); \
                _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                                                  \
                    fa = *reinterpret_cast<const f32x4*>(s + a_off + goff[g]);                                                   \
                    MM(fa, bcur[g])                                                                                              \
                }                                                                                                                \
                __builtin_amdgcn_s_waitcnt(0x0070);                                                                              \
                __builtin_amdgcn_s_barrier();                                                                                    \
                nb = buf; buf ^= 1;                                                                                              \
            }
            TILE11(b0, b1)
            TILE11(b1, b0)
        }
        fa[0] += b0[0][0] * 1e-30f;
    } else if (MODE == 12 || MODE == 13) {
        // ring of 2 (12) / 3 (13) with the LDS-DMA stream, fragments software-pipelined WITHIN the tile (group g+1 is read before the
        // MFMAs of group g; group 0 right after the barrier) -- 13 also reads the next tile's group 0 before the barrier
        constexpr int NSTG = MODE == 13 ? 3 : 2;
        int nb = NSTG - 1;
        f32x4 fa0, fb0, fa1, fb1;
        RD(fa0, fb0, smem, 0)
        for (int t = 0; t < tiles; ++t) {
            const float* s = smem + buf * 128 * 32;
            const int buf1 = buf == NSTG - 1 ? 0 : buf + 1;
            const float* s1 = smem + buf1 * 128 * 32;
            float* dst = smem + nb * 128 * 32 + (tid >> 6) * 8 * 32;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(dst + i * 32 * 32), 16, (int)((goffb + (unsigned)i * 4096u) WRAP span), 0, 0, 0);
            goffb = (goffb + 16384u) WRAP span;
            RD(fa1, fb1, s, 1) __builtin_amdgcn_sched_barrier(0); MM(fa0, fb0) __builtin_amdgcn_sched_barrier(0);
            RD(fa0, fb0, s, 2) __builtin_amdgcn_sched_barrier(0); MM(fa1, fb1) __builtin_amdgcn_sched_barrier(0);
            RD(fa1, fb1, s, 3) __builtin_amdgcn_sched_barrier(0); MM(fa0, fb0) __builtin_amdgcn_sched_barrier(0);
            if (MODE == 13) { RD(fa0, fb0, s1, 0) __builtin_amdgcn_sched_barrier(0); }
            MM(fa1, fb1)
            if (MODE == 13) __builtin_amdgcn_s_waitcnt(0x0070); else __builtin_amdgcn_s_waitcnt(0x0070);
            __builtin_amdgcn_s_barrier();
            if (MODE == 12) { RD(fa0, fb0, s1, 0) }
            nb = buf;
            buf = buf1;
        }
    } else if (MODE >= 4) {
        constexpr int NSTG = MODE == 6 ? 3 : 2;
        constexpr int NLD = MODE == 7 ? 2 : 4;
        int nb = NSTG - 1;
        for (int t = 0; t < tiles; ++t) {
            const float* s = smem + buf * 128 * 32;
            float* dst = smem + nb * 128 * 32 + (tid >> 6) * 8 * 32;
            f32x4 st[4];
            if (MODE == 10) {
                if ((tid >> 6) == 0) {
#pragma unroll
                    for (int i = 0; i < 16; ++i)
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(smem + nb * 128 * 32 + i * 8 * 32), 16, (int)((goffb + (unsigned)i * 1024u) WRAP span), 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int i = 0; i < NLD; ++i) {
                    const unsigned o = (goffb + (unsigned)i * 4096u) WRAP span;
                    if (MODE == 5 || MODE == 9) st[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)o, 0, 0));
                    else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(dst + i * 32 * 32), 16, (int)o, 0, 0, 0);
                }
            }
            goffb = (goffb + 16384u) WRAP span;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                fa = *reinterpret_cast<const f32x4*>(s + a_off + goff[g]);
                fb = *reinterpret_cast<const f32x4*>(s + b_off + goff[g]);
                MM(fa, fb)
            }
            if (MODE == 5) {
#pragma unroll
                for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(smem + nb * 128 * 32 + i * 32 * 32 + tid * 4) = st[i];
            }
            if (MODE == 9) {                 // keep the loads alive without spending VALU on them (an earlier `fa[0] += ...` was dead code:
#pragma unroll                               // the compiler dropped the loads and this mode measured nothing)
                for (int i = 0; i < 4; ++i) asm volatile("" ::"v"(st[i]));
            }
            if (MODE == 6) __builtin_amdgcn_s_waitcnt(0x0074);
            else if (MODE == 8) { if ((t & 7) == 7) __builtin_amdgcn_s_waitcnt(0x0070); else __builtin_amdgcn_s_waitcnt(0xC07F); }
            else __builtin_amdgcn_s_waitcnt(0x0070);
            __builtin_amdgcn_s_barrier();
            nb = buf;
            buf = buf == NSTG - 1 ? 0 : buf + 1;
        }
    } else {
        for (int t = 0; t < tiles; ++t) {
            const float* s = smem + buf * 128 * 32;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (MODE >= 1) {
                    fa = *reinterpret_cast<const f32x4*>(s + a_off + goff[g]);
                    fb = *reinterpret_cast<const f32x4*>(s + b_off + goff[g]);
                }
                MM(fa, fb)
            }
            if (MODE >= 2) {
                __builtin_amdgcn_s_waitcnt(0xC07F);
                __builtin_amdgcn_s_barrier();
            }
            buf ^= 1;
        }
    }
    float sum = 0.f;
    for (int i = 0; i < 16; ++i) sum += acc[i] + acc2[i];
    if (sum == 1.2345f) out[0] = sum;
}

template <int MODE>
void run(const char* name, int w) {
    float* out;
    (void)hipMalloc(&out, 4);
    const int tiles = 20000, wgs = 256 * w;
    const int lds = (MODE == 6 || MODE == 13) ? 49152 : 32768;
    float* src;
    (void)hipMalloc(&src, 1 << 20);
    (void)hipMemset(src, 0, 1 << 20);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(wgs), dim3(256), lds, 0, out, tiles, src, 1u << 20);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double flop = (double)wgs * 4 * tiles * 16 * 4096.0;
    printf("%-58s %d wave(s)/SIMD: %7.1f TFLOP/s (%.0f %% of 157.3)\n", name, w, flop / (best * 1e-3) / 1e12, flop / (best * 1e-3) / 1e12 / 1.573);
    (void)hipFree(out);
}

int main(int argc, char**) {
    if (argc > 1) {          // counter runs (rocprofv3 --pmc): one launch set per interesting mode at 3 waves per SIMD
        run<3>("fragments software-pipelined + s_barrier per K tile", 3);
        run<13>("LDS-DMA ring of 3, fragments pipelined across the barrier", 3);
        run<5>("barrier + operand stream through VGPRs + ds_write", 3);
        run<9>("loads to VGPRs only (no ds_write)", 3);
        run<7>("LDS-DMA stream, HALF the bytes (2 loads per lane and tile)", 3);
        run<12>("LDS-DMA ring of 2, fragments pipelined within the tile", 3);
        run<11>("A through LDS (DMA), B fragments straight from global (coalesced)", 3);
        return 0;
    }
    for (int w : {1, 2, 3, 4}) run<0>("MFMA only, two alternating accumulators", w);
    for (int w : {1, 2, 3, 4}) run<1>("+ 2 ds_read_b128 and lgkmcnt(0) per 4 MFMAs", w);
    for (int w : {1, 2, 3, 4}) run<2>("+ s_barrier per K tile (16 MFMAs)", w);
    for (int w : {1, 2, 3, 4}) run<3>("fragments software-pipelined + s_barrier per K tile", w);
    for (int w : {1, 2, 3, 4}) run<4>("barrier + operand stream by LDS-DMA, ring of 2", w);
    for (int w : {1, 2, 3}) run<6>("barrier + operand stream by LDS-DMA, ring of 3", w);
    for (int w : {1, 2, 3, 4}) run<5>("barrier + operand stream through VGPRs + ds_write", w);
    for (int w : {1, 3}) run<7>("LDS-DMA stream, HALF the bytes (2 loads per lane and tile)", w);
    for (int w : {1, 3}) run<8>("LDS-DMA stream, vmcnt waited only every 8th tile", w);
    for (int w : {1, 3}) run<9>("loads to VGPRs only (no ds_write)", w);
    for (int w : {1, 3}) run<10>("LDS-DMA stream issued by ONE wave of the workgroup", w);
    for (int w : {1, 2, 3, 4}) run<12>("LDS-DMA ring of 2, fragments pipelined within the tile", w);
    for (int w : {1, 2, 3}) run<13>("LDS-DMA ring of 3, fragments pipelined across the barrier", w);
    for (int w : {1, 2, 3, 4}) run<11>("A through LDS (DMA), B fragments straight from global (coalesced)", w);
    return 0;
}
