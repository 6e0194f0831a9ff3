// How fast can the CUs pull L2-resident data?  Every workgroup (256 threads) streams a small buffer (`span` bytes, shared by the
// workgroups of its XCD chunk: L2 hits after the first pass) into its CU, 16 bytes per lane per load, as
//   mode 0: buffer_load_dwordx4 ... lds   (global -> LDS DMA, the staging of conv_igemm_f32<..., DL> / conv_igemm_pers)
//   mode 1: buffer_load_dwordx4 -> VGPR   (register staging; the ds_write is left out: the load path alone)
// with `inflight` loads per lane outstanding.  Prints aggregate TB/s.  rows: every lane's 16 bytes lie `pitch` bytes after the
// previous row's (128 = a 32-float K tile row, 8 lanes per row), like the conv operand fetch.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/l2_lds_bw.hip -o tools/micro/l2_lds_bw
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int INFLIGHT>
__global__ __launch_bounds__(256) void k_bw(const float* src, unsigned span, int iters, float* sink, int same) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef __attribute__((address_space(3))) void* lds_ptr;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, span, 0x00020000);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    // workgroup b starts at its own offset unless `same` (then every workgroup of the chip reads the same bytes at the same time)
    unsigned off = (unsigned)(tid * 16) + (same ? 0u : (unsigned)((blockIdx.x * 4096u) % span));
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
        f32x4 v[INFLIGHT];
#pragma unroll
        for (int k = 0; k < INFLIGHT; ++k) {
            const unsigned o = (off + (unsigned)k * 4096u) % span;
            if (MODE == 0)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(smem + (k * 4 + wave) * 1024), 16, (int)o, 0, 0, 0);
            else
                v[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)o, 0, 0));
        }
        if (MODE == 0) {
            __builtin_amdgcn_s_waitcnt(0x0070);       // vmcnt(0)
        } else {
#pragma unroll
            for (int k = 0; k < INFLIGHT; ++k) acc += v[k];
        }
        off = (off + (unsigned)INFLIGHT * 4096u) % span;
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 1.2345f || (MODE == 0 && smem[lane] == 77 && iters < 0)) sink[0] = acc[0];
}

template <int MODE, int INFLIGHT>
void run(const float* src, unsigned span, float* sink, int wgs_per_cu, int same) {
    const int wgs = 256 * wgs_per_cu, iters = 2000 / INFLIGHT * 4;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k_bw<MODE, INFLIGHT>), dim3(wgs), dim3(256), INFLIGHT * 4096, 0, src, span, iters, sink, same);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double bytes = (double)wgs * 256 * 16 * INFLIGHT * iters;
    printf("  %s, %d load(s) in flight per lane, %d WG/CU, %s: %.2f TB/s\n", MODE == 0 ? "LDS-DMA " : "VGPR load", INFLIGHT, wgs_per_cu,
           same ? "all WGs the same bytes" : "WGs spread over the span", bytes / (best * 1e-3) / 1e12);
}

int main() {
    float *src, *sink;
    const unsigned span_max = 64u << 20;
    hipMalloc(&src, span_max); hipMalloc(&sink, 64);
    hipMemset(src, 0, span_max);
    for (unsigned span : {1u << 20, 16u << 20}) {
        printf("span %u MB (%s)\n", span >> 20, span <= (2u << 20) ? "fits every XCD's 4 MB L2" : "larger than an XCD's L2: Infinity Cache");
        for (int same : {0, 1}) {
            for (int w : {1, 2, 4}) {
                run<0, 4>(src, span, sink, w, same);
                run<1, 4>(src, span, sink, w, same);
            }
            run<0, 8>(src, span, sink, 2, same);
            run<1, 8>(src, span, sink, 2, same);
        }
    }
    return 0;
}
