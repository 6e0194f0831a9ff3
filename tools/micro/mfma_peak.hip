// Sustained matrix-pipe rate of the whole chip, no memory traffic: every wave issues independent MFMAs in a loop for a few
// milliseconds (long enough for the power management to settle).  Prints TFLOP/s for v_mfma_f32_32x32x2_f32 (nominal 157.3 at
// 2.4 GHz) and v_mfma_f32_32x32x16_bf16 (nominal 2516), with 1 / 2 / 4 waves per SIMD.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_peak.hip -o tools/micro/mfma_peak && tools/micro/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int KIND>
__global__ __launch_bounds__(256) void k_peak(float* out, int iters, float seed) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    // per-lane pseudo-random operands, a different register every MFMA (realistic operand toggling: constant operands would
    // understate the power the matrix pipe draws)
    unsigned rng = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + (unsigned)seed;
    float fa[8], fb[8];
    bf16x8 va[8], vb[8];
    for (int k = 0; k < 8; ++k) {
        rng = rng * 1664525u + 1013904223u; fa[k] = ((rng >> 8) * (1.0f / 8388608.0f) - 1.0f) * 1e-3f;
        rng = rng * 1664525u + 1013904223u; fb[k] = ((rng >> 8) * (1.0f / 8388608.0f) - 1.0f) * 1e-3f;
        for (int e = 0; e < 8; ++e) {
            rng = rng * 1664525u + 1013904223u; va[k][e] = (__bf16)(((rng >> 8) * (1.0f / 8388608.0f) - 1.0f) * 1e-3f);
            rng = rng * 1664525u + 1013904223u; vb[k][e] = (__bf16)(((rng >> 8) * (1.0f / 8388608.0f) - 1.0f) * 1e-3f);
        }
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (KIND == 0) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[(u * 4 + i) & 7], fb[(u * 5 + i * 3) & 7], acc[i], 0, 0, 0);
                else acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va[(u * 4 + i) & 7], vb[(u * 5 + i * 3) & 7], acc[i], 0, 0, 0);
            }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 12345.678f) out[0] = s;      // keep the loop
}

template <int KIND>
void run(const char* name, double flop_per_mfma, int waves_per_simd) {
    float* out;
    hipMalloc(&out, 4);
    const int wgs = 256 * waves_per_simd;          // 256 threads = 4 waves = one per SIMD per workgroup
    const int iters = KIND == 0 ? 40000 : 80000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        for (int l = 0; l < 4; ++l) hipLaunchKernelGGL(k_peak<KIND>, dim3(wgs), dim3(256), 0, 0, out, iters, 1.0f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double mfmas = 4.0 * (double)wgs * 4 * iters * 16;
        if (rep == 2) printf("%s, %d wave(s) per SIMD: %.1f ms, %.1f TFLOP/s\n", name, waves_per_simd, ms, mfmas * flop_per_mfma / (ms * 1e-3) / 1e12);
    }
    hipFree(out);
}

int main() {
    for (int w : {1, 2, 4}) run<0>("v_mfma_f32_32x32x2_f32  ", 2.0 * 32 * 32 * 2, w);
    for (int w : {1, 2, 4}) run<1>("v_mfma_f32_32x32x16_bf16", 2.0 * 32 * 32 * 16, w);
    return 0;
}
