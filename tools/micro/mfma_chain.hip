// Micro-benchmark: cost of dependent vs independent v_mfma_f32_32x32x2_f32 chains for a wave that owns its SIMD, and the
// s_memtime tick rate (tools/micro is not part of the library).  Build: hipcc --offload-arch=gfx950 -O3 mfma_chain.hip -o mfma_chain
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, long long* ticks, int iters) {
    f32x16 acc[NACC];
    for (int a = 0; a < NACC; ++a)
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    float x = threadIdx.x * 1e-3f, y = 1.0f;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) acc[u % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[u % NACC], 0, 0, 0);
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int a = 0; a < NACC; ++a)
        for (int r = 0; r < 16; ++r) s += acc[a][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

template <int NACC>
void run(int blocks, int iters) {
    float* out; long long* ticks;
    hipMalloc(&out, blocks * 256 * 4); hipMalloc(&ticks, blocks * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(256), 0, 0, out, ticks, iters);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(256), 0, 0, out, ticks, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h[4]; hipMemcpy(h, ticks, 32, hipMemcpyDeviceToHost);
    const double mf = 16.0 * iters;
    printf("blocks %4d accs %d: %.1f ticks/MFMA, %.2f ns/MFMA (wall) -> %.2f GHz if 64 cycles/MFMA; tick rate %.3f GHz\n", blocks, NACC,
           h[0] / mf, ms * 1e6 / mf, 64.0 / (ms * 1e6 / mf), h[0] / (ms * 1e6));
    hipFree(out); hipFree(ticks);
}

int main() {
    for (int blocks : {1, 256, 512, 1024}) {
        run<1>(blocks, 20000);
        run<2>(blocks, 20000);
        run<4>(blocks, 20000);
    }
    return 0;
}
