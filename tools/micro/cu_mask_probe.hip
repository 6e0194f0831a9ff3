// Where do the workgroups of a launch on a CU-masked stream run (hipExtStreamCreateWithCUMask), does a hipGraph captured on / launched
// into such a stream keep the mask, and do four masked streams with disjoint quarters of the chip run side by side?
//   hipcc -O3 --offload-arch=gfx950 tools/micro/cu_mask_probe.hip -o /tmp/cu_mask_probe && /tmp/cu_mask_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ __launch_bounds__(256) void k_where(unsigned* out, int spin) {
    // HW_REG_XCC_ID (id 20) bits 3:0; HW_REG_HW_ID (id 4): CU_ID bits 11:8, SH_ID bit 12, SE_ID bits 15:13
    const unsigned xcc = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20) & 15u;
    const unsigned hw = __builtin_amdgcn_s_getreg((32 - 1) << 11 | 0 << 6 | 4);
    float a = threadIdx.x;
    for (int i = 0; i < spin; ++i) a = a * 1.0001f + 0.5f;
    if (threadIdx.x == 0) out[blockIdx.x] = xcc << 16 | ((hw >> 8) & 0xffu) | (a == 12345.f ? 1u << 31 : 0u);
}

static void histogram(const char* what, const std::vector<unsigned>& h) {
    int per_xcc[16] = {0};
    std::vector<int> seen(16 * 256, 0);
    for (unsigned v : h) { per_xcc[(v >> 16) & 15]++; seen[((v >> 16) & 15) * 256 + (v & 0xff)]++; }
    int cus = 0;
    for (int s : seen) cus += s > 0;
    printf("%-44s distinct (xcc, cu/se) slots %3d; workgroups per XCC:", what, cus);
    for (int x = 0; x < 8; ++x) printf(" %4d", per_xcc[x]);
    printf("\n");
}

int main() {
    const int WGS = 2048;
    unsigned* d;
    CK(hipMalloc(&d, WGS * 4 * 8));
    std::vector<unsigned> h(WGS);
    hipStream_t plain;
    CK(hipStreamCreate(&plain));
    hipLaunchKernelGGL(k_where, dim3(WGS), dim3(256), 0, plain, d, 2000);
    CK(hipStreamSynchronize(plain));
    CK(hipMemcpy(h.data(), d, WGS * 4, hipMemcpyDeviceToHost));
    histogram("plain stream", h);
    // masks: 256 bits; candidates for "a quarter of the chip"
    struct { const char* name; uint32_t m[8]; } masks[] = {
        {"bits 0..63", {0xffffffffu, 0xffffffffu, 0, 0, 0, 0, 0, 0}},
        {"bits 64..127", {0, 0, 0xffffffffu, 0xffffffffu, 0, 0, 0, 0}},
        {"every 4th bit (0,4,8,..)", {0x11111111u, 0x11111111u, 0x11111111u, 0x11111111u, 0x11111111u, 0x11111111u, 0x11111111u, 0x11111111u}},
        {"bits with (i % 8) < 2", {0x03030303u, 0x03030303u, 0x03030303u, 0x03030303u, 0x03030303u, 0x03030303u, 0x03030303u, 0x03030303u}},
        {"bits 0..31", {0xffffffffu, 0, 0, 0, 0, 0, 0, 0}},
    };
    hipStream_t ms[5];
    for (int i = 0; i < 5; ++i) {
        hipError_t e = hipExtStreamCreateWithCUMask(&ms[i], 8, masks[i].m);
        if (e != hipSuccess) { printf("hipExtStreamCreateWithCUMask(%s) -> %s\n", masks[i].name, hipGetErrorString(e)); return 1; }
        hipLaunchKernelGGL(k_where, dim3(WGS), dim3(256), 0, ms[i], d, 2000);
        CK(hipStreamSynchronize(ms[i]));
        CK(hipMemcpy(h.data(), d, WGS * 4, hipMemcpyDeviceToHost));
        char buf[96];
        snprintf(buf, sizeof buf, "masked stream, %s", masks[i].name);
        histogram(buf, h);
    }
    // a graph captured on the masked stream 0, replayed into it and into the plain stream
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(ms[0], hipStreamCaptureModeThreadLocal));
    hipLaunchKernelGGL(k_where, dim3(WGS), dim3(256), 0, ms[0], d, 2000);
    CK(hipStreamEndCapture(ms[0], &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, ms[0])); CK(hipStreamSynchronize(ms[0]));
    CK(hipMemcpy(h.data(), d, WGS * 4, hipMemcpyDeviceToHost));
    histogram("graph (captured on mask 0) -> masked stream 0", h);
    CK(hipGraphLaunch(ge, plain)); CK(hipStreamSynchronize(plain));
    CK(hipMemcpy(h.data(), d, WGS * 4, hipMemcpyDeviceToHost));
    histogram("graph (captured on mask 0) -> plain stream", h);
    // concurrency: the same long kernel on one masked stream vs on two disjoint masked streams at once
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, plain));
        CK(hipStreamWaitEvent(ms[0], e0, 0)); CK(hipStreamWaitEvent(ms[1], e0, 0));
        hipLaunchKernelGGL(k_where, dim3(WGS), dim3(256), 0, ms[0], d, 200000);
        if (rep == 1) hipLaunchKernelGGL(k_where, dim3(WGS), dim3(256), 0, ms[1], d + WGS, 200000);
        hipEvent_t j0, j1; CK(hipEventCreate(&j0)); CK(hipEventCreate(&j1));
        CK(hipEventRecord(j0, ms[0])); CK(hipEventRecord(j1, ms[1]));
        CK(hipStreamWaitEvent(plain, j0, 0)); CK(hipStreamWaitEvent(plain, j1, 0));
        CK(hipEventRecord(e1, plain));
        CK(hipEventSynchronize(e1));
        float ms_ = 0; CK(hipEventElapsedTime(&ms_, e0, e1));
        printf("%s: %.3f ms\n", rep == 0 ? "one masked stream (bits 0..63), long kernel" : "two disjoint masked streams, the same kernel each", ms_);
    }
    hipLaunchKernelGGL(k_where, dim3(WGS), dim3(256), 0, plain, d, 200000);
    CK(hipEventRecord(e0, plain));
    hipLaunchKernelGGL(k_where, dim3(WGS), dim3(256), 0, plain, d, 200000);
    CK(hipEventRecord(e1, plain)); CK(hipEventSynchronize(e1));
    float t = 0; CK(hipEventElapsedTime(&t, e0, e1));
    printf("plain stream, the same kernel: %.3f ms\n", t);
    return 0;
}
