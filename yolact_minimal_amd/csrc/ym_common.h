// Shared helpers for the gfx950 kernels of libyolact_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "yolact_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

void ym_set_error(const char* fmt, ...);

#define YM_REQUIRE(cond, ...)                \
    do {                                     \
        if (!(cond)) {                       \
            ym_set_error(__VA_ARGS__);       \
            return YM_EINVAL;                \
        }                                    \
    } while (0)

static inline int ym_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        ym_set_error("%s: %s", what, hipGetErrorString(e));
        return YM_ELAUNCH;
    }
    return YM_OK;
}

static inline int ym_cdiv(int a, int b) { return (a + b - 1) / b; }

// hipFuncAttributeMaxDynamicSharedMemorySize belongs to ONE DEVICE's copy of a kernel: a process that drives a second GPU (nms_batch
// and the request pipeline accept tensors on a non-current device) must raise it there as well.  One slot per (call site, device),
// raised when a launch asks for more than was granted; a refusal is reported instead of surfacing as an opaque launch failure.
struct YmLdsAttr { size_t granted[32]; };
static inline int ym_ensure_dyn_lds(YmLdsAttr& a, const void* fn, size_t bytes, const char* what) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 32) {
        ym_set_error("%s: no current HIP device (or device index >= 32)", what);
        return YM_EINVAL;
    }
    if (bytes > a.granted[dev]) {
        const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) {
            ym_set_error("%s: %zu bytes of dynamic LDS refused on device %d: %s", what, bytes, dev, hipGetErrorString(e));
            return YM_EINVAL;
        }
        a.granted[dev] = bytes;
    }
    return YM_OK;
}

// Bijective XCD-aware remap (cdna guide T1): hardware places block b on XCD b % 8; give each XCD a
// contiguous chunk of the logical tile space so neighbouring tiles share one L2.
__device__ __forceinline__ int ym_xcd_remap(int bid, int nwg) {
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (bid >> 3);
}

// The normalise + affine step of train-mode BN, in ONE fixed operation order (sub, mul, fused multiply-add): the forward pass and
// every backward kernel that re-derives the ReLU mask from y instead of reading `out` (train_ops.hip, the fused statistics of
// the data-gradient epilogue in conv_mfma.hip) must agree on the sign bit for bit.
__device__ __forceinline__ float bn_affine(float v, float mu, float is, float g, float b) {
    return __builtin_fmaf((v - mu) * is, g, b);
}

__device__ __forceinline__ float ym_apply_act(float v, int act) {
    if (act == YM_ACT_RELU) return v < 0.f ? 0.f : v;   // NaN stays NaN, like torch.relu
    if (act == YM_ACT_TANH) return tanhf(v);
    if (act == YM_ACT_GELU) return 0.5f * v * (1.f + erff(v * 0.70710678118654752440f));
    return v;
}
