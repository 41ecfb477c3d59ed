// Shared device helpers for the gfx950 (MI355X, CDNA4) kernels.  wave = 64 lanes, hard-coded.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/fvk_amd.h"

typedef __bf16 bf16_t;
typedef bf16_t bf16x8 __attribute__((ext_vector_type(8)));
typedef bf16_t bf16x4 __attribute__((ext_vector_type(4)));
typedef bf16_t bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define FVK_WAVE 64

// error plumbing (host) ---------------------------------------------------------------------
void fvk_set_error(const char* fmt, ...);
#define FVK_CHECK(cond, code, ...)        \
    do {                                  \
        if (!(cond)) {                    \
            fvk_set_error(__VA_ARGS__);   \
            return (code);                \
        }                                 \
    } while (0)
#define FVK_LAUNCH_CHECK()                                                     \
    do {                                                                       \
        hipError_t e__ = hipGetLastError();                                    \
        if (e__ != hipSuccess) {                                               \
            fvk_set_error("%s:%d launch failed: %s", __FILE__, __LINE__,       \
                          hipGetErrorString(e__));                             \
            return FVK_ERR_LAUNCH;                                             \
        }                                                                      \
    } while (0)

// hipFuncAttributeMaxDynamicSharedMemorySize is a property of (function, device): a process that drives several GPUs must set it
// on each.  `flags` = one function-local static array per kernel instantiation; returns true when the caller still has to set the
// attribute on the CURRENT device.
#define FVK_MAX_DEVICES 64
inline bool fvk_needs_lds_config(bool* flags) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= FVK_MAX_DEVICES) return true;  // unknown device: always (re)configure
    if (flags[dev]) return false;
    flags[dev] = true;
    return true;
}

// device helpers ----------------------------------------------------------------------------
__device__ __forceinline__ bf16x8 ld_bf16x8(const void* p) {
    uint4 u = *reinterpret_cast<const uint4*>(p);
    return *reinterpret_cast<bf16x8*>(&u);
}
__device__ __forceinline__ void st_bf16x8(void* p, bf16x8 v) {
    *reinterpret_cast<uint4*>(p) = *reinterpret_cast<uint4*>(&v);
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float bf16_round(float f) { return (float)(bf16_t)f; }

// tanh-approximate GELU in fp32, same formula as at::gelu(approximate="tanh"):
// 0.5*x*(1+tanh(sqrt(2/pi)*(x+0.044715*x^3)))
__device__ __forceinline__ float gelu_tanh_f32(float x) {
    const float kBeta = 0.7978845608028654f;  // sqrt(2/pi)
    const float kKappa = 0.044715f;
    float inner = kBeta * (x + kKappa * x * x * x);
    return 0.5f * x * (1.0f + tanhf(inner));
}
__device__ __forceinline__ float silu_f32(float x) { return x / (1.0f + __expf(-x)); }
