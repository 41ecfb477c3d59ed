// Shared device helpers for the gfx950 (MI355X, CDNA4) kernels.  wave = 64 lanes, hard-coded.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

#include "../../include/fvk_amd.h"

typedef __bf16 bf16_t;
typedef bf16_t bf16x8 __attribute__((ext_vector_type(8)));
typedef bf16_t bf16x4 __attribute__((ext_vector_type(4)));
typedef bf16_t bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define FVK_WAVE 64

// 1 only in the measurement build (scripts/probes/libfvk_probe.so, -DFVK_PROBE_BUILD): non-shipping kernel variants, ablation probes and
// the fvk_set_tunable knobs that select them are compiled in.  The product library compiles none of that (gemm_common.h: fvk::tunable).
#ifdef FVK_PROBE_BUILD
#define FVK_VARIANTS 1
#else
#define FVK_VARIANTS 0
#endif

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
// on each.  One function-local static FvkLdsConfigured per kernel instantiation; the per-device flag is committed only AFTER
// hipFuncSetAttribute succeeded (a failed call is retried — and reported — by the next launch instead of surfacing later as an
// unrelated launch error), and it is atomic, so host threads driving different GPUs do not race (setting the attribute twice
// from two threads is harmless).
#define FVK_MAX_DEVICES 64
struct FvkLdsConfigured {
    std::atomic<bool> done[FVK_MAX_DEVICES];
    FvkLdsConfigured() {
        for (auto& d : done) d.store(false, std::memory_order_relaxed);
    }
};
inline int fvk_config_lds(FvkLdsConfigured& c, const void* func, int bytes, const char* who) {
    int dev = -1;
    const bool known = hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < FVK_MAX_DEVICES;  // unknown device: always (re)configure
    if (known && c.done[dev].load(std::memory_order_acquire)) return FVK_OK;
    hipError_t e = hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) {
        fvk_set_error("%s: cannot set dynamic LDS size %d on device %d: %s", who, bytes, dev, hipGetErrorString(e));
        return FVK_ERR_LAUNCH;
    }
    if (known) c.done[dev].store(true, std::memory_order_release);
    return FVK_OK;
}

// A kernel that streams v_mfma_*_16x16x32 (or the 16x16x128 fp8 form) with ONE wave per SIMD must own that SIMD's whole register file.
// Found in round 4 (scripts/coresidency/pk_f32_mfma_probe.py, profiles/r04z_pk_f32_beside_mfma.log): while a gemm_w1 wave (408 of 512 registers) runs,
// a wave of ANOTHER kernel that fits into the 104 left over and shares the SIMD gets wrong LOW halves out of its packed-fp32 VALU instructions
// (v_pk_mul_f32 / v_pk_add_f32 with op_sel / neg modifiers: the RoPE arithmetic of rmsnorm_rope_kernel, 192 of 200 launches wrong beside a
// gemm_w1 on another stream; never beside the vendor GEMM, never with the packed instructions compiled out, never once gemm_w1 claims all 512
// registers).  Same stream = no overlap = never seen in a single-stream process; two streams, or two processes on one GPU, hit it.  The
// clobbers make the compiler report 256 arch VGPRs + 256 AGPRs, so the dispatcher places no second wave on the SIMD; it costs nothing (the
// kernels run one wave per SIMD by design).
#if defined(FVK_NO_REGISTER_CLAIM)  // scripts/probes/libfvk_bug.so only (round 5: the pre-fix state rebuilt on purpose, to study the bug)
#define FVK_CLAIM_WHOLE_REGISTER_FILE()
#else
#define FVK_CLAIM_WHOLE_REGISTER_FILE() asm volatile("" ::: "v255", "a255")
#endif

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
// The same rounding, opaque to the optimiser.  bf16(bf16(a * b) + c) written with plain casts may be narrowed to bf16 operations and then CONTRACTED
// to one fused multiply-add (a single rounding) — seen in round 6 when the VSA combine arithmetic was tried inside attn_bs16's epilogue (one-ulp
// differences against vsa_combine_kernel, commit f214e3e; profiles/r06g_vsa_step_ab_fused_combine_and_split.log).
__device__ __forceinline__ float bf16_round_opaque(float f) {
    float r = (float)(bf16_t)f;
    asm volatile("" : "+v"(r));
    return r;
}

// tanh-approximate GELU in fp32, same formula as at::gelu(approximate="tanh"):
// 0.5*x*(1+tanh(sqrt(2/pi)*(x+0.044715*x^3)))
__device__ __forceinline__ float gelu_tanh_f32(float x) {
    const float kBeta = 0.7978845608028654f;  // sqrt(2/pi)
    const float kKappa = 0.044715f;
    float inner = kBeta * (x + kKappa * x * x * x);
    return 0.5f * x * (1.0f + tanhf(inner));
}
__device__ __forceinline__ float silu_f32(float x) { return x / (1.0f + __expf(-x)); }
