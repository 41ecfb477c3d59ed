// The matrix pipe's SUSTAINED bf16 rate at the socket power cap, measured by the library that reports against it (bench.py:
// roofline.sustained_matrix_rate_at_cap_tf; stand-alone twin with more shapes / operand data: scripts/probes/mfma_energy.cpp).
// A registers-only stream of v_mfma_f32_16x16x32_bf16 — the instruction of attn_w16 / gemm_w1 / vae_conv3w — one wave per SIMD, 64 independent
// accumulator tiles, no LDS, no memory traffic in the loop.  The rate depends on the operand DATA (toggling bits cost energy: 2418 TF on zeros,
// 2049-2078 TF on random mantissas at 1400 W, profiles/r05b_attn_energy_ab.log), so the operands are filled with normal-like values from a
// per-lane xorshift stream (data = 1; data = 0: zeros, the cycle-bound rate).  Measurement only: nothing on the product path calls it.
#include "fvk_common.h"

namespace {

__device__ __forceinline__ unsigned xs32(unsigned& s) { s ^= s << 13; s ^= s >> 17; s ^= s << 5; return s; }
__device__ __forceinline__ float uni(unsigned& s) { return (float)(xs32(s) >> 8) * (2.0f / 16777216.0f) - 1.0f; }

__global__ __launch_bounds__(256, 1) void mfma_sustained_probe_kernel(float* out, int iters, int data) {
#if defined(__HIP_DEVICE_COMPILE__)
    FVK_CLAIM_WHOLE_REGISTER_FILE();
    bf16x8 a[8], b[8];
    unsigned s = 0x9E3779B9u * (blockIdx.x * 256 + threadIdx.x + 1);
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float va = data ? 0.866f * (uni(s) + uni(s) + uni(s) + uni(s)) : 0.f;   // sum of four uniforms: variance 1
            const float vb = data ? 0.866f * (uni(s) + uni(s) + uni(s) + uni(s)) : 0.f;
            a[i][e] = (bf16_t)va;
            b[i][e] = (bf16_t)vb;
        }
    f32x4 acc[64];
#pragma unroll
    for (int i = 0; i < 64; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 64; ++i)
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a[(i >> 2) & 7]), "v"(b[i & 7]));
    }
    asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");  // the last MFMAs' results before the compiler's v_accvgpr_read
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 64; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) t += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = t;
#endif
}

}  // namespace

// out: >= workgroups * 256 floats of scratch (written, never read back by the caller); FLOP per launch = workgroups * 4 waves * iters * 64 MFMAs
// * 16 384 (2 * 16 * 16 * 32).  workgroups = 256 fills the chip with one wave per SIMD.
extern "C" int fvk_mfma_sustained_probe_bf16(float* out, int workgroups, int iters, int data, void* stream) {
    FVK_CHECK(out && workgroups > 0 && workgroups <= 65536 && iters > 0 && (data == 0 || data == 1), FVK_ERR_ARG,
              "fvk_mfma_sustained_probe_bf16: bad arguments (workgroups=%d iters=%d data=%d)", workgroups, iters, data);
    hipLaunchKernelGGL(mfma_sustained_probe_kernel, dim3((unsigned)workgroups), dim3(256), 0, (hipStream_t)stream, out, iters, data);
    FVK_LAUNCH_CHECK();
    return FVK_OK;
}
