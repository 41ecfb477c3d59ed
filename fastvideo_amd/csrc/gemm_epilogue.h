// Epilogue shared by the 256x256-tile GEMM kernels (gemm_pp.hip, gemm_ph.hip): 8 waves, wave tile 128(m) x 64(n), swapped operands.
//   acc[nb][mb][r] = D[n = nb*32 + (r&3) + 8(r>>2) + 4hi][m = mb*32 + l31]   (A = w rows, B = x rows of v_mfma_f32_32x32x16_bf16)
// The accumulator is packed to bf16 (+bias: first rounding), bounced through the wave's private LDS region and stored as whole
// 128-B output rows (16 B per lane) with the activation / gated residual applied on the way out.  Rounding points equal
// gemm_bf16.hip: y = bf16(acc + bias), then the epilogue on float(y), then one more rounding.
#pragma once
#include "gemm_common.h"

namespace fvk {

constexpr int EPI_PITCH = 144;             // bytes per staged output row (64 bf16 + 16 B pad)
constexpr int EPI_WAVE = 128 * EPI_PITCH;  // 18 432 B per wave
constexpr int EPI_LDS_BYTES = 8 * EPI_WAVE;  // 147 456 B

__device__ __forceinline__ float gelu_tanh_fast(float x) {
    // 0.5 x (1 + tanh(u)) == x / (1 + exp(-2u)),  u = sqrt(2/pi) (x + 0.044715 x^3); exp via v_exp_f32 (exp2).
    const float c0 = -2.0f * 0.7978845608028654f * 1.4426950408889634f;
    const float c1 = c0 * 0.044715f;
    const float t = __builtin_amdgcn_exp2f(x * __builtin_fmaf(x * x, c1, c0));
    return x * __builtin_amdgcn_rcpf(1.0f + t);
}

// residual + y * gate with the product rounded before the sum — the reference's two eager ops (`residual + x * gate`: layernorm.py:91-109).
// hipcc contracts __fmul_rn + __fadd_rn into one fused multiply-add unless contraction is switched off where they meet.
__device__ __forceinline__ float mul_then_add(float r, float y, float g) {
#pragma clang fp contract(off)
    const float p = y * g;
    return r + p;
}

// Two values at once: the multiplies / adds as packed fp32 operations (v_pk_mul_f32, v_pk_fma_f32, v_pk_add_f32 — the same IEEE results as the
// scalar forms; a gain only where no MFMA runs beside them, i.e. in an exposed epilogue), the two transcendentals per value as before.
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2_t gelu_tanh_fast2(f32x2_t x) {
    const float c0 = -2.0f * 0.7978845608028654f * 1.4426950408889634f;
    const float c1 = c0 * 0.044715f;
    const f32x2_t c0v = {c0, c0}, c1v = {c1, c1}, one = {1.0f, 1.0f};
    const f32x2_t u = x * __builtin_elementwise_fma(x * x, c1v, c0v);
    f32x2_t t;
    t[0] = __builtin_amdgcn_exp2f(u[0]);
    t[1] = __builtin_amdgcn_exp2f(u[1]);
    const f32x2_t d = one + t;
    f32x2_t r;
    r[0] = __builtin_amdgcn_rcpf(d[0]);
    r[1] = __builtin_amdgcn_rcpf(d[1]);
    return x * r;
}

// The caller guarantees that no wave still reads or DMA-writes the LDS ring (drained + barrier) before calling.
// PREF (gated-residual epilogue): all 16 residual vectors of the lane are requested BEFORE the accumulators are packed and bounced
// (one exposed HBM latency per tile instead of four), and the gate row is loaded once when the wave's 128 rows share a batch.
// MI16: the accumulator is f32x4 acc[4][8] of v_mfma_f32_16x16x32_bf16 tiles (gemm_w1.hip): acc[nb][mb][e] = D[n = nb*16 + 4(lane>>4) + e][m = mb*16 + (lane&15)].
template <int EPI, bool FP8, bool PREF = false, bool MI16 = false, typename Acc = f32x16[2][4]>
__device__ __forceinline__ void gemm_tile_epilogue(const GemmArgs& a, Acc& acc, unsigned char* smem, int wave, int lane, int m0, int n0) {
    const int l31 = lane & 31, hi = lane >> 5;
    const int wm = wave & 1, wn = wave >> 1;
    constexpr bool kPref = PREF && EPI == FVK_EPI_RESIDUAL_GATE;
    bf16x8 resv[kPref ? 16 : 1];
    float gt0[8];
    bool gate_uniform = false;
    if (kPref) {
        const int n = n0 + wn * 64 + (lane & 7) * 8;
        const int nc = n < a.N ? n : 0;  // clamped addresses: always inside the operand, masked at the store
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            int m = m0 + wm * 128 + it * 8 + (lane >> 3);
            m = m < a.M ? m : a.M - 1;
            resv[it] = ld_bf16x8(a.residual + (long)m * a.ldc + nc);
        }
        // (clamped: a wave tile that starts at or past row M stores nothing, but must not read a gate row past the end of the gate tensor)
        const int mf = (m0 + wm * 128 < a.M ? m0 + wm * 128 : a.M - 1), ml = (m0 + wm * 128 + 127 < a.M ? m0 + wm * 128 + 127 : a.M - 1);
        gate_uniform = a.gate && (mf / a.rows_per_batch == ml / a.rows_per_batch);
        if (gate_uniform) {
            const float* gp = a.gate + (long)(mf / a.rows_per_batch) * a.N + nc;
            const f32x4 g0 = *reinterpret_cast<const f32x4*>(gp), g1 = *reinterpret_cast<const f32x4*>(gp + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { gt0[e] = g0[e]; gt0[4 + e] = g1[e]; }
        }
    }
    unsigned char* st = smem + wave * EPI_WAVE;
    const int ncol0 = n0 + wn * 64;
    if constexpr (MI16) {
        static_assert(!FP8, "the 16x16x32 accumulator layout is a bf16-GEMM path");
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
            const int nl = nb * 16 + 4 * (lane >> 4);
            float b4[4] = {0.f, 0.f, 0.f, 0.f};
            if (a.bias && ncol0 + nl < a.N) {
                const bf16x4 bv = *reinterpret_cast<const bf16x4*>(a.bias + ncol0 + nl);
#pragma unroll
                for (int e = 0; e < 4; ++e) b4[e] = (float)bv[e];
            }
#pragma unroll
            for (int mb = 0; mb < 8; ++mb) {
                bf16x4 y;
#pragma unroll
                for (int e = 0; e < 4; ++e) y[e] = (bf16_t)(acc[nb][mb][e] + b4[e]);
                *reinterpret_cast<bf16x4*>(st + (mb * 16 + (lane & 15)) * EPI_PITCH + nl * 2) = y;
            }
        }
    } else
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int nl = nb * 32 + 8 * g + 4 * hi;
            float b4[4] = {0.f, 0.f, 0.f, 0.f};
            if (a.bias && ncol0 + nl < a.N) {
                const bf16x4 bv = *reinterpret_cast<const bf16x4*>(a.bias + ncol0 + nl);
#pragma unroll
                for (int e = 0; e < 4; ++e) b4[e] = (float)bv[e];
            }
            float sb4[4] = {1.f, 1.f, 1.f, 1.f};
            if (FP8) {
#pragma unroll
                for (int e = 0; e < 4; ++e) sb4[e] = a.scale_b_rowwise ? (ncol0 + nl + e < a.N ? a.scale_b[ncol0 + nl + e] : 0.f) : a.scale_b[0];
            }
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) {
                bf16x4 y;
                if (FP8) {
                    // ref: torch._scaled_mm(x_fp8, w_fp8.t(), scale_a, scale_b, out_dtype=bf16) then `out + bias` in bf16
                    // (fastvideo/layers/quantization/fp8_config.py:141-152): two roundings
                    const int mrow = m0 + wm * 128 + mb * 32 + l31;
                    const float sa = a.scale_a_rowwise ? (mrow < a.M ? a.scale_a[mrow] : 0.f) : a.scale_a[0];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float y0 = (float)(bf16_t)(acc[nb][mb][4 * g + e] * (sa * sb4[e]));
                        y[e] = a.bias ? (bf16_t)(y0 + b4[e]) : (bf16_t)y0;
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) y[e] = (bf16_t)(acc[nb][mb][4 * g + e] + b4[e]);
                }
                *reinterpret_cast<bf16x4*>(st + (mb * 32 + l31) * EPI_PITCH + nl * 2) = y;
            }
        }
    // the staging region is private to this wave: program order + the compiler's lgkmcnt wait are sufficient
    if (kPref) {
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int row = it * 8 + (lane >> 3), ch = lane & 7;
            const int m = m0 + wm * 128 + row, n = ncol0 + ch * 8;
            bf16x8 y = *reinterpret_cast<const bf16x8*>(st + row * EPI_PITCH + ch * 16);
            if (m < a.M && n < a.N) {
                float gt[8];
                if (gate_uniform) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) gt[e] = gt0[e];
                } else if (a.gate) {
                    const float* gp = a.gate + (long)(m / a.rows_per_batch) * a.N + n;
                    const f32x4 g0 = *reinterpret_cast<const f32x4*>(gp), g1 = *reinterpret_cast<const f32x4*>(gp + 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { gt[e] = g0[e]; gt[4 + e] = g1[e]; }
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) gt[e] = 1.0f;
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) y[e] = (bf16_t)mul_then_add((float)resv[it][e], (float)y[e], gt[e]);
                st_bf16x8(a.out + (long)m * a.ldc + n, y);
            }
        }
        return;
    }
#pragma unroll 4
    for (int it = 0; it < 16; ++it) {
        const int row = it * 8 + (lane >> 3), ch = lane & 7;
        const int m = m0 + wm * 128 + row, n = ncol0 + ch * 8;
        bf16x8 y = *reinterpret_cast<const bf16x8*>(st + row * EPI_PITCH + ch * 16);
        if (m < a.M && n < a.N) {
            if (EPI == FVK_EPI_GELU_TANH) {
#pragma unroll
                for (int e = 0; e < 8; ++e) y[e] = (bf16_t)gelu_tanh_fast((float)y[e]);
            } else if (EPI == FVK_EPI_SILU) {
#pragma unroll
                for (int e = 0; e < 8; ++e) y[e] = (bf16_t)silu_f32((float)y[e]);
            } else if (EPI == FVK_EPI_DIV) {
#pragma unroll
                for (int e = 0; e < 8; ++e) y[e] = (bf16_t)__fdiv_rn((float)y[e], a.epi_scalar);
            } else if (EPI == FVK_EPI_RESIDUAL_GATE) {
                const bf16x8 res = ld_bf16x8(a.residual + (long)m * a.ldc + n);
                float gt[8];
                if (a.gate) {
                    const float* gp = a.gate + (long)(m / a.rows_per_batch) * a.N + n;
                    const f32x4 g0 = *reinterpret_cast<const f32x4*>(gp), g1 = *reinterpret_cast<const f32x4*>(gp + 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { gt[e] = g0[e]; gt[4 + e] = g1[e]; }
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) gt[e] = 1.0f;
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) y[e] = (bf16_t)mul_then_add((float)res[e], (float)y[e], gt[e]);
            }
            st_bf16x8(a.out + (long)m * a.ldc + n, y);
        }
    }
}

}  // namespace fvk
