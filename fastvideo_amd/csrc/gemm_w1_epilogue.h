// Direct (register) epilogue of the one-wave-per-SIMD bf16 / fp8 GEMM kernels on 16x16x32 accumulators: gemm_w1.hip (wave tile 128 x 128: NP = 4
// column groups of 32) and gemm_w1n.hip (wave tile 128 x 64: NP = 2).  Same rounding points as every GEMM epilogue of the library.
#pragma once
#include "gemm_common.h"
#include "gemm_epilogue.h"
#ifndef W1_ABL
#define W1_ABL 0  // measurement builds (FVK_EXTRA_FLAGS=-DW1_ABL=n): 1 = epilogue without stores, 2 = no wait for the stores after the epilogue (UNSAFE)
#endif

namespace fvk {

// r + y * g on two values with the product rounded before the sum (the reference's two eager ops: no fused multiply-add)
__device__ __forceinline__ fvk::f32x2_t w1_mul_add2(fvk::f32x2_t r, fvk::f32x2_t y, fvk::f32x2_t g) {
#pragma clang fp contract(off)
    const fvk::f32x2_t p = y * g;
    return r + p;
}

// Direct epilogue (VAR bit 3, 16x16x32 accumulators): no LDS bounce.  The w rows of a 32-row group are fed to the two MFMA tiles of the group
// in the order that leaves lane (l15, g) with EIGHT consecutive output columns — tile 2P row 4g + e = column 32P + 8g + e, tile 2P + 1 the
// columns + 4 — so a lane stores 16 B per (16-row m block, 32-column group): 64 contiguous bytes per output row and instruction.
// Rounding points as everywhere: y = bf16(acc + bias), the epilogue on float(y), one more rounding.
template <int EPI, bool FP8 = false, bool NT = false, int NP = 4>
__device__ __forceinline__ void w1_direct_epilogue(const fvk::GemmArgs& a, f32x4 (&acc)[2 * NP][8], int m0, int n0, int wm, int wn, int lane) {
    const int l15 = lane & 15, g = lane >> 4;
    const int ncol = n0 + wn * (32 * NP) + 8 * g;  // + 32 P
    const int mrow = m0 + wm * 128 + l15;    // + 16 mb
    constexpr bool RG = EPI == FVK_EPI_RESIDUAL_GATE;
    // V^T form (fvk_gemm_vt_bf16): the bias belongs to the output ROW (a V channel), output column p holds token swap23(p) (GemmArgs::w_row_perm)
    // and the columns whose token does not exist (p's token >= N, up to n_store) are the zero padding of V^T
    constexpr bool VT = EPI == fvk::FVK_EPI_VT;
    const int n_lim = VT ? a.n_store : a.N;
    // gate rows (one per rows_per_batch output rows; rows_per_batch >= 128 here — gemm_w1_launch sends finer gates to the LDS-bounce variant):
    // the wave's 128 rows see at most TWO of them, loaded once per column group; row m takes the second one from `bnd` on
    const int mf = m0 + wm * 128;
    // (a wave tile that starts at or past row M stores nothing, but its gate loads below must stay inside the [M / rows_per_batch, N] gate
    // tensor: round 6's guard-page runs caught the last layer's FFN-out reading one gate row past the end — profiles/r06d_guard_page_runs.md)
    const int gb0 = RG ? (mf < a.M ? mf : a.M - 1) / a.rows_per_batch : 0;
    const int bnd = (gb0 + 1) * a.rows_per_batch;
    const bool two_gates = RG && a.gate && bnd < mf + 128 && bnd < a.M;
    // One 32-column group at a time, its eight m blocks inside: the per-column state (bias, fp8 weight scales, gate) and the group's eight
    // residual vectors are all that is live beside the accumulators and the next tile's first fragments (which stay in registers across the
    // epilogue) — no spills.
    bf16x8 resv[RG ? 2 : 1][RG ? 8 : 1];  // the group's eight residual vectors, requested one group ahead
    auto load_res = [&](int P) {
        const int n = ncol + 32 * P, nc = n < a.N ? n : 0;
#pragma unroll
        for (int mb = 0; mb < (RG ? 8 : 0); ++mb) {
            int m = mrow + 16 * mb;
            m = m < a.M ? m : a.M - 1;  // clamped addresses: always inside the operand, masked at the store
            resv[P & 1][mb] = ld_bf16x8(a.residual + (long)m * a.ldc + nc);
        }
    };
    load_res(0);
#pragma unroll
    for (int P = 0; P < NP; ++P) {
        const int n = ncol + 32 * P;
        const int nc = n < a.N ? n : 0;
        float b8[8], sb8[FP8 ? 8 : 1], gt0[RG ? 8 : 1], gt1[RG ? 8 : 1];
        if (RG && P < NP - 1) load_res(P + 1);
        __builtin_amdgcn_sched_barrier(0);  // (no further hoisting: two groups of residual vectors in flight, not four)
        if (!VT && a.bias && n < a.N) {
            const bf16x8 bv = ld_bf16x8(a.bias + n);
#pragma unroll
            for (int e = 0; e < 8; ++e) b8[e] = (float)bv[e];
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) b8[e] = 0.f;
        }
        if constexpr (FP8) {
#pragma unroll
            for (int e = 0; e < 8; ++e) sb8[e] = a.scale_b_rowwise ? (n + e < a.N ? a.scale_b[n + e] : 0.f) : a.scale_b[0];
        }
        if constexpr (RG) {
            if (a.gate) {
                const float* gp = a.gate + (long)gb0 * a.N + nc;
                const f32x4 g0 = *reinterpret_cast<const f32x4*>(gp), g1 = *reinterpret_cast<const f32x4*>(gp + 4);
                const float* gq = two_gates ? gp + a.N : gp;
                const f32x4 h0 = *reinterpret_cast<const f32x4*>(gq), h1 = *reinterpret_cast<const f32x4*>(gq + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { gt0[e] = g0[e]; gt0[4 + e] = g1[e]; gt1[e] = h0[e]; gt1[4 + e] = h1[e]; }
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) gt0[e] = gt1[e] = 1.0f;
            }
        }
#pragma unroll
        for (int mb = 0; mb < 8; ++mb) {
            const int m = mrow + 16 * mb;
            bf16x8 y;
            if (FP8) {
                // ref: torch._scaled_mm(x_fp8, w_fp8.t(), scale_a, scale_b, out_dtype=bf16) then `out + bias` in bf16
                // (fastvideo/layers/quantization/fp8_config.py:141-152): two roundings
                const float sa = a.scale_a_rowwise ? (m < a.M ? a.scale_a[m] : 0.f) : a.scale_a[0];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float y0 = (float)(bf16_t)((e < 4 ? acc[2 * P][mb][e] : acc[2 * P + 1][mb][e - 4]) * (sa * sb8[e]));
                    y[e] = a.bias ? (bf16_t)(y0 + b8[e]) : (bf16_t)y0;
                }
            } else if constexpr (VT) {
                const float br = (a.bias && m < a.M) ? (float)a.bias[m] : 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int tok = (n & ~8) | (e & 3) | ((e & 4) << 1) | ((n & 8) >> 1);  // swap23(n + e): n is a multiple of 8
                    const float v = (e < 4 ? acc[2 * P][mb][e] : acc[2 * P + 1][mb][e - 4]) + br;
                    y[e] = tok < a.N ? (bf16_t)v : (bf16_t)0.f;
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; e += 2) {  // (pairs: v_pk_add_f32 — the same sums)
                    const fvk::f32x2_t lo = fvk::f32x2_t{acc[2 * P][mb][e], acc[2 * P][mb][e + 1]} + fvk::f32x2_t{b8[e], b8[e + 1]};
                    const fvk::f32x2_t hi = fvk::f32x2_t{acc[2 * P + 1][mb][e], acc[2 * P + 1][mb][e + 1]} + fvk::f32x2_t{b8[4 + e], b8[5 + e]};
                    y[e] = (bf16_t)lo[0]; y[e + 1] = (bf16_t)lo[1];
                    y[4 + e] = (bf16_t)hi[0]; y[5 + e] = (bf16_t)hi[1];
                }
            }
            if (m < a.M && n < n_lim) {
                if (EPI == FVK_EPI_GELU_TANH) {
#pragma unroll
                    for (int e = 0; e < 8; e += 2) {  // pairs: packed fp32 arithmetic (this epilogue runs with the matrix pipe idle)
                        const fvk::f32x2_t g2 = fvk::gelu_tanh_fast2(fvk::f32x2_t{(float)y[e], (float)y[e + 1]});
                        y[e] = (bf16_t)g2[0];
                        y[e + 1] = (bf16_t)g2[1];
                    }
                } else if (EPI == FVK_EPI_SILU) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) y[e] = (bf16_t)silu_f32((float)y[e]);
                } else if (EPI == FVK_EPI_DIV) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) y[e] = (bf16_t)__fdiv_rn((float)y[e], a.epi_scalar);
                } else if constexpr (RG) {
                    const bool second = m >= bnd;
#pragma unroll
                    for (int e = 0; e < 8; e += 2) {  // (pairs; multiply and add stay two roundings: w1_mul_add2)
                        const fvk::f32x2_t o2 = w1_mul_add2(fvk::f32x2_t{(float)resv[P & 1][mb][e], (float)resv[P & 1][mb][e + 1]}, fvk::f32x2_t{(float)y[e], (float)y[e + 1]},
                                                            fvk::f32x2_t{second ? gt1[e] : gt0[e], second ? gt1[e + 1] : gt0[e + 1]});
                        y[e] = (bf16_t)o2[0]; y[e + 1] = (bf16_t)o2[1];
                    }
                }
#if W1_ABL != 1  // (timing ablation 1: the epilogue without its stores)
                if (NT) __builtin_nontemporal_store(y, reinterpret_cast<bf16x8*>(a.out + (long)m * a.ldc + n));  // streaming store (VAR bit 7)
                else st_bf16x8(a.out + (long)m * a.ldc + n, y);
#else
                asm volatile("" :: "v"(y));
#endif
            }
        }
    }
}

}  // namespace fvk
