// Argument block shared by the two bf16 GEMM kernels (gemm_bf16.hip: 128x128 register-staged tile for small / odd
// shapes; gemm_ph.hip: 256x256 LDS-DMA tile with a 128-B-row K-step for the token-axis GEMMs; gemm_pp.hip: its 64-B-row predecessor, kept for
// K % 64 != 0 and as the fp8 kernel) and the tunables registry.
#pragma once
#include "fvk_common.h"

namespace fvk {

struct GemmArgs {
    const bf16_t* x;
    const bf16_t* w;
    const bf16_t* bias;
    bf16_t* out;
    const bf16_t* residual;
    const float* gate;
    int M, N, K;
    long lda, ldc;
    int rows_per_batch;
    int ntm, ntn;
    float epi_scalar;
    long x_bstride, w_bstride, out_bstride;
    // fp8 path (gemm_pp_fp8_launch): x / w point at e4m3 bytes; dequantisation scales, one value or one per row / output channel
    const float* scale_a;
    const float* scale_b;
    int scale_a_rowwise, scale_b_rowwise;
    // V^T form (fvk_gemm_vt_bf16, gemm_w1.hip only, epilogue FVK_EPI_VT): the w operand's rows (tokens) are staged with bits 2 and 3 of their
    // row index swapped — output column p then holds token swap(p), the key order of the attention kernels' V^T layout — and columns are
    // stored up to n_store (>= N: the zero padding of V^T)
    int w_row_perm, n_store;
};

constexpr int FVK_EPI_VT = 5;  // internal: y = bf16(acc + bias[m]) for columns whose token exists, 0 for padding columns (see w_row_perm)

// gemm_pp.hip
bool gemm_pp_eligible(const GemmArgs& a);
int gemm_pp_launch(GemmArgs a, int epilogue, int batch, hipStream_t s);
int gemm_pp_fp8_launch(GemmArgs a, int epilogue, hipStream_t s);

// gemm_ph.hip (K-step 64, unit-phased staging; gemm_impl 4)
bool gemm_ph_eligible(const GemmArgs& a);
int gemm_ph_launch(GemmArgs a, int epilogue, int batch, hipStream_t s);

// gemm_w1.hip (gemm_ph's tile and unit FIFO with four 128 x 128 waves, one per SIMD; gemm_impl 5)
bool gemm_w1_eligible(const GemmArgs& a);
int gemm_w1_launch(GemmArgs a, int epilogue, int batch, hipStream_t s);
int gemm_w1_vt_launch(GemmArgs a, int batch, hipStream_t s);  // epilogue FVK_EPI_VT (V^T form: a.w_row_perm, a.n_store)
// gemm_w1n.hip (round 4): the same arithmetic (byte-identical results) on a 256 x 128 workgroup tile, for problems whose 256 x 256 grid would
// leave half the chip idle (per-rank shapes under sequence parallelism); gemm_impl 6 forces it, 14 forbids it
bool gemm_w1n_eligible(const GemmArgs& a, int epilogue);
int gemm_w1n_launch(GemmArgs a, int epilogue, int batch, hipStream_t s);
bool gemm_w1_fp8_eligible(const GemmArgs& a);
int gemm_w1_fp8_launch(GemmArgs a, int epilogue, hipStream_t s);  // fvk_gemm_fp8 on the same kernel (16x16x128 MX-fp8 MFMAs)

// Integer knobs for within-process A/B measurements (fvk_set_tunable).  They exist only in the MEASUREMENT build of the library
// (scripts/probes/libfvk_probe.so, compiled with -DFVK_PROBE_BUILD by fastvideo_amd/_build.py: build_probe): there FVK_VARIANTS is 1,
// the non-shipping kernels and schedules (attn_pp.hip, attn_vsa.hip, the attn_pp2 / gemm_ph / vae_conv3 variants, ablation probes) are
// compiled in and `tunable()` reads the knob.  In the product library libfvk_amd.so FVK_VARIANTS is 0: every knob is the constant 0,
// the variant dispatch is not compiled, and fvk_set_tunable accepts only the value 0.
enum Tunable { TUNE_GEMM_IMPL = 0, TUNE_ATTN_IMPL = 1, TUNE_VAE_CONV_IMPL = 2, TUNE_VSA_IMPL = 3, TUNE_COUNT = 8 };
#if FVK_VARIANTS  // fvk_common.h
int tunable(int id);
#else
constexpr int tunable(int) { return 0; }
#endif

}  // namespace fvk
