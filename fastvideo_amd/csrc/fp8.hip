// OCP fp8 (e4m3fn) linear path of the DiT (BASELINE config 5), gfx950.
// ref: fastvideo/layers/quantization/fp8_config.py:55-68 (_quantize_tensorwise / _quantize_rowwise), :119-158 (FP8QuantizeMethod.apply:
//      dynamic activation quantisation -> torch._scaled_mm(x_fp8, w_fp8.t(), scale_a, scale_b, out_dtype=bf16) -> + bias),
//      :211-245 (convert_model_to_fp8: the same arithmetic on the weights, once).
//   absmax  = max |x|  over the tensor (tensorwise) or per row (rowwise = per token / per output channel)
//   scale   = max(absmax / 448, 1 / (448 * 512))                                            fp32
//   q       = e4m3fn_rne( clamp( bf16( float(x) / float(bf16(scale)) ), -448, 448 ) )       (the reference divides in bf16)
// The GEMM itself is gemm_pp.hip's kernel instantiated with FP8 = true (v_mfma_scale_f32_32x32x64_f8f6f4, unit block scales).
#include "gemm_common.h"
#include "fp8_common.h"

namespace {

using fvk::FP8_MAX;
using fvk::FP8_MIN_SCALE;
using fvk::fp8_pack8;

// rowwise: one wave per row.  tensorwise: grid-stride over rows, one atomicMax per wave on the (non-negative) float bits.
template <bool ROWWISE>
__global__ __launch_bounds__(256) void fp8_absmax_kernel(const bf16_t* __restrict__ x, float* __restrict__ out, int M, int K, long lda) {
    const int lane = threadIdx.x & 63;
    const int wave_g = (blockIdx.x * 256 + threadIdx.x) >> 6;
    const int n_waves = (gridDim.x * 256) >> 6;
    float mx = 0.f;
    for (int m = wave_g; m < M; m += n_waves) {
        float r = 0.f;
        const bf16_t* row = x + (long)m * lda;
        for (int k = lane * 8; k < K; k += 512) {
            const bf16x8 v = ld_bf16x8(row + k);
#pragma unroll
            for (int e = 0; e < 8; ++e) r = fmaxf(r, fabsf((float)v[e]));
        }
        if (ROWWISE) {
            r = wave_max(r);
            if (lane == 0) out[m] = r;
        } else {
            mx = fmaxf(mx, r);
        }
    }
    if (!ROWWISE) {
        mx = wave_max(mx);
        if (lane == 0) atomicMax(reinterpret_cast<unsigned int*>(out), __float_as_uint(mx));
    }
}

// tensorwise absmax of a dense [M, K] tensor (lda == K): a flat stream of 16-B chunks, four independent loads in flight per thread and
// iteration (the row-walking kernel above keeps one) — HBM-bound.  max is order-independent, so the result is identical.
__global__ __launch_bounds__(256) void fp8_absmax_flat_kernel(const bf16_t* __restrict__ x, float* __restrict__ out, long n_chunks) {
    const long tid = (long)blockIdx.x * 256 + threadIdx.x, stride = (long)gridDim.x * 256;
    float r = 0.f;
    long c = tid;
    for (; c + 3 * stride < n_chunks; c += 4 * stride) {
        const bf16x8 v0 = ld_bf16x8(x + c * 8), v1 = ld_bf16x8(x + (c + stride) * 8), v2 = ld_bf16x8(x + (c + 2 * stride) * 8),
                     v3 = ld_bf16x8(x + (c + 3 * stride) * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e)
            r = fmaxf(fmaxf(r, fmaxf(fabsf((float)v0[e]), fabsf((float)v1[e]))), fmaxf(fabsf((float)v2[e]), fabsf((float)v3[e])));
    }
    for (; c < n_chunks; c += stride) {
        const bf16x8 v = ld_bf16x8(x + c * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) r = fmaxf(r, fabsf((float)v[e]));
    }
    // ONE atomic per workgroup (same-address atomics serialise at ~11 ns each: 4 per workgroup x 6 000 workgroups cost 270 us)
    __shared__ float part[4];
    r = wave_max(r);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = r;
    __syncthreads();
    if (threadIdx.x == 0)
        atomicMax(reinterpret_cast<unsigned int*>(out), __float_as_uint(fmaxf(fmaxf(part[0], part[1]), fmaxf(part[2], part[3]))));
}

template <bool ROWWISE>
__global__ __launch_bounds__(256) void fp8_quantize_kernel(const bf16_t* __restrict__ x, const float* __restrict__ absmax,
                                                           unsigned char* __restrict__ q, float* __restrict__ scale_out, int M, int K,
                                                           long lda) {
    const long chunk = (long)blockIdx.x * 256 + threadIdx.x;  // one 8-element chunk per thread
    const int kc = K >> 3;
    const long m = chunk / kc;
    if (m >= M) return;
    const int k = (int)(chunk - m * kc) * 8;
    const float am = ROWWISE ? absmax[m] : absmax[0];
    const float scale = fmaxf(__fdiv_rn(am, FP8_MAX), FP8_MIN_SCALE);
    if (k == 0 && (ROWWISE || m == 0)) scale_out[ROWWISE ? m : 0] = scale;
    const float sb = (float)(bf16_t)scale;  // the reference divides by x_scale.to(x.dtype)
    *reinterpret_cast<int2*>(q + m * (long)K + k) = fp8_pack8(ld_bf16x8(x + m * lda + k), sb);
}

// rowwise (per-token / per-output-channel) quantisation in ONE pass (round 6): one wave per row keeps the row's 16-B chunks in registers (NCH per
// lane: K <= 512 NCH), takes their absmax, then scales and packs them — the row is read once instead of twice (the two-kernel form above: absmax
// pass + quantise pass; at [32 760, 8 960], the GELU output of the DiT's FFN, 113 + 229 us).  Same arithmetic per element, same bytes and scales.
template <int NCH>
__global__ __launch_bounds__(256) void fp8_quantize_row_kernel(const bf16_t* __restrict__ x, unsigned char* __restrict__ q, float* __restrict__ scale_out,
                                                               int M, int K, long lda) {
    const int lane = threadIdx.x & 63;
    const int m = (blockIdx.x * 256 + threadIdx.x) >> 6;
    if (m >= M) return;  // wave-uniform
    const bf16_t* row = x + (long)m * lda;
    bf16x8 v[NCH];
    float r = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int k = (lane + 64 * i) * 8;
        if (k < K) {
            v[i] = ld_bf16x8(row + k);
#pragma unroll
            for (int e = 0; e < 8; ++e) r = fmaxf(r, fabsf((float)v[i][e]));
        }
    }
    r = wave_max(r);
    const float scale = fmaxf(__fdiv_rn(r, FP8_MAX), FP8_MIN_SCALE);
    if (lane == 0) scale_out[m] = scale;
    const float sb = (float)(bf16_t)scale;  // the reference divides by x_scale.to(x.dtype)
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int k = (lane + 64 * i) * 8;
        if (k < K) *reinterpret_cast<int2*>(q + m * (long)K + k) = fp8_pack8(v[i], sb);
    }
}

// tensorwise quantisation of a dense tensor: flat chunk stream, four chunks per thread and iteration (same arithmetic per element).
__global__ __launch_bounds__(256) void fp8_quantize_flat_kernel(const bf16_t* __restrict__ x, const float* __restrict__ absmax,
                                                                unsigned char* __restrict__ q, float* __restrict__ scale_out, long n_chunks) {
    const long tid = (long)blockIdx.x * 256 + threadIdx.x, stride = (long)gridDim.x * 256;
    const float scale = fmaxf(__fdiv_rn(absmax[0], FP8_MAX), FP8_MIN_SCALE);
    if (tid == 0) scale_out[0] = scale;
    const float sb = (float)(bf16_t)scale;
    long c = tid;
    for (; c + 3 * stride < n_chunks; c += 4 * stride) {
        const bf16x8 v0 = ld_bf16x8(x + c * 8), v1 = ld_bf16x8(x + (c + stride) * 8), v2 = ld_bf16x8(x + (c + 2 * stride) * 8),
                     v3 = ld_bf16x8(x + (c + 3 * stride) * 8);
        *reinterpret_cast<int2*>(q + c * 8) = fp8_pack8(v0, sb);
        *reinterpret_cast<int2*>(q + (c + stride) * 8) = fp8_pack8(v1, sb);
        *reinterpret_cast<int2*>(q + (c + 2 * stride) * 8) = fp8_pack8(v2, sb);
        *reinterpret_cast<int2*>(q + (c + 3 * stride) * 8) = fp8_pack8(v3, sb);
    }
    for (; c < n_chunks; c += stride) *reinterpret_cast<int2*>(q + c * 8) = fp8_pack8(ld_bf16x8(x + c * 8), sb);
}

}  // namespace

extern "C" int fvk_fp8_quantize_bf16(const void* x, void* q, float* scale, float* absmax_scratch, int M, int K, long lda, int rowwise,
                                     void* stream) {
    FVK_CHECK(x && q && scale && absmax_scratch, FVK_ERR_ARG, "fvk_fp8_quantize_bf16: null pointer");
    FVK_CHECK(M > 0 && K > 0 && K % 8 == 0 && lda % 8 == 0, FVK_ERR_ARG, "fvk_fp8_quantize_bf16: M=%d K=%d lda=%ld (K, lda multiples of 8)", M, K, lda);
    hipStream_t s = (hipStream_t)stream;
    if (!rowwise && lda == K) {  // dense tensorwise (every activation of the DiT): flat streaming kernels
        if (hipMemsetAsync(absmax_scratch, 0, sizeof(float), s) != hipSuccess) {
            fvk_set_error("fvk_fp8_quantize_bf16: memset failed");
            return FVK_ERR_LAUNCH;
        }
        const long n_chunks = (long)M * (K / 8);
        const long want = (n_chunks + 1023) / 1024;  // >= 4 chunks per thread
        const unsigned blocks = (unsigned)(want < 1 ? 1 : (want > 8192 ? 8192 : want));
        const unsigned blocks_a = blocks > 512 ? 512 : blocks;  // 2 workgroups per CU: <= 512 serialised atomics (~6 us)
        hipLaunchKernelGGL(fp8_absmax_flat_kernel, dim3(blocks_a), dim3(256), 0, s, (const bf16_t*)x, absmax_scratch, n_chunks);
        FVK_LAUNCH_CHECK();
        hipLaunchKernelGGL(fp8_quantize_flat_kernel, dim3(blocks), dim3(256), 0, s, (const bf16_t*)x, absmax_scratch, (unsigned char*)q, scale,
                           n_chunks);
        FVK_LAUNCH_CHECK();
        return FVK_OK;
    }
    if (rowwise && K <= 512 * 36 && fvk::tunable(fvk::TUNE_GEMM_IMPL) != 9) {  // one pass ("gemm_impl" 9, measurement build: the two-kernel form)
        const unsigned blocks_r = (unsigned)((M + 3) / 4);
        if (K <= 512 * 4) hipLaunchKernelGGL((fp8_quantize_row_kernel<4>), dim3(blocks_r), dim3(256), 0, s, (const bf16_t*)x, (unsigned char*)q, scale, M, K, lda);
        else if (K <= 512 * 18) hipLaunchKernelGGL((fp8_quantize_row_kernel<18>), dim3(blocks_r), dim3(256), 0, s, (const bf16_t*)x, (unsigned char*)q, scale, M, K, lda);
        else hipLaunchKernelGGL((fp8_quantize_row_kernel<36>), dim3(blocks_r), dim3(256), 0, s, (const bf16_t*)x, (unsigned char*)q, scale, M, K, lda);
        FVK_LAUNCH_CHECK();
        return FVK_OK;
    }
    const int blocks_a = M < 4096 ? (M + 3) / 4 : 1024;
    if (rowwise) {
        hipLaunchKernelGGL((fp8_absmax_kernel<true>), dim3((M + 3) / 4), dim3(256), 0, s, (const bf16_t*)x, absmax_scratch, M, K, lda);
    } else {
        if (hipMemsetAsync(absmax_scratch, 0, sizeof(float), s) != hipSuccess) {
            fvk_set_error("fvk_fp8_quantize_bf16: memset failed");
            return FVK_ERR_LAUNCH;
        }
        hipLaunchKernelGGL((fp8_absmax_kernel<false>), dim3(blocks_a), dim3(256), 0, s, (const bf16_t*)x, absmax_scratch, M, K, lda);
    }
    FVK_LAUNCH_CHECK();
    const long chunks = (long)M * (K / 8);
    const unsigned blocks_q = (unsigned)((chunks + 255) / 256);
    if (rowwise)
        hipLaunchKernelGGL((fp8_quantize_kernel<true>), dim3(blocks_q), dim3(256), 0, s, (const bf16_t*)x, absmax_scratch, (unsigned char*)q, scale, M, K, lda);
    else
        hipLaunchKernelGGL((fp8_quantize_kernel<false>), dim3(blocks_q), dim3(256), 0, s, (const bf16_t*)x, absmax_scratch, (unsigned char*)q, scale, M, K, lda);
    FVK_LAUNCH_CHECK();
    return FVK_OK;
}

extern "C" int fvk_gemm_fp8(const void* x_fp8, const void* w_fp8, const float* scale_a, const float* scale_b, const void* bias, void* out,
                            int M, int N, int K, long ldc, int a_rowwise, int b_rowwise, int epilogue, const void* residual,
                            const float* gate, int rows_per_batch, void* stream) {
    FVK_CHECK(x_fp8 && w_fp8 && scale_a && scale_b && out, FVK_ERR_ARG, "fvk_gemm_fp8: null pointer");
    FVK_CHECK(M > 0 && N > 0 && K > 0 && K % 64 == 0 && N % 8 == 0 && ldc % 8 == 0, FVK_ERR_ARG,
              "fvk_gemm_fp8: M=%d N=%d K=%d ldc=%ld (K %% 64 == 0, N and ldc multiples of 8)", M, N, K, ldc);
    FVK_CHECK(epilogue == FVK_EPI_NONE || epilogue == FVK_EPI_GELU_TANH || epilogue == FVK_EPI_SILU || epilogue == FVK_EPI_RESIDUAL_GATE,
              FVK_ERR_ARG, "fvk_gemm_fp8: epilogue %d unsupported", epilogue);
    FVK_CHECK(epilogue != FVK_EPI_RESIDUAL_GATE || residual, FVK_ERR_ARG, "fvk_gemm_fp8: residual epilogue without residual");
    FVK_CHECK(((uintptr_t)x_fp8 & 15) == 0 && ((uintptr_t)w_fp8 & 15) == 0 && ((uintptr_t)out & 15) == 0, FVK_ERR_ARG,
              "fvk_gemm_fp8: operands must be 16-byte aligned");
    FVK_CHECK(255L * K + K < 0x7fffffffL, FVK_ERR_ARG, "fvk_gemm_fp8: K too large");
    fvk::GemmArgs a{};
    a.x = (const bf16_t*)x_fp8; a.w = (const bf16_t*)w_fp8; a.bias = (const bf16_t*)bias; a.out = (bf16_t*)out;
    a.residual = (const bf16_t*)residual; a.gate = gate;
    a.M = M; a.N = N; a.K = K; a.lda = K; a.ldc = ldc; a.rows_per_batch = rows_per_batch > 0 ? rows_per_batch : M;
    a.scale_a = scale_a; a.scale_b = scale_b; a.scale_a_rowwise = a_rowwise; a.scale_b_rowwise = b_rowwise;
    // gemm_w1.hip's kernel in its fp8 mode (16x16x128 MX-fp8 MFMAs) where K is a whole number of 256-element double tiles; gemm_impl 2 forces gemm_pp's
    // (its direct epilogue holds two gate rows per wave: a gate finer than 128 rows stays on gemm_pp's LDS-bounce epilogue)
    const bool fine_gate = epilogue == FVK_EPI_RESIDUAL_GATE && gate && a.rows_per_batch < 128;
    if (fvk::tunable(fvk::TUNE_GEMM_IMPL) != 2 && !fine_gate && fvk::gemm_w1_fp8_eligible(a)) return fvk::gemm_w1_fp8_launch(a, epilogue, (hipStream_t)stream);
    return fvk::gemm_pp_fp8_launch(a, epilogue, (hipStream_t)stream);
}
