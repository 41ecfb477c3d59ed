// OCP e4m3 quantisation arithmetic shared by fp8.hip (the stand-alone quantisers) and norm_mod.hip (per-token quantisation fused into the
// LayerNorm / modulation pass).  ref: fastvideo/layers/quantization/fp8_config.py:55-68.
//   scale = max(absmax / 448, 1 / (448 * 512))   (fp32);   q = e4m3fn_rne( clamp( bf16( float(x) / float(bf16(scale)) ), -448, 448 ) )
#pragma once
#include "fvk_common.h"

namespace fvk {

constexpr float FP8_MAX = 448.0f;
constexpr float FP8_MIN_SCALE = 1.0f / (448.0f * 512.0f);

__device__ __forceinline__ float fp8_scale_of(float absmax) { return fmaxf(__fdiv_rn(absmax, FP8_MAX), FP8_MIN_SCALE); }

// eight bf16 values / sb (the scale rounded to bf16: the reference divides by x_scale.to(x.dtype)) -> eight e4m3 bytes
__device__ __forceinline__ int2 fp8_pack8(const bf16x8 v, float sb) {
    float f[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float d = (float)(bf16_t)__fdiv_rn((float)v[e], sb);
        f[e] = fminf(fmaxf(d, -FP8_MAX), FP8_MAX);
    }
    int w0 = 0, w1 = 0;
    w0 = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], w0, false);
    w0 = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], w0, true);
    w1 = __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], w1, false);
    w1 = __builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], w1, true);
    return make_int2(w0, w1);
}

}  // namespace fvk
