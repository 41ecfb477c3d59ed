// Tile cross-fade and pixel post-processing of the Wan VAE decode (HBM-bound elementwise kernels).
// ref: fastvideo/models/vaes/common.py:94-114  blend_v / blend_h / blend_t — a python loop of `extent` slice assignments
//      b[i] = a[-extent + i] * (1 - i/extent) + b[i] * (i/extent), i.e. 4 eager kernels per slice; here one launch per tile edge.
//      fastvideo/pipelines/stages/decoding.py:210 (image/2 + 0.5).clamp(0, 1)  and
//      fastvideo/entrypoints/video_generator.py:912-913 (src*255).clamp_(0,255).to(uint8) + "b c t h w -> t b c h w".
// Built with -ffp-contract=off: the eager reference rounds the two products and the sum separately, so must we (bit-exact fp32).
#include "fvk_common.h"

namespace {

struct BlendArgs {
    const float* a; float* b;
    long a_off;                 // element offset of a's first blended slice (= (len_a - extent) * stride_a)
    long sa_o0, sa_o1, sb_o0, sb_o1;  // strides of the two (collapsed) dims OUTSIDE the blend axis, e.g. C and T*H of a [C,T,H,W] view
    long n_o1;                  // extent of the second outer dim (outer = n_o0 * n_o1)
    long sa_axis, sb_axis;      // stride of the blend axis
    long inner, outer;          // contiguous elements inside one slice of the axis; number of outer indices
    int extent;
};

// 1-D grid over (outer, i < extent, inner / V) with V = 4 when every slice is 16-B aligned (VEC) else 1; consecutive threads walk
// the contiguous inner run first, then the axis (so a W-axis blend, inner = 1, still coalesces over the `extent` columns).
// The cross-fade weights are computed the way python does (double), then rounded to fp32 like torch's scalar operand.
template <bool VEC>
__global__ __launch_bounds__(256) void vae_blend_kernel(BlendArgs p) {
#pragma clang fp contract(off)
    const long per = VEC ? p.inner / 4 : p.inner;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= per * p.extent * p.outer) return;
    const long e = (idx % per) * (VEC ? 4 : 1);
    const int i = (int)((idx / per) % p.extent);
    const long o = idx / (per * p.extent);
    const double r = (double)i / (double)p.extent;
    const float wa = (float)(1.0 - r), wb = (float)r;
    const long o0 = o / p.n_o1, o1 = o % p.n_o1;
    const float* a = p.a + p.a_off + o0 * p.sa_o0 + o1 * p.sa_o1 + (long)i * p.sa_axis + e;
    float* b = p.b + o0 * p.sb_o0 + o1 * p.sb_o1 + (long)i * p.sb_axis + e;
    if (VEC) {
        const float4 va = *(const float4*)a;
        float4 vb = *(float4*)b;
        vb.x = va.x * wa + vb.x * wb; vb.y = va.y * wa + vb.y * wb; vb.z = va.z * wa + vb.z * wb; vb.w = va.w * wa + vb.w * wb;
        *(float4*)b = vb;
    } else {
        *b = *a * wa + *b * wb;
    }
}

// planar fp32 [3][T][H*W] in [-1,1] -> interleaved u8 [T][H*W][3]; 4 pixels per thread (3 x 16-B loads, 12-B store)
__global__ __launch_bounds__(256) void vae_post_u8_kernel(const float* __restrict__ x, unsigned char* __restrict__ y, long hw, long plane) {
#pragma clang fp contract(off)
    const long t = blockIdx.y;
    const long p0 = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (p0 >= hw) return;
    const int cnt = hw - p0 < 4 ? (int)(hw - p0) : 4;
    unsigned char q[12];
    for (int c = 0; c < 3; ++c) {
        const float* src = x + c * plane + t * hw + p0;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (cnt == 4 && ((size_t)src & 15) == 0) {
            const float4 f = *(const float4*)src;
            v[0] = f.x; v[1] = f.y; v[2] = f.z; v[3] = f.w;
        } else {
            for (int e = 0; e < cnt; ++e) v[e] = src[e];
        }
        for (int e = 0; e < 4; ++e) {
            float u = v[e] / 2.0f + 0.5f;
            u = fminf(fmaxf(u, 0.0f), 1.0f);
            u = u * 255.0f;
            u = fminf(fmaxf(u, 0.0f), 255.0f);
            q[e * 3 + c] = (unsigned char)(int)u;  // .to(uint8) truncates toward zero
        }
    }
    unsigned char* dst = y + (t * hw + p0) * 3;
    if (cnt == 4 && ((size_t)dst & 3) == 0) {
        unsigned int* d32 = (unsigned int*)dst;
        d32[0] = q[0] | (q[1] << 8) | (q[2] << 16) | ((unsigned)q[3] << 24);
        d32[1] = q[4] | (q[5] << 8) | (q[6] << 16) | ((unsigned)q[7] << 24);
        d32[2] = q[8] | (q[9] << 8) | (q[10] << 16) | ((unsigned)q[11] << 24);
    } else {
        for (int e = 0; e < cnt * 3; ++e) dst[e] = q[e];
    }
}

}  // namespace

// a, b: fp32 tensors of logical shape [outer0, outer1, len, inner] addressed with explicit strides (inner is contiguous in both):
// b[o0, o1, i, :] = a[o0, o1, len_a - e + i, :] * (1 - i/e) + b[o0, o1, i, :] * (i/e)  for i < e = min(extent, len_a, len_b)  (in place on b).
extern "C" int fvk_vae_blend_f32(const float* a, float* b, long outer0, long outer1, long inner, int len_a, int len_b, int extent,
                                 long a_stride0, long a_stride1, long a_axis_stride, long b_stride0, long b_stride1, long b_axis_stride,
                                 void* stream) {
    const long outer = outer0 * outer1;
    FVK_CHECK(a && b, FVK_ERR_ARG, "fvk_vae_blend_f32: null pointer");
    FVK_CHECK(outer0 > 0 && outer1 > 0 && inner > 0 && len_a > 0 && len_b > 0 && extent >= 0, FVK_ERR_ARG,
              "fvk_vae_blend_f32: empty shape outer=%ldx%ld inner=%ld len_a=%d len_b=%d extent=%d", outer0, outer1, inner, len_a, len_b, extent);
    int ext = extent < len_a ? extent : len_a;   // common.py:95 — the extent is clamped to both tiles
    ext = ext < len_b ? ext : len_b;
    if (ext == 0) return FVK_OK;
    BlendArgs p{};
    p.a = a; p.b = b; p.a_off = (long)(len_a - ext) * a_axis_stride; p.sa_o0 = a_stride0; p.sa_o1 = a_stride1; p.sb_o0 = b_stride0; p.sb_o1 = b_stride1; p.n_o1 = outer1;
    p.sa_axis = a_axis_stride; p.sb_axis = b_axis_stride; p.inner = inner; p.outer = outer; p.extent = ext;
    const bool vec = inner % 4 == 0 && (((size_t)a | (size_t)b) & 15) == 0 && ((p.a_off | a_stride0 | a_stride1 | a_axis_stride | b_stride0 | b_stride1 | b_axis_stride) & 3) == 0;
    const long n = (vec ? inner / 4 : inner) * ext * outer;
    FVK_CHECK((n + 255) / 256 < 0x7FFFFFFFL, FVK_ERR_ARG, "fvk_vae_blend_f32: %ld work items exceed one launch", n);
    const dim3 grid((unsigned)((n + 255) / 256));
    if (vec) hipLaunchKernelGGL(vae_blend_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(vae_blend_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, p);
    FVK_LAUNCH_CHECK();
    return FVK_OK;
}

// pixels: planar fp32 [3, T, H, W] (plane_stride elements between channel planes) in [-1, 1]  ->  frames: u8 [T, H, W, 3]
extern "C" int fvk_vae_postprocess_u8(const float* pixels, void* frames_u8, int T, int H, int W, long plane_stride, void* stream) {
    FVK_CHECK(pixels && frames_u8, FVK_ERR_ARG, "fvk_vae_postprocess_u8: null pointer");
    FVK_CHECK(T > 0 && H > 0 && W > 0 && T <= 65535 && plane_stride >= (long)T * H * W, FVK_ERR_ARG,
              "fvk_vae_postprocess_u8: bad shape T=%d H=%d W=%d plane_stride=%ld", T, H, W, plane_stride);
    const long hw = (long)H * W;
    hipLaunchKernelGGL(vae_post_u8_kernel, dim3((unsigned)((hw + 1023) / 1024), (unsigned)T), dim3(256), 0, (hipStream_t)stream, pixels,
                       (unsigned char*)frames_u8, hw, plane_stride);
    FVK_LAUNCH_CHECK();
    return FVK_OK;
}
