// HBM-bound norm / modulate / RoPE family for the Wan DiT block (gfx950).
//
// Design: ONE WAVE PER ROW.  A row of d bf16 (d = 1536 for Wan2.1-1.3B, 5120 for A14B) is held entirely in
// registers as VPL chunks of 8 elements per lane (16-byte global loads, 1 KiB per wave-instruction,
// fully coalesced); mean / variance / sum-of-squares are wave64 shuffle reductions — no LDS, no barrier,
// every byte of the row is read once and written once.  4 waves (rows) per 256-thread workgroup.
//
// The fp32 rounding points follow the reference's eager path; fused-multiply-add
// contraction is disabled in this file so separately-rounded reference ops stay separately rounded.
#pragma clang fp contract(off)
#include "fvk_common.h"
#include "fp8_common.h"

namespace {

struct LnArgs {
    const bf16_t* x;
    const bf16_t* residual;
    const float* gate;
    const float* ln_w;
    const float* ln_b;
    const float* mul;
    const float* add;
    bf16_t* res_out;
    bf16_t* out;
    int M, d, rows_per_batch;
    float eps;
    int flags;
    // fvk_ln_modulate_fp8_bf16: the row's PER-TOKEN e4m3 quantisation (fp8_config.py:62-68 applied to the bf16 output row, which one wave
    // holds in registers) written beside — or instead of (out == NULL) — the bf16 row: q_out [M, d] bytes, q_scale [M] fp32
    unsigned char* q_out;
    float* q_scale;
};

template <int VPL, bool QUANT = false>
__global__ __launch_bounds__(256) void ln_modulate_kernel(LnArgs a) {
    bf16x8 ov[QUANT ? VPL : 1];  // QUANT: the output row, kept for the quantisation pass
    float amax = 0.f;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= a.M) return;
    const int nchunks = a.d >> 3;
    const long rbase = (long)row * a.d;
    const long bbase = (long)(row / a.rows_per_batch) * a.d;
    float v[VPL][8];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int c = lane + 64 * i;
        if (c < nchunks) {
            bf16x8 xv = ld_bf16x8(a.x + rbase + c * 8);
            if (a.residual) {
                bf16x8 rv = ld_bf16x8(a.residual + rbase + c * 8);
                if (a.gate) {
                    const float4 g0 = *reinterpret_cast<const float4*>(a.gate + bbase + c * 8);
                    const float4 g1 = *reinterpret_cast<const float4*>(a.gate + bbase + c * 8 + 4);
                    const float g[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[i][j] = (float)rv[j] + (float)xv[j] * g[j];
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[i][j] = (float)rv[j] + (float)xv[j];
                }
                if (a.flags & FVK_LN_ROUND_RESIDUAL) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[i][j] = bf16_round(v[i][j]);
                }
                if (a.res_out) {
                    bf16x8 ro;
#pragma unroll
                    for (int j = 0; j < 8; ++j) ro[j] = (bf16_t)v[i][j];
                    st_bf16x8(a.res_out + rbase + c * 8, ro);
                }
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[i][j] = (float)xv[j];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) sum += v[i][j];
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[i][j] = 0.f;
        }
    }
    const float inv_d = 1.0f / (float)a.d;
    const float mean = wave_sum(sum) * inv_d;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        if (lane + 64 * i < nchunks) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float t = v[i][j] - mean;
                sq += t * t;
            }
        }
    }
    const float rstd = rsqrtf(wave_sum(sq) * inv_d + a.eps);
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int c = lane + 64 * i;
        if (c < nchunks) {
            float n[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) n[j] = (v[i][j] - mean) * rstd;
            if (a.ln_w) {
                const float4 w0 = *reinterpret_cast<const float4*>(a.ln_w + c * 8);
                const float4 w1 = *reinterpret_cast<const float4*>(a.ln_w + c * 8 + 4);
                const float w[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
                for (int j = 0; j < 8; ++j) n[j] = n[j] * w[j];
            }
            if (a.ln_b) {
                const float4 b0 = *reinterpret_cast<const float4*>(a.ln_b + c * 8);
                const float4 b1 = *reinterpret_cast<const float4*>(a.ln_b + c * 8 + 4);
                const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
                for (int j = 0; j < 8; ++j) n[j] = n[j] + bb[j];
            }
            if (a.flags & FVK_LN_ROUND_NORM) {
#pragma unroll
                for (int j = 0; j < 8; ++j) n[j] = bf16_round(n[j]);
            }
            if (a.mul) {
                const float4 m0 = *reinterpret_cast<const float4*>(a.mul + bbase + c * 8);
                const float4 m1 = *reinterpret_cast<const float4*>(a.mul + bbase + c * 8 + 4);
                const float mm[8] = {m0.x, m0.y, m0.z, m0.w, m1.x, m1.y, m1.z, m1.w};
#pragma unroll
                for (int j = 0; j < 8; ++j) n[j] = n[j] * mm[j];
            }
            if (a.add) {
                const float4 s0 = *reinterpret_cast<const float4*>(a.add + bbase + c * 8);
                const float4 s1 = *reinterpret_cast<const float4*>(a.add + bbase + c * 8 + 4);
                const float ss[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
#pragma unroll
                for (int j = 0; j < 8; ++j) n[j] = n[j] + ss[j];
            }
            bf16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (bf16_t)n[j];
            if (a.out) st_bf16x8(a.out + rbase + c * 8, o);
            if (QUANT) {
                ov[i] = o;
#pragma unroll
                for (int j = 0; j < 8; ++j) amax = fmaxf(amax, fabsf((float)o[j]));
            }
        }
    }
    if (QUANT) {  // the stand-alone quantiser's arithmetic (fp8.hip: fp8_absmax_kernel<true> + fp8_quantize_kernel<true>) on the row in registers
        const float scale = fvk::fp8_scale_of(wave_max(amax));
        if (lane == 0) a.q_scale[row] = scale;
        const float sb = (float)(bf16_t)scale;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int c = lane + 64 * i;
            if (c < nchunks) *reinterpret_cast<int2*>(a.q_out + rbase + c * 8) = fvk::fp8_pack8(ov[i], sb);
        }
    }
}

struct RmsArgs {
    const bf16_t* in[4];
    bf16_t* out[4];
    const bf16_t* w[4];
    const float* cos;
    const float* sin;
    int M, width, head_dim, seq_len, pos_offset;
    long in_stride, out_stride;
    float eps;
    int rope_mask;       // bit t: tensor t is rotated (when cos/sin are given)
    // sequence-parallel exchange packing (fvk_qkv_norm_rope_pack_bf16): when pack_dst is set, column block g (pack_W wide) of row m of
    // tensor t is written to every destination rank rp = g + pack_G*u', u' < pack_U, at pack_dst[((rp*M + m)*pack_ns + pack_slot[t])*pack_W + cc]
    bf16_t* pack_dst;
    int pack_G, pack_U, pack_W, pack_ns;  // pack_ns = slots per message row: 3 = [K | V | Q], 4 = [K | V | Q | gate]
    // two head chunks per group (fvk_qkv_norm_rope_pack2_bf16, the pipelined exchange): the first pack_Wa columns of every group go to pack_dst
    // (message rows pack_Wa wide), the remaining pack_W - pack_Wa to pack_dst2; pack_Wa == pack_W: one buffer
    bf16_t* pack_dst2;
    int pack_Wa;
    int pack_slot[4];
    // scatter (fvk_rmsnorm_rope_scatter_bf16): row m of tensor t is written to row row_map[t][m] of out[t] (negative: dropped); NULL = row m.
    // Folds the tile-major / window-class gathers of the sparse attention paths into this pass.
    const int32_t* row_map[4];
};

template <int VPL>
__global__ __launch_bounds__(256) void rmsnorm_rope_kernel(RmsArgs a) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= a.M) return;
    const int t = blockIdx.y;
    const int orow = a.row_map[t] ? a.row_map[t][row] : row;
    if (orow < 0) return;
    const bf16_t* in = a.in[t] + (long)row * a.in_stride;
    bf16_t* out = a.out[t] + (long)orow * a.out_stride;
    const bf16_t* w = a.w[t];
    const int nchunks = a.width >> 3;
    float v[VPL][8];
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int c = lane + 64 * i;
        if (c < nchunks) {
            bf16x8 xv = ld_bf16x8(in + c * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                v[i][j] = (float)xv[j];
                ss += v[i][j] * v[i][j];
            }
        }
    }
    if (w) {
        const float r = rsqrtf(wave_sum(ss) / (float)a.width + a.eps);
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int c = lane + 64 * i;
            if (c < nchunks) {
                bf16x8 wv = ld_bf16x8(w + c * 8);
#pragma unroll
                for (int j = 0; j < 8; ++j) v[i][j] = bf16_round(bf16_round(v[i][j] * r) * (float)wv[j]);
            }
        }
    }
    const long pos = (long)((row + a.pos_offset) % a.seq_len) * a.head_dim;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int c = lane + 64 * i;
        if (c < nchunks) {
            bf16x8 o;
            if (a.cos && ((a.rope_mask >> t) & 1)) {
                const int dd = (c * 8) % a.head_dim;
                const float4 c0 = *reinterpret_cast<const float4*>(a.cos + pos + dd);
                const float4 c1 = *reinterpret_cast<const float4*>(a.cos + pos + dd + 4);
                const float4 s0 = *reinterpret_cast<const float4*>(a.sin + pos + dd);
                const float4 s1 = *reinterpret_cast<const float4*>(a.sin + pos + dd + 4);
                const float cs[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
                const float sn[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
#pragma unroll
                for (int j = 0; j < 8; j += 2) {
                    const float xr = v[i][j], xi = v[i][j + 1];
                    o[j] = (bf16_t)(xr * cs[j] + (-xi) * sn[j]);
                    o[j + 1] = (bf16_t)(xi * cs[j + 1] + xr * sn[j + 1]);
                }
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = (bf16_t)v[i][j];
            }
            if (a.pack_dst) {
                const int col = c * 8, g = col / a.pack_W, cc = col - g * a.pack_W;
                const bool second = cc >= a.pack_Wa;
                bf16_t* dst = second ? a.pack_dst2 : a.pack_dst;
                const int wc = second ? a.pack_W - a.pack_Wa : a.pack_Wa, c2 = second ? cc - a.pack_Wa : cc;
                for (int uu = 0; uu < a.pack_U; ++uu) {
                    const long rp = g + (long)a.pack_G * uu;
                    st_bf16x8(dst + ((rp * a.M + row) * a.pack_ns + a.pack_slot[t]) * wc + c2, o);
                }
            } else {
                st_bf16x8(out + c * 8, o);
            }
        }
    }
}

__global__ __launch_bounds__(256) void scale_residual_kernel(const bf16_t* residual, const bf16_t* x, const float* gate,
                                                             bf16_t* out, long n_chunks, int d, long rows_per_batch_d) {
    for (long c = blockIdx.x * 256L + threadIdx.x; c < n_chunks; c += (long)gridDim.x * 256L) {
        const long e = c * 8;
        bf16x8 rv = ld_bf16x8(residual + e), xv = ld_bf16x8(x + e);
        bf16x8 o;
        if (gate) {
            const long g = (e / rows_per_batch_d) * d + (e % d);
            const float4 g0 = *reinterpret_cast<const float4*>(gate + g);
            const float4 g1 = *reinterpret_cast<const float4*>(gate + g + 4);
            const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (bf16_t)((float)rv[j] + (float)xv[j] * gg[j]);
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (bf16_t)((float)rv[j] + (float)xv[j]);
        }
        st_bf16x8(out + e, o);
    }
}

// V [B,S,H,128] -> Vt [B,H,128,S_pad], key order within each 16-group permuted (swap bits 2,3), pad zero.
// One workgroup per (64-key tile, head, batch); transposition through LDS.
// src_rows (optional, int32 [S_pad]): key position p of V^T takes row src_rows[p] of v (negative: zeros) — the tile-major gather of the sparse
// attention paths folded into the transpose (S then bounds the SOURCE rows).
__global__ __launch_bounds__(256) void v_transpose_kernel(const bf16_t* v, bf16_t* vt, int S, int H, long in_stride,
                                                          long in_batch_stride, long in_head_stride, int S_pad, const int32_t* src_rows) {
    __shared__ bf16_t tile[64][128 + 8];
    const int kt = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    const int tid = threadIdx.x;
    const bf16_t* src = v + (long)b * in_batch_stride + (long)h * in_head_stride;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = tid + 256 * i;  // 1024 chunks: 64 rows x 16 chunks
        const int r = c >> 4, ch = c & 15;
        int key = kt * 64 + r;
        if (src_rows) key = src_rows[key];
        bf16x8 val;
        if (key >= 0 && key < S) {
            val = ld_bf16x8(src + (long)key * in_stride + ch * 8);
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) val[j] = (bf16_t)0.f;
        }
        *reinterpret_cast<bf16x8*>(&tile[r][ch * 8]) = val;
    }
    __syncthreads();
    bf16_t* dst = vt + (((long)b * H + h) * 128) * S_pad + (long)kt * 64;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = tid + 256 * i;  // 1024 chunks: 128 d-rows x 8 chunks of 8 (permuted) keys
        const int d = c >> 3, ch = c & 7;
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int p = ch * 8 + j;                                           // stored position in the tile
            const int key = (p & ~12) | ((p & 4) << 1) | ((p & 8) >> 1);        // swap bits 2 and 3
            o[j] = tile[key][d];
        }
        st_bf16x8(dst + (long)d * S_pad + ch * 8, o);
    }
}

template <typename F>
int dispatch_vpl(int width, F&& f) {
    const int need = (width / 8 + 63) / 64;
    if (need <= 1) return f(std::integral_constant<int, 1>{});
    if (need <= 2) return f(std::integral_constant<int, 2>{});
    if (need <= 3) return f(std::integral_constant<int, 3>{});
    if (need <= 4) return f(std::integral_constant<int, 4>{});
    if (need <= 6) return f(std::integral_constant<int, 6>{});
    if (need <= 8) return f(std::integral_constant<int, 8>{});
    if (need <= 10) return f(std::integral_constant<int, 10>{});
    if (need <= 16) return f(std::integral_constant<int, 16>{});
    return FVK_ERR_ARG;
}

}  // namespace

extern "C" int fvk_ln_modulate_bf16(const void* x, const void* residual, const float* gate, const float* ln_w,
                                    const float* ln_b, const float* mul, const float* add, void* res_out, void* out,
                                    int M, int d, int rows_per_batch, float eps, int flags, void* stream) {
    FVK_CHECK(x && out, FVK_ERR_ARG, "fvk_ln_modulate_bf16: null x/out");
    FVK_CHECK(M >= 0 && d > 0 && d % 8 == 0 && d <= 8192, FVK_ERR_ARG, "fvk_ln_modulate_bf16: d=%d must be a multiple of 8 and <= 8192", d);
    FVK_CHECK(rows_per_batch > 0, FVK_ERR_ARG, "fvk_ln_modulate_bf16: rows_per_batch must be > 0");
    FVK_CHECK(!(gate && !residual), FVK_ERR_ARG, "fvk_ln_modulate_bf16: gate without residual");
    if (M == 0) return FVK_OK;
    LnArgs a{(const bf16_t*)x, (const bf16_t*)residual, gate, ln_w, ln_b, mul, add, (bf16_t*)res_out, (bf16_t*)out,
             M, d, rows_per_batch, eps, flags, nullptr, nullptr};
    int rc = dispatch_vpl(d, [&](auto vpl) {
        hipLaunchKernelGGL((ln_modulate_kernel<decltype(vpl)::value>), dim3((M + 3) / 4), dim3(256), 0,
                           (hipStream_t)stream, a);
        return FVK_OK;
    });
    FVK_CHECK(rc == FVK_OK, rc, "fvk_ln_modulate_bf16: unsupported d=%d", d);
    FVK_LAUNCH_CHECK();
    return FVK_OK;
}

extern "C" int fvk_ln_modulate_fp8_bf16(const void* x, const void* residual, const float* gate, const float* ln_w, const float* ln_b,
                                        const float* mul, const float* add, void* res_out, void* out, void* q_out, float* q_scale, int M,
                                        int d, int rows_per_batch, float eps, int flags, void* stream) {
    FVK_CHECK(x && q_out && q_scale, FVK_ERR_ARG, "fvk_ln_modulate_fp8_bf16: null x / q_out / q_scale");
    FVK_CHECK(M >= 0 && d > 0 && d % 8 == 0 && d <= 8192, FVK_ERR_ARG, "fvk_ln_modulate_fp8_bf16: d=%d must be a multiple of 8 and <= 8192", d);
    FVK_CHECK(rows_per_batch > 0, FVK_ERR_ARG, "fvk_ln_modulate_fp8_bf16: rows_per_batch must be > 0");
    FVK_CHECK(!(gate && !residual), FVK_ERR_ARG, "fvk_ln_modulate_fp8_bf16: gate without residual");
    if (M == 0) return FVK_OK;
    LnArgs a{(const bf16_t*)x, (const bf16_t*)residual, gate, ln_w, ln_b, mul, add, (bf16_t*)res_out, (bf16_t*)out,
             M, d, rows_per_batch, eps, flags, (unsigned char*)q_out, q_scale};
    int rc = dispatch_vpl(d, [&](auto vpl) {
        hipLaunchKernelGGL((ln_modulate_kernel<decltype(vpl)::value, true>), dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream, a);
        return FVK_OK;
    });
    FVK_CHECK(rc == FVK_OK, rc, "fvk_ln_modulate_fp8_bf16: unsupported d=%d", d);
    FVK_LAUNCH_CHECK();
    return FVK_OK;
}

extern "C" int fvk_scale_residual_bf16(const void* residual, const void* x, const float* gate, void* out, int M, int d,
                                       int rows_per_batch, void* stream) {
    FVK_CHECK(residual && x && out, FVK_ERR_ARG, "fvk_scale_residual_bf16: null pointer");
    FVK_CHECK(d > 0 && d % 8 == 0 && rows_per_batch > 0, FVK_ERR_ARG, "fvk_scale_residual_bf16: bad d=%d", d);
    if (M <= 0) return FVK_OK;
    const long n_chunks = (long)M * d / 8;
    const int grid = (int)((n_chunks + 255) / 256 < 4096 ? (n_chunks + 255) / 256 : 4096);
    hipLaunchKernelGGL(scale_residual_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)residual,
                       (const bf16_t*)x, gate, (bf16_t*)out, n_chunks, d, (long)rows_per_batch * d);
    FVK_LAUNCH_CHECK();
    return FVK_OK;
}

static int rmsnorm_rope_impl(const void* const* in, void* const* out, const void* const* weight, int n_tensors, const float* cos,
                             const float* sin, int M, int width, int head_dim, int seq_len, int pos_offset, long in_stride,
                             long out_stride, float eps, const int32_t* const* row_map, void* stream);

extern "C" int fvk_rmsnorm_rope_bf16(const void* const* in, void* const* out, const void* const* weight, int n_tensors,
                                     const float* cos, const float* sin, int M, int width, int head_dim, int seq_len,
                                     int pos_offset, long in_stride, long out_stride, float eps, void* stream) {
    return rmsnorm_rope_impl(in, out, weight, n_tensors, cos, sin, M, width, head_dim, seq_len, pos_offset, in_stride, out_stride, eps,
                             nullptr, stream);
}

extern "C" int fvk_rmsnorm_rope_scatter_bf16(const void* const* in, void* const* out, const void* const* weight, int n_tensors,
                                             const float* cos, const float* sin, int M, int width, int head_dim, int seq_len,
                                             int pos_offset, long in_stride, long out_stride, float eps,
                                             const int32_t* const* row_map, void* stream) {
    FVK_CHECK(row_map, FVK_ERR_ARG, "fvk_rmsnorm_rope_scatter_bf16: null row_map array (use fvk_rmsnorm_rope_bf16)");
    return rmsnorm_rope_impl(in, out, weight, n_tensors, cos, sin, M, width, head_dim, seq_len, pos_offset, in_stride, out_stride, eps,
                             row_map, stream);
}

static int rmsnorm_rope_impl(const void* const* in, void* const* out, const void* const* weight, int n_tensors, const float* cos,
                             const float* sin, int M, int width, int head_dim, int seq_len, int pos_offset, long in_stride,
                             long out_stride, float eps, const int32_t* const* row_map, void* stream) {
    FVK_CHECK(in && out && n_tensors >= 1 && n_tensors <= 3, FVK_ERR_ARG, "fvk_rmsnorm_rope_bf16: n_tensors=%d", n_tensors);
    FVK_CHECK(width > 0 && width % 8 == 0 && head_dim > 0 && head_dim % 8 == 0 && width % head_dim == 0, FVK_ERR_ARG,
              "fvk_rmsnorm_rope_bf16: width=%d head_dim=%d", width, head_dim);
    FVK_CHECK((cos == nullptr) == (sin == nullptr), FVK_ERR_ARG, "fvk_rmsnorm_rope_bf16: cos/sin must both be set");
    FVK_CHECK(pos_offset >= 0 && seq_len > 0 && in_stride % 8 == 0 && out_stride % 8 == 0, FVK_ERR_ARG, "fvk_rmsnorm_rope_bf16: strides must be multiples of 8");
    if (M <= 0) return FVK_OK;
    RmsArgs a{};
    for (int i = 0; i < n_tensors; ++i) {
        FVK_CHECK(in[i] && out[i], FVK_ERR_ARG, "fvk_rmsnorm_rope_bf16: null tensor %d", i);
        a.in[i] = (const bf16_t*)in[i];
        a.out[i] = (bf16_t*)out[i];
        a.w[i] = weight ? (const bf16_t*)weight[i] : nullptr;
        a.row_map[i] = row_map ? row_map[i] : nullptr;
    }
    a.cos = cos; a.sin = sin; a.M = M; a.width = width; a.head_dim = head_dim; a.seq_len = seq_len; a.pos_offset = pos_offset;
    a.in_stride = in_stride; a.out_stride = out_stride; a.eps = eps; a.rope_mask = 7;
    int rc = dispatch_vpl(width, [&](auto vpl) {
        hipLaunchKernelGGL((rmsnorm_rope_kernel<decltype(vpl)::value>), dim3((M + 3) / 4, n_tensors), dim3(256), 0,
                           (hipStream_t)stream, a);
        return FVK_OK;
    });
    FVK_CHECK(rc == FVK_OK, rc, "fvk_rmsnorm_rope_bf16: unsupported width=%d", width);
    FVK_LAUNCH_CHECK();
    return FVK_OK;
}

static int qkv_pack_impl(const void* q, const void* k, const void* v, const void* gate, const void* wq, const void* wk, const float* cos,
                         const float* sin, void* send, int Sl, int width, int head_dim, int seq_len, int pos_offset, long in_stride, int G, int U,
                         float eps, void* stream, void* send2 = nullptr, int heads_a = 0);

extern "C" int fvk_qkv_norm_rope_pack_bf16(const void* q, const void* k, const void* v, const void* wq, const void* wk, const float* cos,
                                           const float* sin, void* send, int Sl, int width, int head_dim, int seq_len, int pos_offset,
                                           long in_stride, int G, int U, float eps, void* stream) {
    return qkv_pack_impl(q, k, v, nullptr, wq, wk, cos, sin, send, Sl, width, head_dim, seq_len, pos_offset, in_stride, G, U, eps, stream);
}

extern "C" int fvk_qkvg_norm_rope_pack_bf16(const void* q, const void* k, const void* v, const void* gate, const void* wq, const void* wk,
                                            const float* cos, const float* sin, void* send, int Sl, int width, int head_dim, int seq_len,
                                            int pos_offset, long in_stride, int G, int U, float eps, void* stream) {
    FVK_CHECK(gate, FVK_ERR_ARG, "fvk_qkvg_norm_rope_pack_bf16: null gate (use fvk_qkv_norm_rope_pack_bf16)");
    return qkv_pack_impl(q, k, v, gate, wq, wk, cos, sin, send, Sl, width, head_dim, seq_len, pos_offset, in_stride, G, U, eps, stream);
}

extern "C" int fvk_qkv_norm_rope_pack2_bf16(const void* q, const void* k, const void* v, const void* wq, const void* wk, const float* cos,
                                            const float* sin, void* send_a, void* send_b, int heads_a, int Sl, int width, int head_dim, int seq_len,
                                            int pos_offset, long in_stride, int G, int U, float eps, void* stream) {
    FVK_CHECK(send_b && G >= 1 && head_dim > 0 && width % (G * head_dim) == 0 && heads_a >= 1 && heads_a < width / head_dim / G, FVK_ERR_ARG,
              "fvk_qkv_norm_rope_pack2_bf16: heads_a=%d must leave both chunks of a %d-head group non-empty (and send_b set)", heads_a,
              (G >= 1 && head_dim > 0) ? width / head_dim / G : 0);
    return qkv_pack_impl(q, k, v, nullptr, wq, wk, cos, sin, send_a, Sl, width, head_dim, seq_len, pos_offset, in_stride, G, U, eps, stream, send_b,
                         heads_a);
}

static int qkv_pack_impl(const void* q, const void* k, const void* v, const void* gate, const void* wq, const void* wk, const float* cos,
                         const float* sin, void* send, int Sl, int width, int head_dim, int seq_len, int pos_offset, long in_stride, int G, int U,
                         float eps, void* stream, void* send2, int heads_a) {
    FVK_CHECK(q && k && v && send, FVK_ERR_ARG, "fvk_qkv_norm_rope_pack_bf16: null pointer");
    FVK_CHECK(width > 0 && head_dim > 0 && head_dim % 8 == 0 && width % head_dim == 0, FVK_ERR_ARG,
              "fvk_qkv_norm_rope_pack_bf16: width=%d head_dim=%d", width, head_dim);
    FVK_CHECK(G >= 1 && U >= 1 && (width / head_dim) % G == 0, FVK_ERR_ARG,
              "fvk_qkv_norm_rope_pack_bf16: %d heads do not split into G=%d head groups", width / head_dim, G);
    FVK_CHECK((cos == nullptr) == (sin == nullptr), FVK_ERR_ARG, "fvk_qkv_norm_rope_pack_bf16: cos/sin must both be set");
    FVK_CHECK(pos_offset >= 0 && seq_len > 0 && in_stride % 8 == 0, FVK_ERR_ARG, "fvk_qkv_norm_rope_pack_bf16: strides must be multiples of 8");
    if (Sl <= 0) return FVK_OK;
    RmsArgs a{};
    const int ns = gate ? 4 : 3;
    a.in[0] = (const bf16_t*)q; a.in[1] = (const bf16_t*)k; a.in[2] = (const bf16_t*)v; a.in[3] = (const bf16_t*)gate;
    a.out[0] = a.out[1] = a.out[2] = a.out[3] = (bf16_t*)send;  // unused: every store goes through pack_dst
    a.w[0] = (const bf16_t*)wq; a.w[1] = (const bf16_t*)wk; a.w[2] = a.w[3] = nullptr;  // V, gate: no norm
    a.cos = cos; a.sin = sin; a.M = Sl; a.width = width; a.head_dim = head_dim; a.seq_len = seq_len; a.pos_offset = pos_offset;
    a.in_stride = in_stride; a.out_stride = 0; a.eps = eps;
    a.rope_mask = 3;  // q and k are rotated, v (and the gate) are copied
    a.pack_dst = (bf16_t*)send; a.pack_G = G; a.pack_U = U; a.pack_W = width / G; a.pack_ns = ns;
    a.pack_dst2 = (bf16_t*)send2; a.pack_Wa = send2 ? heads_a * head_dim : a.pack_W;
    a.pack_slot[0] = 2; a.pack_slot[1] = 0; a.pack_slot[2] = 1; a.pack_slot[3] = 3;  // message row = [K | V | Q (| gate)] of one token
    int rc = dispatch_vpl(width, [&](auto vpl) {
        hipLaunchKernelGGL((rmsnorm_rope_kernel<decltype(vpl)::value>), dim3((Sl + 3) / 4, ns), dim3(256), 0, (hipStream_t)stream, a);
        return FVK_OK;
    });
    FVK_CHECK(rc == FVK_OK, rc, "fvk_qkv_norm_rope_pack_bf16: unsupported width=%d", width);
    FVK_LAUNCH_CHECK();
    return FVK_OK;
}

extern "C" int fvk_v_transpose_bf16(const void* v, void* vt, int B, int S, int H, int D, long in_stride,
                                    long in_batch_stride, long in_head_stride, int S_pad, void* stream) {
    FVK_CHECK(v && vt, FVK_ERR_ARG, "fvk_v_transpose_bf16: null pointer");
    FVK_CHECK(D == 128, FVK_ERR_ARG, "fvk_v_transpose_bf16: head_dim %d != 128", D);
    FVK_CHECK(S_pad % 64 == 0 && S_pad >= S && in_stride % 8 == 0 && in_head_stride % 8 == 0, FVK_ERR_ARG, "fvk_v_transpose_bf16: S_pad=%d S=%d", S_pad, S);
    if (B <= 0 || S_pad == 0) return FVK_OK;
    hipLaunchKernelGGL(v_transpose_kernel, dim3(S_pad / 64, H, B), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)v,
                       (bf16_t*)vt, S, H, in_stride, in_batch_stride, in_head_stride, S_pad, (const int32_t*)nullptr);
    FVK_LAUNCH_CHECK();
    return FVK_OK;
}

extern "C" int fvk_v_transpose_gather_bf16(const void* v, void* vt, const int32_t* src_rows, int B, int S, int H, int D, long in_stride,
                                           long in_batch_stride, long in_head_stride, int S_pad, void* stream) {
    FVK_CHECK(v && vt && src_rows, FVK_ERR_ARG, "fvk_v_transpose_gather_bf16: null pointer");
    FVK_CHECK(D == 128, FVK_ERR_ARG, "fvk_v_transpose_gather_bf16: head_dim %d != 128", D);
    FVK_CHECK(S_pad % 64 == 0 && S > 0 && in_stride % 8 == 0 && in_head_stride % 8 == 0, FVK_ERR_ARG, "fvk_v_transpose_gather_bf16: S_pad=%d S=%d", S_pad, S);
    if (B <= 0 || S_pad == 0) return FVK_OK;
    hipLaunchKernelGGL(v_transpose_kernel, dim3(S_pad / 64, H, B), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)v,
                       (bf16_t*)vt, S, H, in_stride, in_batch_stride, in_head_stride, S_pad, src_rows);
    FVK_LAUNCH_CHECK();
    return FVK_OK;
}
