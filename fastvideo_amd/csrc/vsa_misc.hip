// VSA coarse stage (block mean, exact top-k mask, index compaction, combine), tile/untile row gathers,
// and the small patch/time-embedding glue kernels of the Wan DiT forward.  All HBM-bound byte/index work:
// 16-byte coalesced accesses, no MFMA.  Integer outputs are bit-exact restatements of the reference.
#pragma clang fp contract(off)
#include <math.h>

#include <vector>

#include "gemm_common.h"

// ------------------------------------------------------------------------------------------------
// host-side VSA metadata (pure integer; ref: fastvideo/attention/backends/video_sparse_attn.py:31-114, 226)
// ------------------------------------------------------------------------------------------------
extern "C" int fvk_vsa_build_metadata_host(int T, int H, int W, int tt, int th, int tw, int32_t* perm, int32_t* rev,
                                           int32_t* vbs, int32_t* non_pad, int32_t* untile) {
    FVK_CHECK(T > 0 && H > 0 && W > 0 && tt > 0 && th > 0 && tw > 0, FVK_ERR_ARG, "fvk_vsa_build_metadata_host: bad shape");
    const int nt = (T + tt - 1) / tt, nh = (H + th - 1) / th, nw = (W + tw - 1) / tw;
    const int n = T * H * W, blk = tt * th * tw;
    std::vector<int32_t> p(n), r(n), np_(n), sizes((size_t)nt * nh * nw);
    int pos = 0, tile = 0;
    for (int a = 0; a < nt; ++a)
        for (int b = 0; b < nh; ++b)
            for (int c = 0; c < nw; ++c, ++tile) {
                const int t1 = (a * tt + tt < T) ? a * tt + tt : T;
                const int h1 = (b * th + th < H) ? b * th + th : H;
                const int w1 = (c * tw + tw < W) ? c * tw + tw : W;
                int cnt = 0;
                for (int t = a * tt; t < t1; ++t)
                    for (int h = b * th; h < h1; ++h)
                        for (int w = c * tw; w < w1; ++w, ++cnt) {
                            const int raster = (t * H + h) * W + w;
                            p[pos] = raster;
                            r[raster] = pos;
                            np_[pos] = tile * blk + cnt;
                            ++pos;
                        }
                sizes[tile] = cnt;
            }
    for (int i = 0; i < n; ++i) {
        if (perm) perm[i] = p[i];
        if (rev) rev[i] = r[i];
        if (non_pad) non_pad[i] = np_[i];
        if (untile) untile[i] = np_[r[i]];
    }
    if (vbs)
        for (size_t i = 0; i < sizes.size(); ++i) vbs[i] = sizes[i];
    return FVK_OK;
}

namespace {

// dst[b, di[i], :] = src[b, si[i], :]; one 16-B chunk per thread.
__global__ __launch_bounds__(256) void gather_rows_kernel(const bf16_t* src, bf16_t* dst, const int32_t* si, const int32_t* di,
                                                          int n, int chunks_per_row, long sbs, long dbs, long srs, long drs) {
    const int b = blockIdx.y;
    const long total = (long)n * chunks_per_row;
    for (long c = blockIdx.x * 256L + threadIdx.x; c < total; c += (long)gridDim.x * 256L) {
        const int i = (int)(c / chunks_per_row), ch = (int)(c % chunks_per_row);
        const long s = si ? si[i] : i, d = di ? di[i] : i;
        st_bf16x8(dst + b * dbs + d * drs + ch * 8, ld_bf16x8(src + b * sbs + s * srs + ch * 8));
    }
}

// one workgroup per (block, head, batch): fp32 sum of `block` rows of D=128, / vbs, -> bf16.
// src_rows (optional, int32 [n_blocks * block]): row p of the tile-major order is row src_rows[p] of x (negative: a padding row = zeros) — the
// tile gather folded in (same rows, same summation order: bit-identical to the mean of the gathered copy).
__global__ __launch_bounds__(256) void block_mean_kernel(const bf16_t* x, bf16_t* out, const int32_t* vbs, int H, int n_blocks,
                                                         int block, long x_bs, long x_ss, long x_hs, const int32_t* src_rows) {
    __shared__ float part[16][128];
    const int blk = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    const int ch = threadIdx.x & 15, rg = threadIdx.x >> 4;
    const bf16_t* base = x + b * x_bs + h * x_hs + (src_rows ? 0L : (long)blk * block * x_ss) + ch * 8;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int r = rg; r < block; r += 16) {
        const int sr = src_rows ? src_rows[blk * block + r] : r;
        if (sr < 0) continue;
        bf16x8 v = ld_bf16x8(base + (long)sr * x_ss);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += (float)v[j];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) part[rg][ch * 8 + j] = acc[j];
    __syncthreads();
    if (threadIdx.x < 128) {
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) s += part[r][threadIdx.x];
        out[(((long)b * H + h) * n_blocks + blk) * 128 + threadIdx.x] = (bf16_t)__fdiv_rn(s, (float)vbs[blk]);
    }
}

__device__ __forceinline__ int block_sum_int(int v, int* red) {
    // 256 threads: wave reduce then 4-way LDS
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}
__device__ __forceinline__ float block_min_f(float v, float* red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return fminf(fminf(red[0], red[1]), fminf(red[2], red[3]));
}
__device__ __forceinline__ float block_max_f(float v, float* red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// Exact top-k mask per row — operation-for-operation restatement of _fused_topk_mask_kernel
// (fastvideo_kernel/triton_kernels/fused_compress_topk.py:211-277): 32 fp32 bisection steps on the threshold,
// then "> thr" plus the first (topk - n_above) entries "== thr" in index order.
template <int VPT>
__global__ __launch_bounds__(256) void topk_mask_kernel(const void* scores, int is_fp32, uint8_t* mask, int n, int topk) {
    __shared__ float redf[4];
    __shared__ int redi[4];
    __shared__ int seg_cnt[256];
    const long row = blockIdx.x;
    const int tid = threadIdx.x;
    // thread t owns the contiguous segment [t*VPT, (t+1)*VPT)
    float v[VPT];
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
        const int idx = tid * VPT + i;
        if (idx < n)
            v[i] = is_fp32 ? ((const float*)scores)[row * n + idx] : (float)((const bf16_t*)scores)[row * n + idx];
        else
            v[i] = -INFINITY;
    }
    float lo = INFINITY, hi = -INFINITY;
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
        const bool ok = tid * VPT + i < n;
        if (ok && v[i] > -INFINITY) lo = fminf(lo, v[i]);
        if (ok) hi = fmaxf(hi, v[i]);
    }
    lo = block_min_f(lo, redf);
    hi = block_max_f(hi, redf);
    lo = fminf(lo, hi);
    for (int it = 0; it < 32; ++it) {
        const float mid = (lo + hi) * 0.5f;
        int c = 0;
#pragma unroll
        for (int i = 0; i < VPT; ++i) c += (tid * VPT + i < n && v[i] >= mid) ? 1 : 0;
        c = block_sum_int(c, redi);
        if (c >= topk) lo = mid; else hi = mid;
    }
    const float thr = lo;
    int n_above = 0, n_at = 0;
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
        const bool ok = tid * VPT + i < n;
        n_above += (ok && v[i] > thr) ? 1 : 0;
        n_at += (ok && v[i] == thr) ? 1 : 0;
    }
    n_above = block_sum_int(n_above, redi);
    const int need = topk - n_above;
    seg_cnt[tid] = n_at;
    __syncthreads();
    int before = 0;  // number of "== thr" entries in lower-index segments
    for (int t = 0; t < tid; ++t) before += seg_cnt[t];
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
        const int idx = tid * VPT + i;
        if (idx < n) {
            bool sel = v[i] > thr;
            if (v[i] == thr) {
                ++before;  // inclusive cumulative count, as tl.cumsum
                sel = before <= need;
            }
            mask[row * n + idx] = sel ? 1 : 0;
        }
    }
}

// Same algorithm with ONE WAVE per row (4 rows per workgroup): lane l owns the contiguous segment [l*VPT, (l+1)*VPT), every count is a
// wave reduction (DPP shuffles, no LDS, no barrier) — the block-per-row form above spends its time in 2 x 34 workgroup barriers per row
// (103 us per layer at 12 x 624 rows of 624; 25 920 rows of 2160 at 129f x 720p).  Identical comparisons in identical fp32 arithmetic,
// identical tie rule => identical masks.
__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
template <int VPT>
__global__ __launch_bounds__(256) void topk_mask_wave_kernel(const void* scores, int is_fp32, uint8_t* mask, int rows, int n, int topk) {
    const int lane = threadIdx.x & 63;
    const long row = blockIdx.x * 4L + (threadIdx.x >> 6);
    if (row >= rows) return;
    float v[VPT];
    float lo = INFINITY, hi = -INFINITY;
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
        const int idx = lane * VPT + i;
        if (idx < n) {
            v[i] = is_fp32 ? ((const float*)scores)[row * n + idx] : (float)((const bf16_t*)scores)[row * n + idx];
            if (v[i] > -INFINITY) lo = fminf(lo, v[i]);
            hi = fmaxf(hi, v[i]);
        } else {
            v[i] = -INFINITY;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        lo = fminf(lo, __shfl_xor(lo, o, 64));
        hi = fmaxf(hi, __shfl_xor(hi, o, 64));
    }
    lo = fminf(lo, hi);
    for (int it = 0; it < 32; ++it) {
        const float mid = (lo + hi) * 0.5f;
        int c = 0;
#pragma unroll
        for (int i = 0; i < VPT; ++i) c += (lane * VPT + i < n && v[i] >= mid) ? 1 : 0;
        c = wave_sum_i(c);
        if (c >= topk) lo = mid; else hi = mid;
    }
    const float thr = lo;
    int n_above = 0, n_at = 0;
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
        const bool ok = lane * VPT + i < n;
        n_above += (ok && v[i] > thr) ? 1 : 0;
        n_at += (ok && v[i] == thr) ? 1 : 0;
    }
    n_above = wave_sum_i(n_above);
    const int need = topk - n_above;
    int incl = n_at;  // inclusive prefix over lanes of the "== thr" counts
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t;
    }
    int before = incl - n_at;
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
        const int idx = lane * VPT + i;
        if (idx < n) {
            bool sel = v[i] > thr;
            if (v[i] == thr) {
                ++before;  // inclusive cumulative count, as tl.cumsum
                sel = before <= need;
            }
            mask[row * n + idx] = sel ? 1 : 0;
        }
    }
}

// ascending compaction, one wave per row (ballot + popcount prefix).
__global__ __launch_bounds__(256) void map_to_index_kernel(const uint8_t* mask, int32_t* idx, int32_t* num, int rows, int n) {
    const int lane = threadIdx.x & 63;
    const long row = blockIdx.x * 4L + (threadIdx.x >> 6);
    if (row >= rows) return;
    int cnt = 0;
    for (int base = 0; base < n; base += 64) {
        const int i = base + lane;
        const bool on = i < n && mask[row * n + i] != 0;
        const unsigned long long bal = __ballot(on);
        if (on) idx[row * n + cnt + __popcll(bal & ((1ull << lane) - 1ull))] = i;
        cnt += __popcll(bal);
    }
    for (int i = cnt + lane; i < n; i += 64) idx[row * n + i] = 0;
    if (lane == 0) num[row] = cnt;
}

__global__ __launch_bounds__(256) void softmax_rows_kernel(const bf16_t* in, bf16_t* out, int n) {
    __shared__ float redf[4];
    const long row = blockIdx.x;
    float mx = -INFINITY;
    for (int i = threadIdx.x; i < n; i += 256) mx = fmaxf(mx, (float)in[row * n + i]);
    mx = block_max_f(mx, redf);
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) s += expf((float)in[row * n + i] - mx);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) redf[threadIdx.x >> 6] = s;
    __syncthreads();
    s = redf[0] + redf[1] + redf[2] + redf[3];
    for (int i = threadIdx.x; i < n; i += 256) out[row * n + i] = (bf16_t)__fdiv_rn(expf((float)in[row * n + i] - mx), s);
}

// token_of_row (optional, int32 [S]): tile-major row s holds token token_of_row[s] (negative: a padding row, skipped); gate is then read and
// out written at the TOKEN's row with their own strides (g_* / o_*) — the gate's tile gather and the output's untile gather folded in.
__global__ __launch_bounds__(256) void vsa_combine_kernel(const bf16_t* out_c, const bf16_t* out_s, const bf16_t* gate, bf16_t* out,
                                                          int S, int H, int block, long bs, long ss, long hs, long total,
                                                          const int32_t* token_of_row, long g_bs, long g_ss, long g_hs, long o_bs,
                                                          long o_ss, long o_hs) {
    // one 16-B chunk (8 of D=128) per thread; chunk id -> (b, s, h, ch)
    for (long c = blockIdx.x * 256L + threadIdx.x; c < total; c += (long)gridDim.x * 256L) {
        const int ch = (int)(c & 15);
        long r = c >> 4;
        const int h = (int)(r % H); r /= H;
        const int s = (int)(r % S);
        const int b = (int)(r / S);
        const long off = b * bs + s * ss + h * hs + ch * 8;
        long goff = off, ooff = off;
        if (token_of_row) {
            const int tok = token_of_row[s];
            if (tok < 0) continue;
            goff = b * g_bs + tok * g_ss + h * g_hs + ch * 8;
            ooff = b * o_bs + tok * o_ss + h * o_hs + ch * 8;
        }
        const bf16x8 oc = ld_bf16x8(out_c + ((((long)b * H + h) * (S / block)) + s / block) * 128 + ch * 8);
        const bf16x8 os = ld_bf16x8(out_s + off);
        bf16x8 o;
        if (gate) {
            const bf16x8 g = ld_bf16x8(gate + goff);
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (bf16_t)(bf16_round_opaque((float)oc[j] * (float)g[j]) + (float)os[j]);
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (bf16_t)((float)oc[j] + (float)os[j]);
        }
        st_bf16x8(out + ooff, o);
    }
}

// latent [B,C,T,H,W] -> rows [B, S, C*pt*ph*pw] in Conv3d-weight element order (c, dt, dh, dw).
__global__ __launch_bounds__(256) void patchify_kernel(const bf16_t* x, bf16_t* out, int C, int T, int Hh, int W, int pt, int ph,
                                                       int pw, long total) {
    const int gt = T / pt, gh = Hh / ph, gw = W / pw, pe = C * pt * ph * pw;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256L) {
        int e = (int)(i % pe);
        long s = i / pe;
        const int dw = e % pw; e /= pw;
        const int dh = e % ph; e /= ph;
        const int dt = e % pt;
        const int c = e / pt;
        const int w = (int)(s % gw); s /= gw;
        const int h = (int)(s % gh); s /= gh;
        const int t = (int)(s % gt);
        const long b = s / gt;
        out[i] = x[(((b * C + c) * T + (t * pt + dt)) * Hh + (h * ph + dh)) * W + (w * pw + dw)];
    }
}
// rows [B, S, pt*ph*pw*C] (element order dt,dh,dw,c — wanvideo.py:761-764) -> latent [B,C,T,H,W]
__global__ __launch_bounds__(256) void unpatchify_kernel(const bf16_t* x, bf16_t* out, int C, int T, int Hh, int W, int pt, int ph,
                                                         int pw, long total) {
    const int gt = T / pt, gh = Hh / ph, gw = W / pw;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256L) {
        long r = i;  // index into out [B,C,T,H,W]
        const int wv = (int)(r % W); r /= W;
        const int hv = (int)(r % Hh); r /= Hh;
        const int tv = (int)(r % T); r /= T;
        const int c = (int)(r % C);
        const long b = r / C;
        const int t = tv / pt, dt = tv % pt, h = hv / ph, dh = hv % ph, w = wv / pw, dw = wv % pw;
        const long s = ((b * gt + t) * gh + h) * gw + w;
        out[i] = x[s * ((long)pt * ph * pw * C) + ((dt * ph + dh) * pw + dw) * C + c];
    }
}

__global__ void timestep_embedding_kernel(const float* t, bf16_t* out, int dim, float max_period) {
    const int b = blockIdx.x, half = dim / 2;
    for (int i = threadIdx.x; i < half; i += blockDim.x) {
        // freqs = exp(-log(max_period) * arange(half, fp32) / half)  (visual_embedding.py:150-152)
        const float f = expf(-logf(max_period) * (float)i / (float)half);
        const float arg = t[b] * f;
        out[(long)b * dim + i] = (bf16_t)cosf(arg);
        out[(long)b * dim + half + i] = (bf16_t)sinf(arg);
    }
}

__global__ __launch_bounds__(256) void silu_kernel(const bf16_t* x, bf16_t* y, long n_chunks) {
    for (long c = blockIdx.x * 256L + threadIdx.x; c < n_chunks; c += (long)gridDim.x * 256L) {
        bf16x8 v = ld_bf16x8(x + c * 8), o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float f = (float)v[j];
            o[j] = (bf16_t)(f / (1.0f + expf(-f)));
        }
        st_bf16x8(y + c * 8, o);
    }
}

inline int grid_for(long work_items) {
    long g = (work_items + 255) / 256;
    return (int)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}

}  // namespace

extern "C" int fvk_gather_rows_bf16(const void* src, void* dst, const int32_t* src_index, const int32_t* dst_index, int B, int n,
                                    int row_elems, long src_batch_stride, long dst_batch_stride, void* stream) {
    FVK_CHECK(src && dst, FVK_ERR_ARG, "fvk_gather_rows_bf16: null pointer");
    FVK_CHECK(row_elems > 0 && row_elems % 8 == 0, FVK_ERR_ARG, "fvk_gather_rows_bf16: row_elems=%d must be a multiple of 8", row_elems);
    if (B <= 0 || n <= 0) return FVK_OK;
    hipLaunchKernelGGL(gather_rows_kernel, dim3(grid_for((long)n * row_elems / 8), B), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)src, (bf16_t*)dst, src_index, dst_index, n, row_elems / 8, src_batch_stride,
                       dst_batch_stride, (long)row_elems, (long)row_elems);
    FVK_LAUNCH_CHECK();
    return FVK_OK;
}

extern "C" int fvk_gather_rows_strided_bf16(const void* src, void* dst, const int32_t* src_index, const int32_t* dst_index, int B, int n,
                                            int row_elems, long src_row_stride, long dst_row_stride, long src_batch_stride,
                                            long dst_batch_stride, void* stream) {
    FVK_CHECK(src && dst, FVK_ERR_ARG, "fvk_gather_rows_strided_bf16: null pointer");
    FVK_CHECK(row_elems > 0 && row_elems % 8 == 0 && src_row_stride % 8 == 0 && dst_row_stride % 8 == 0 && src_row_stride >= row_elems &&
                  dst_row_stride >= row_elems,
              FVK_ERR_ARG, "fvk_gather_rows_strided_bf16: row_elems=%d and the row strides must be multiples of 8, strides >= row", row_elems);
    FVK_CHECK((reinterpret_cast<uintptr_t>(src) & 15) == 0 && (reinterpret_cast<uintptr_t>(dst) & 15) == 0, FVK_ERR_ARG,
              "fvk_gather_rows_strided_bf16: 16-byte aligned pointers expected");
    if (B <= 0 || n <= 0) return FVK_OK;
    hipLaunchKernelGGL(gather_rows_kernel, dim3(grid_for((long)n * row_elems / 8), B), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)src, (bf16_t*)dst, src_index, dst_index, n, row_elems / 8, src_batch_stride,
                       dst_batch_stride, src_row_stride, dst_row_stride);
    FVK_LAUNCH_CHECK();
    return FVK_OK;
}

extern "C" int fvk_block_mean_bf16(const void* x, void* out, const int32_t* vbs, int B, int H, int n_blocks, int block, int D,
                                   long x_bs, long x_ss, long x_hs, void* stream) {
    FVK_CHECK(x && out && vbs, FVK_ERR_ARG, "fvk_block_mean_bf16: null pointer");
    FVK_CHECK(D == 128 && block > 0, FVK_ERR_ARG, "fvk_block_mean_bf16: D=%d must be 128", D);
    if (B <= 0 || H <= 0 || n_blocks <= 0) return FVK_OK;
    hipLaunchKernelGGL(block_mean_kernel, dim3(n_blocks, H, B), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (bf16_t*)out,
                       vbs, H, n_blocks, block, x_bs, x_ss, x_hs, (const int32_t*)nullptr);
    FVK_LAUNCH_CHECK();
    return FVK_OK;
}

extern "C" int fvk_block_mean_gather_bf16(const void* x, void* out, const int32_t* vbs, const int32_t* src_rows, int B, int H, int n_blocks,
                                          int block, int D, long x_bs, long x_ss, long x_hs, void* stream) {
    FVK_CHECK(x && out && vbs && src_rows, FVK_ERR_ARG, "fvk_block_mean_gather_bf16: null pointer");
    FVK_CHECK(D == 128 && block > 0, FVK_ERR_ARG, "fvk_block_mean_gather_bf16: D=%d must be 128", D);
    if (B <= 0 || H <= 0 || n_blocks <= 0) return FVK_OK;
    hipLaunchKernelGGL(block_mean_kernel, dim3(n_blocks, H, B), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (bf16_t*)out,
                       vbs, H, n_blocks, block, x_bs, x_ss, x_hs, src_rows);
    FVK_LAUNCH_CHECK();
    return FVK_OK;
}

extern "C" int fvk_topk_mask(const void* scores, int scores_is_fp32, uint8_t* mask, int rows, int n, int topk, void* stream) {
    FVK_CHECK(scores && mask, FVK_ERR_ARG, "fvk_topk_mask: null pointer");
    FVK_CHECK(n > 0 && n <= 8192 && topk >= 1, FVK_ERR_ARG, "fvk_topk_mask: n=%d (max 8192) topk=%d", n, topk);
    if (rows <= 0) return FVK_OK;
    if (topk > n) topk = n;
    hipStream_t s = (hipStream_t)stream;
    if (fvk::tunable(fvk::TUNE_VSA_IMPL) != 1) {  // shipped: one wave per row ("vsa_impl" 1 = the block-per-row kernel, measurement build)
        const int wv = (n + 63) / 64;
#define FVK_TOPKW_CASE(V)                                                                                                             \
    if (wv <= V) {                                                                                                                    \
        hipLaunchKernelGGL((topk_mask_wave_kernel<V>), dim3((rows + 3) / 4), dim3(256), 0, s, scores, scores_is_fp32, mask, rows, n, topk); \
        FVK_LAUNCH_CHECK();                                                                                                           \
        return FVK_OK;                                                                                                                \
    }
        FVK_TOPKW_CASE(4) FVK_TOPKW_CASE(8) FVK_TOPKW_CASE(16) FVK_TOPKW_CASE(32) FVK_TOPKW_CASE(64) FVK_TOPKW_CASE(128)
#undef FVK_TOPKW_CASE
    }
#if FVK_VARIANTS
    const int vpt = (n + 255) / 256;
#define FVK_TOPK_CASE(V)                                                                                             \
    if (vpt <= V) {                                                                                                  \
        hipLaunchKernelGGL((topk_mask_kernel<V>), dim3(rows), dim3(256), 0, s, scores, scores_is_fp32, mask, n, topk); \
        FVK_LAUNCH_CHECK();                                                                                          \
        return FVK_OK;                                                                                               \
    }
    FVK_TOPK_CASE(1) FVK_TOPK_CASE(2) FVK_TOPK_CASE(4) FVK_TOPK_CASE(8) FVK_TOPK_CASE(16) FVK_TOPK_CASE(32)
#undef FVK_TOPK_CASE
#endif
    return FVK_ERR_ARG;
}

extern "C" int fvk_map_to_index(const uint8_t* mask, int32_t* idx, int32_t* num, int rows, int n, void* stream) {
    FVK_CHECK(mask && idx && num && n > 0, FVK_ERR_ARG, "fvk_map_to_index: null pointer / n");
    if (rows <= 0) return FVK_OK;
    hipLaunchKernelGGL(map_to_index_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, mask, idx, num, rows, n);
    FVK_LAUNCH_CHECK();
    return FVK_OK;
}

// Union of the KV lists of two neighbouring query blocks (fvk_attn_block_sparse_union_bf16): one thread per (batch * head, pair) merges the two
// ASCENDING lists of blocks 2p and 2p + 1 (fvk_map_to_index's order) into one ascending list of packed entries
//     block id | valid keys (kv_block_sizes[id]) << 22 | halves << 29        (halves: bit 0 = block 2p selected it, bit 1 = block 2p + 1)
// A missing second block (odd block count) contributes nothing.  Integer work: bit-exact against a host merge (tests/test_gpu_kernels.py).
__global__ __launch_bounds__(256) void vsa_union_lists_kernel(const int32_t* q2k_idx, const int32_t* q2k_num, const int32_t* kv_block_sizes, int32_t* u_idx,
                                                              int32_t* u_num, long rows, int nq, int max_kv) {
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    const int npair = (nq + 1) / 2;
    if (t >= rows * npair) return;
    const long bh = t / npair;
    const int p = (int)(t - bh * npair);
    const int qa = 2 * p, qb = 2 * p + 1;
    const int32_t* la = q2k_idx + (bh * nq + qa) * max_kv;
    const int32_t* lb = q2k_idx + (bh * nq + (qb < nq ? qb : qa)) * max_kv;
    int na = q2k_num[bh * nq + qa], nb = qb < nq ? q2k_num[bh * nq + qb] : 0;
    na = na < max_kv ? na : max_kv;
    nb = nb < max_kv ? nb : max_kv;
    int32_t* out = u_idx + t * (2L * max_kv);
    int i = 0, j = 0, n = 0;
    while (i < na || j < nb) {
        const int a_ = i < na ? la[i] : 0x7fffffff, b_ = j < nb ? lb[j] : 0x7fffffff;
        const int id = a_ < b_ ? a_ : b_;
        const int halves = (a_ == id ? 1 : 0) | (b_ == id ? 2 : 0);
        i += a_ == id;
        j += b_ == id;
        out[n++] = id | (kv_block_sizes[id] << 22) | (halves << 29);
    }
    u_num[t] = n;
}

extern "C" int fvk_vsa_union_lists(const int32_t* q2k_idx, const int32_t* q2k_num, const int32_t* kv_block_sizes, int32_t* u_idx, int32_t* u_num,
                                   int rows, int nq, int max_kv, void* stream) {
    FVK_CHECK(q2k_idx && q2k_num && kv_block_sizes && u_idx && u_num, FVK_ERR_ARG, "fvk_vsa_union_lists: null pointer");
    FVK_CHECK(nq > 0 && max_kv > 0 && max_kv <= 2048, FVK_ERR_ARG, "fvk_vsa_union_lists: nq=%d max_kv=%d (1..2048: a merged list has at most 4096 entries)", nq, max_kv);
    if (rows <= 0) return FVK_OK;
    const long threads = (long)rows * ((nq + 1) / 2);
    hipLaunchKernelGGL(vsa_union_lists_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, q2k_idx, q2k_num,
                       kv_block_sizes, u_idx, u_num, (long)rows, nq, max_kv);
    FVK_LAUNCH_CHECK();
    return FVK_OK;
}

extern "C" int fvk_softmax_rows_bf16(const void* in, void* out, int rows, int n, void* stream) {
    FVK_CHECK(in && out && n > 0, FVK_ERR_ARG, "fvk_softmax_rows_bf16: null pointer / n");
    if (rows <= 0) return FVK_OK;
    hipLaunchKernelGGL(softmax_rows_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)in, (bf16_t*)out, n);
    FVK_LAUNCH_CHECK();
    return FVK_OK;
}

extern "C" int fvk_vsa_combine_bf16(const void* out_c, const void* out_s, const void* gate, void* out, int B, int S, int H, int D,
                                    int block, long bs, long ss, long hs, void* stream) {
    FVK_CHECK(out_c && out_s && out, FVK_ERR_ARG, "fvk_vsa_combine_bf16: null pointer");
    FVK_CHECK(D == 128 && block > 0 && S % block == 0, FVK_ERR_ARG, "fvk_vsa_combine_bf16: D=%d S=%d block=%d", D, S, block);
    const long total = (long)B * S * H * 16;
    if (total <= 0) return FVK_OK;
    hipLaunchKernelGGL(vsa_combine_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)out_c,
                       (const bf16_t*)out_s, (const bf16_t*)gate, (bf16_t*)out, S, H, block, bs, ss, hs, total, (const int32_t*)nullptr, 0L, 0L,
                       0L, 0L, 0L, 0L);
    FVK_LAUNCH_CHECK();
    return FVK_OK;
}

extern "C" int fvk_vsa_combine_scatter_bf16(const void* out_c, const void* out_s, const void* gate, void* out, const int32_t* token_of_row,
                                            int B, int S, int H, int D, int block, long bs, long ss, long hs, long g_bs, long g_ss,
                                            long g_hs, long o_bs, long o_ss, long o_hs, void* stream) {
    FVK_CHECK(out_c && out_s && out && token_of_row, FVK_ERR_ARG, "fvk_vsa_combine_scatter_bf16: null pointer");
    FVK_CHECK(D == 128 && block > 0 && S % block == 0, FVK_ERR_ARG, "fvk_vsa_combine_scatter_bf16: D=%d S=%d block=%d", D, S, block);
    FVK_CHECK((g_bs | g_ss | g_hs | o_bs | o_ss | o_hs) % 8 == 0, FVK_ERR_ARG, "fvk_vsa_combine_scatter_bf16: strides must keep 16-byte alignment");
    const long total = (long)B * S * H * 16;
    if (total <= 0) return FVK_OK;
    hipLaunchKernelGGL(vsa_combine_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)out_c,
                       (const bf16_t*)out_s, (const bf16_t*)gate, (bf16_t*)out, S, H, block, bs, ss, hs, total, token_of_row, g_bs, g_ss,
                       g_hs, o_bs, o_ss, o_hs);
    FVK_LAUNCH_CHECK();
    return FVK_OK;
}

extern "C" int fvk_patchify_bf16(const void* latent, void* out, int B, int C, int T, int Hh, int W, int pt, int ph, int pw,
                                 void* stream) {
    FVK_CHECK(latent && out, FVK_ERR_ARG, "fvk_patchify_bf16: null pointer");
    FVK_CHECK(pt > 0 && ph > 0 && pw > 0 && T % pt == 0 && Hh % ph == 0 && W % pw == 0, FVK_ERR_ARG,
              "fvk_patchify_bf16: latent %dx%dx%d not divisible by patch %dx%dx%d", T, Hh, W, pt, ph, pw);
    const long total = (long)B * C * T * Hh * W;
    if (total <= 0) return FVK_OK;
    hipLaunchKernelGGL(patchify_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)latent,
                       (bf16_t*)out, C, T, Hh, W, pt, ph, pw, total);
    FVK_LAUNCH_CHECK();
    return FVK_OK;
}

extern "C" int fvk_unpatchify_bf16(const void* x, void* latent, int B, int C, int T, int Hh, int W, int pt, int ph, int pw,
                                   void* stream) {
    FVK_CHECK(x && latent, FVK_ERR_ARG, "fvk_unpatchify_bf16: null pointer");
    FVK_CHECK(pt > 0 && ph > 0 && pw > 0 && T % pt == 0 && Hh % ph == 0 && W % pw == 0, FVK_ERR_ARG,
              "fvk_unpatchify_bf16: latent %dx%dx%d not divisible by patch %dx%dx%d", T, Hh, W, pt, ph, pw);
    const long total = (long)B * C * T * Hh * W;
    if (total <= 0) return FVK_OK;
    hipLaunchKernelGGL(unpatchify_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x,
                       (bf16_t*)latent, C, T, Hh, W, pt, ph, pw, total);
    FVK_LAUNCH_CHECK();
    return FVK_OK;
}

extern "C" int fvk_timestep_embedding_bf16(const float* t, void* out, int B, int dim, float max_period, void* stream) {
    FVK_CHECK(t && out && dim > 0 && dim % 2 == 0, FVK_ERR_ARG, "fvk_timestep_embedding_bf16: dim=%d must be even", dim);
    if (B <= 0) return FVK_OK;
    hipLaunchKernelGGL(timestep_embedding_kernel, dim3(B), dim3(128), 0, (hipStream_t)stream, t, (bf16_t*)out, dim, max_period);
    FVK_LAUNCH_CHECK();
    return FVK_OK;
}

extern "C" int fvk_silu_bf16(const void* x, void* y, long n, void* stream) {
    FVK_CHECK(x && y && n % 8 == 0, FVK_ERR_ARG, "fvk_silu_bf16: n=%ld must be a multiple of 8", n);
    if (n <= 0) return FVK_OK;
    hipLaunchKernelGGL(silu_kernel, dim3(grid_for(n / 8)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (bf16_t*)y, n / 8);
    FVK_LAUNCH_CHECK();
    return FVK_OK;
}
