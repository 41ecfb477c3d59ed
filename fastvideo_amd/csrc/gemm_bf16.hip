// bf16 GEMM  out[M,N] = epilogue( x[M,K] · w[N,K]^T + bias )  on gfx950 MFMA (v_mfma_f32_32x32x16_bf16).
//
// Both operands are K-contiguous ("B^T input"), which is exactly the MFMA A/B fragment order (8 consecutive
// k per lane), so fragments are single ds_read_b128.
//
// Tile 128(M) x 128(N) x 64(K), 256 threads = 4 waves as 2x2, wave tile 64x64 = 2x2 MFMA blocks of 32x32
// (64 fp32 accumulator VGPRs per lane).  LDS: 2 stages x (A 16 KiB + B 16 KiB) = 64 KiB -> 2 workgroups/CU.
// Staging is register-staged and software-pipelined ("issue early / write late", guide T14): the global
// loads of K-tile t+1 are issued before the MFMAs of tile t and written to the other LDS stage after them;
// one barrier per K-tile.  LDS rows are 128 B (64 bf16); the 16-B chunk index is XOR-swizzled with
// (row>>1)&7 so that a fragment read (32 rows x same chunk) touches 16 distinct 16-B slots per 16-lane
// group: conflict-free ds_read_b128 (guide T2 / Guideline 4).
// Workgroup ids are remapped so each XCD (private L2) owns a contiguous range of output tiles (guide T1).
//
// Epilogue variants follow the reference's rounding points: y = bf16(acc + bias) first, then the activation /
// gated residual on float(y), then one more rounding.
#include "gemm_common.h"
#include "gemm_epilogue.h"

namespace {

using fvk::GemmArgs;

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int STAGE_BYTES = (BM + BN) * BK * 2;  // 32 KiB


__device__ __forceinline__ int swz_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_bf16_kernel(GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, hi = lane >> 5;

    // XCD-aware, bijective remap of the linear workgroup id (8 XCDs, block b runs on XCD b % 8).
    int tile_id;
    {
        const int nwg = gridDim.x, bid = blockIdx.x;
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
        tile_id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int pid_m = tile_id / a.ntn, pid_n = tile_id % a.ntn;
    const int m0 = pid_m * BM, n0 = pid_n * BN;
    a.x += blockIdx.y * a.x_bstride;
    a.w += blockIdx.y * a.w_bstride;
    a.out += blockIdx.y * a.out_bstride;

    // staging assignment: 4 A chunks + 4 B chunks (16 B each) per thread per K-tile
    const bf16_t* ga[4];
    const bf16_t* gb[4];
    int soff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = tid + 256 * i;
        const int row = c >> 3, ch = c & 7;
        int mr = m0 + row; mr = mr < a.M ? mr : a.M - 1;
        int nr = n0 + row; nr = nr < a.N ? nr : a.N - 1;
        ga[i] = a.x + (long)mr * a.lda + ch * 8;
        gb[i] = a.w + (long)nr * a.K + ch * 8;
        soff[i] = swz_off(row, ch);
    }
    // fragment read offsets (bytes within a stage); B tile lives BM*BK*2 bytes after A
    int aoff[2][4], boff[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            aoff[i][ks] = swz_off(wm * 64 + i * 32 + l31, 2 * ks + hi);
            boff[i][ks] = BM * BK * 2 + swz_off(wn * 64 + i * 32 + l31, 2 * ks + hi);
        }

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nt = a.K / BK;
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    u32x4 ra[4], rb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        ra[i] = *reinterpret_cast<const u32x4*>(ga[i]);
        rb[i] = *reinterpret_cast<const u32x4*>(gb[i]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        *reinterpret_cast<u32x4*>(smem + soff[i]) = ra[i];
        *reinterpret_cast<u32x4*>(smem + BM * BK * 2 + soff[i]) = rb[i];
    }
    __syncthreads();

    for (int t = 0; t < nt; ++t) {
        unsigned char* cur = smem + (t & 1) * STAGE_BYTES;
        unsigned char* nxt = smem + ((t + 1) & 1) * STAGE_BYTES;
        const bool more = (t + 1) < nt;
        if (more) {
            const int koff = (t + 1) * BK;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                ra[i] = *reinterpret_cast<const u32x4*>(ga[i] + koff);
                rb[i] = *reinterpret_cast<const u32x4*>(gb[i] + koff);
            }
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            bf16x8 fa[2], fb[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                fa[i] = *reinterpret_cast<const bf16x8*>(cur + aoff[i][ks]);
                fb[i] = *reinterpret_cast<const bf16x8*>(cur + boff[i][ks]);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
        if (more) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                *reinterpret_cast<u32x4*>(nxt + soff[i]) = ra[i];
                *reinterpret_cast<u32x4*>(nxt + BM * BK * 2 + soff[i]) = rb[i];
            }
        }
        __syncthreads();
    }

    // epilogue: lane owns column n, 16 rows per 32x32 block: m = (r&3) + 8*(r>>2) + 4*hi
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n0 + wn * 64 + j * 32 + l31;
        if (n >= a.N) continue;
        const float bn = a.bias ? (float)a.bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                if (m >= a.M) continue;
                float y = bf16_round(acc[i][j][r] + bn);
                if (EPI == FVK_EPI_GELU_TANH) {
                    y = gelu_tanh_f32(y);
                } else if (EPI == FVK_EPI_SILU) {
                    y = silu_f32(y);
                } else if (EPI == FVK_EPI_DIV) {
                    y = __fdiv_rn(y, a.epi_scalar);
                } else if (EPI == FVK_EPI_RESIDUAL_GATE) {
                    const float res = (float)a.residual[(long)m * a.ldc + n];
                    const float g = a.gate ? a.gate[(long)(m / a.rows_per_batch) * a.N + n] : 1.0f;
                    y = fvk::mul_then_add(res, y, g);  // two roundings (gemm_epilogue.h)
                }
                a.out[(long)m * a.ldc + n] = (bf16_t)y;
            }
        }
    }
}

template <int EPI>
int launch(const GemmArgs& a, int batch, hipStream_t s) {
    static FvkLdsConfigured configured;
    if (int rc = fvk_config_lds(configured, (const void*)gemm_bf16_kernel<EPI>, 2 * STAGE_BYTES, "fvk_gemm_bf16")) return rc;
    hipLaunchKernelGGL((gemm_bf16_kernel<EPI>), dim3(a.ntm * a.ntn, batch), dim3(256), 2 * STAGE_BYTES, s, a);
    FVK_LAUNCH_CHECK();
    return FVK_OK;
}

}  // namespace

static int gemm_impl(const void* x, const void* w, const void* bias, void* out, int M, int N, int K, long lda, long ldc,
                     int epilogue, const void* residual, const float* gate, int rows_per_batch, float epi_scalar, int batch,
                     long x_bstride, long w_bstride, long out_bstride, void* stream) {
    FVK_CHECK(x && w && out, FVK_ERR_ARG, "fvk_gemm_bf16: null pointer");
    FVK_CHECK(K > 0 && K % 32 == 0, FVK_ERR_ARG, "fvk_gemm_bf16: K=%d must be a positive multiple of 32", K);
    FVK_CHECK(N > 0 && lda >= K && ldc >= N && lda % 8 == 0, FVK_ERR_ARG, "fvk_gemm_bf16: bad N=%d lda=%ld ldc=%ld", N, lda, ldc);
    FVK_CHECK(epilogue >= 0 && epilogue <= 4, FVK_ERR_ARG, "fvk_gemm_bf16: unknown epilogue %d", epilogue);
    FVK_CHECK(epilogue != FVK_EPI_RESIDUAL_GATE || (residual && rows_per_batch > 0), FVK_ERR_ARG,
              "fvk_gemm_bf16: residual epilogue needs residual and rows_per_batch");
    FVK_CHECK(epilogue != FVK_EPI_DIV || epi_scalar != 0.f, FVK_ERR_ARG, "fvk_gemm_bf16: DIV epilogue needs a non-zero scalar");
    FVK_CHECK(batch >= 1 && batch <= 65535 && x_bstride % 8 == 0 && w_bstride % 8 == 0, FVK_ERR_ARG, "fvk_gemm_bf16: bad batch=%d / strides", batch);
    if (M <= 0) return FVK_OK;
    GemmArgs a{(const bf16_t*)x, (const bf16_t*)w, (const bf16_t*)bias, (bf16_t*)out, (const bf16_t*)residual, gate,
               M, N, K, lda, ldc, rows_per_batch > 0 ? rows_per_batch : M, (M + BM - 1) / BM, (N + BN - 1) / BN,
               epi_scalar, x_bstride, w_bstride, out_bstride};
    hipStream_t s = (hipStream_t)stream;
    // token-axis GEMMs go to the 256x256 LDS-DMA kernels: gemm_w1.hip (K % 128 == 0, the default), gemm_ph.hip (K % 64 == 0) or gemm_pp.hip (K % 64 != 0, or forced
    // with gemm_impl 2 / 3); this 128x128 kernel keeps the small / odd shapes (gemm_impl 1 forces it)
    const int impl = fvk::tunable(fvk::TUNE_GEMM_IMPL);  // constant 0 in the product library
    if (impl != 1 && fvk::gemm_pp_eligible(a)) {
        // gemm_w1.hip (four 128 x 128 waves, 16x16x32 MFMAs) where K is a whole number of 128-element double steps; gemm_impl 4 forces gemm_ph
        // few-tile problems (a rank's N = 1536 projections at SP = 8): the 256 x 128 tile form of the same kernel, byte-identical results
        if ((impl == 0 && fvk::gemm_w1n_eligible(a, epilogue)) ||
            (impl == 6 && fvk::gemm_w1_eligible(a) && !(epilogue == FVK_EPI_RESIDUAL_GATE && gate && a.rows_per_batch < 128)))
            return fvk::gemm_w1n_launch(a, epilogue, batch, s);
        if ((impl == 0 || impl == 14 || (impl & 7) == 5) && fvk::gemm_w1_eligible(a)) return fvk::gemm_w1_launch(a, epilogue, batch, s);
        if ((impl == 0 || (impl & 7) == 4) && fvk::gemm_ph_eligible(a)) return fvk::gemm_ph_launch(a, epilogue, batch, s);
        return fvk::gemm_pp_launch(a, epilogue, batch, s);
    }
    FVK_CHECK(K % BK == 0, FVK_ERR_ARG, "fvk_gemm_bf16: K=%d must be a multiple of %d for this shape (M=%d N=%d)", K, BK, M, N);
    switch (epilogue) {
        case FVK_EPI_NONE: return launch<FVK_EPI_NONE>(a, batch, s);
        case FVK_EPI_GELU_TANH: return launch<FVK_EPI_GELU_TANH>(a, batch, s);
        case FVK_EPI_SILU: return launch<FVK_EPI_SILU>(a, batch, s);
        case FVK_EPI_DIV: return launch<FVK_EPI_DIV>(a, batch, s);
        default: return launch<FVK_EPI_RESIDUAL_GATE>(a, batch, s);
    }
}

extern "C" int fvk_gemm_bf16(const void* x, const void* w, const void* bias, void* out, int M, int N, int K, long lda,
                             long ldc, int epilogue, const void* residual, const float* gate, int rows_per_batch,
                             void* stream) {
    return gemm_impl(x, w, bias, out, M, N, K, lda, ldc, epilogue, residual, gate, rows_per_batch, 1.0f, 1, 0, 0, 0, stream);
}

// V^T = Wv · X^T + bias, written in the attention kernels' V^T layout (include/fvk_amd.h).  A plain GEMM with the operand roles exchanged:
// "x" = the V projection's weight rows (M = d output channels), "w" = the token rows (N = S), out row stride S_pad — on gemm_w1.hip, whose w
// staging applies the key permutation for free (a per-lane source-row term of the LDS-DMA address) and whose direct epilogue adds the
// per-row bias and zeroes the padding columns.  The MFMA sees the same two fragments as the fused QKV GEMM, as B / A instead of A / B.
extern "C" int fvk_gemm_vt_bf16(const void* wv, const void* x, const void* bias, void* vt, int B, int S, int d, int K, long ldx,
                                long x_bstride, int S_pad, void* stream) {
    FVK_CHECK(wv && x && vt, FVK_ERR_ARG, "fvk_gemm_vt_bf16: null pointer");
    FVK_CHECK(B >= 1 && B <= 65535 && S > 0 && d > 0 && d % 128 == 0, FVK_ERR_ARG, "fvk_gemm_vt_bf16: bad B=%d S=%d d=%d (d: whole 128-channel heads)", B, S, d);
    FVK_CHECK(S % 8 == 0 && S_pad % 64 == 0 && S_pad >= S && S_pad <= (S + 255) / 256 * 256, FVK_ERR_ARG,
              "fvk_gemm_vt_bf16: S=%d must be a multiple of 8 and S_pad=%d a multiple of 64 in [S, round_up(S, 256)]", S, S_pad);
    FVK_CHECK(K > 0 && ldx >= K && ldx % 8 == 0 && x_bstride % 8 == 0, FVK_ERR_ARG, "fvk_gemm_vt_bf16: bad K=%d ldx=%ld x_bstride=%ld", K, ldx, x_bstride);
    GemmArgs a{(const bf16_t*)wv, (const bf16_t*)x, (const bf16_t*)bias, (bf16_t*)vt, nullptr, nullptr,
               d, S, K, (long)K, (long)S_pad, d, 0, 0, 1.0f, 0L, x_bstride, (long)d * S_pad};
    a.w_row_perm = 1;
    a.n_store = S_pad;
    // the w operand's row pitch is K in gemm_w1.hip (weights are dense [N, K]): the token rows must be dense too
    FVK_CHECK(ldx == K, FVK_ERR_ARG, "fvk_gemm_vt_bf16: token rows must be dense (ldx=%ld != K=%d)", ldx, K);
    FVK_CHECK(fvk::gemm_pp_eligible(a) && fvk::gemm_w1_eligible(a), FVK_ERR_ARG,
              "fvk_gemm_vt_bf16: shape / alignment not served (needs K %% 128 == 0, d > 128, 16-byte aligned operands): d=%d S=%d K=%d", d, S, K);
    return fvk::gemm_w1_vt_launch(a, B, (hipStream_t)stream);
}

extern "C" int fvk_gemm_bf16_batched(const void* x, const void* w, void* out, int batch, int M, int N, int K, long lda, long ldc,
                                     long x_bstride, long w_bstride, long out_bstride, int epilogue, float epi_scalar,
                                     void* stream) {
    FVK_CHECK(epilogue == FVK_EPI_NONE || epilogue == FVK_EPI_DIV, FVK_ERR_ARG, "fvk_gemm_bf16_batched: epilogue %d unsupported", epilogue);
    return gemm_impl(x, w, nullptr, out, M, N, K, lda, ldc, epilogue, nullptr, nullptr, 0, epi_scalar, batch, x_bstride, w_bstride,
                     out_bstride, stream);
}
