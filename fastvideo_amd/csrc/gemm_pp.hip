// GEMM for the token-axis projections of the DiT block (QKV / out / FFN / cross-attn), gfx950 only: the fp8 (e4m3) kernel, and the bf16
// kernel for K % 64 != 0 — for every other bf16 shape gemm_ph.hip (K-step 64, +8..19 %) is the shipped kernel; "gemm_impl" 2 forces this one.
//   out[M,N] = epilogue( x[M,K] · w[N,K]^T + bias )      (same contract as gemm_bf16.hip; see include/fvk_amd.h)
//
// Design ("ping-pong"): one 512-thread workgroup per CU owns a 256(M) x 256(N) output tile; K is walked in steps of 32.
//   * LDS holds a 4-slot ring of K-steps (slot = 256 x-rows + 256 w-rows of 64 B = 32 KiB).  Tiles are filled by LDS-DMA
//     (buffer_load ... lds, 16 B per lane, no staging VGPRs, no ds_write) three K-steps ahead of their use and retired with
//     COUNTED waits (s_waitcnt vmcnt(8): two steps stay in flight across every barrier).
//   * LDS rows are 64 B (4 chunks of 16 B).  The chunk index is XOR-ed with (row>>2)&3, which makes every 32-row fragment
//     read (ds_read_b128) hit 16 distinct 16-B slots per lane group = conflict-free.  LDS-DMA writes lane-linearly, so the
//     swizzle is applied to the per-lane SOURCE address and to the read address (never to the destination).
//   * The 8 waves form two groups of 4 (waves w and w+4 share a SIMD).  Every K-step is {LOAD: 12 ds_read_b128 + 4 LDS-DMA
//     issues} barrier {MFMA: 16 v_mfma_f32_32x32x16_bf16} barrier, and the second group runs ONE barrier behind the first:
//     while one wave of a SIMD is in its MFMA segment its partner is in its LOAD segment, so the matrix pipe always has a
//     wave to issue from and LDS/L2 latency is hidden by the partner rather than by register double-buffering.
//   * Operands are swapped (A = w rows, B = x rows): the accumulator then holds 4 consecutive n per lane for one m, the
//     epilogue packs them to bf16 (+bias, first rounding), bounces the wave's 128 x 64 tile through its private LDS region
//     and stores whole 128-B output rows (16 B per lane) with the activation / gated-residual applied on the way out
//     (residual and gate are read coalesced, 16 B / 32 B per lane).
//   * Workgroup ids are remapped so each XCD (private L2) owns a contiguous range of tiles, walked in groups of 8 m-tiles.
// Rounding points equal gemm_bf16.hip: y = bf16(acc + bias), then the epilogue on float(y), then one more rounding.
#include "gemm_common.h"
#include "gemm_epilogue.h"

namespace {

constexpr int TM = 256, TN = 256, TK = 32;
constexpr int REGION = TM * TK * 2;        // 16 KiB: the x rows of a slot; the w rows follow
constexpr int SLOT = 2 * REGION;           // 32 KiB
constexpr int NSLOT = 4;
constexpr int LDS_BYTES = fvk::EPI_LDS_BYTES;  // 147 456 B  (>= NSLOT * SLOT = 131 072)
static_assert(LDS_BYTES >= NSLOT * SLOT, "epilogue staging must cover the ring");

using fvk::GemmArgs;

// FP8: x and w are OCP e4m3 bytes ([M,K] / [N,K], K-contiguous); a K-step is 64 elements = the same 64-B LDS rows, and the 16 bf16 MFMAs
// of a step become 8 v_mfma_scale_f32_32x32x64_f8f6f4 with unit block scales (2x the bf16 MFMA rate; the k order inside a fragment is
// irrelevant as long as A and B use the same one — scripts/probes/mfma_fp8_layout.hip).  The epilogue applies the dequantisation scales.
// LATE (A/B variant, gemm_impl 3): the fragment reads of a LOAD segment are waited for AFTER the barrier, at the head of the MFMA
// segment, so LDS latency overlaps the barrier wait; the refill distance then drops from three K-steps to two (the slot a step
// refills was read two LOAD segments earlier, i.e. those reads are retired before the barrier in front of this LOAD segment).
template <int EPI, bool FP8, bool LATE = false>
__global__ __launch_bounds__(512, 2) void gemm_pp_kernel(GemmArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef __attribute__((address_space(3))) void lds_void;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;  // 0: leading group, 1: runs one barrier behind
    const int l31 = lane & 31, hi = lane >> 5;
    const int wm = wave & 1, wn = wave >> 1;  // wave tile: rows wm*128.., cols wn*64..

    // ---- tile id: XCD-contiguous (block b runs on XCD b % 8), then groups of 8 m-tiles swept along n ----------------
    int tile_id;
    {
        const int nwg = gridDim.x, bid = blockIdx.x;
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
        tile_id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    constexpr int GM = 8;
    const int per_group = GM * a.ntn;
    const int gid = tile_id / per_group;
    const int first_m = gid * GM;
    const int gsz = (a.ntm - first_m) < GM ? (a.ntm - first_m) : GM;
    const int in_g = tile_id - gid * per_group;
    const int pid_m = first_m + in_g % gsz, pid_n = in_g / gsz;
    const int m0 = pid_m * TM, n0 = pid_n * TN;
    a.x += blockIdx.y * a.x_bstride;
    a.w += blockIdx.y * a.w_bstride;
    a.out += blockIdx.y * a.out_bstride;

    // ---- LDS-DMA staging: waves 0-3 stage the x panel, waves 4-7 the w panel; 4 pieces (16 rows x 64 B) per wave per step
    const bool stage_w = wave >= 4;
    const int srow0 = (wave & 3) * 64;
    constexpr int ES = FP8 ? 1 : 2;  // bytes per operand element
    const unsigned char* sbase = stage_w ? (const unsigned char*)a.w + (long)n0 * a.K * ES : (const unsigned char*)a.x + (long)m0 * a.lda * ES;
    const long sld = stage_w ? (long)a.K : a.lda;
    int srows = stage_w ? a.N - n0 : a.M - m0;
    srows = srows > 256 ? 256 : srows;
    const int nrec = (int)((((long)srows - 1) * sld + a.K) * ES);  // rows past the panel's valid rows read as zeros
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)sbase, 0, nrec, 0x00020000);
    int voff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = srow0 + 16 * i + (lane >> 2);
        const int c = (lane & 3) ^ ((lane >> 4) & 3);  // source chunk that lands at LDS chunk position lane&3 of this row
        voff[i] = (int)((long)row * sld * ES) + c * 16;
    }
    const int dma_dst = (stage_w ? REGION : 0) + srow0 * 64;  // + slot*SLOT + i*1024 (+ lane*16 by the hardware)
    const int nt = a.K * ES / (TK * 2);  // K-steps of 64 bytes per row

#define PP_ISSUE(TILE)                                                                                              \
    {                                                                                                               \
        const int t_ = (TILE);                                                                                      \
        const int so_ = __builtin_amdgcn_readfirstlane(t_ < nt ? t_ * (TK * 2) : 0); /* tail: harmless re-read */   \
        unsigned char* d_ = smem + (t_ & 3) * SLOT + dma_dst;                                                       \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                               \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void*)(d_ + i * 1024), 16, voff[i], so_, 0, 0);    \
    }

    // ---- fragment read offsets (bytes within a slot): row r, k-chunk c at r*64 + ((c ^ ((r>>2)&3)) << 4) ------------
    const int sw = (l31 >> 2) & 3;
    // bf16: fragment ks = chunk 2*ks + hi (8 k per lane).  fp8: ONE 32-byte fragment = chunks 2*hi and 2*hi + 1 (32 k per lane).
    const int c0 = FP8 ? 2 * hi : hi, c1 = FP8 ? 2 * hi + 1 : 2 + hi;
    const int xb0 = (wm * 128 + l31) * 64 + ((c0 ^ sw) << 4);
    const int xb1 = (wm * 128 + l31) * 64 + ((c1 ^ sw) << 4);
    const int wb0 = REGION + (wn * 64 + l31) * 64 + ((c0 ^ sw) << 4);
    const int wb1 = REGION + (wn * 64 + l31) * 64 + ((c1 ^ sw) << 4);

    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // ---- prologue: three steps in flight, step 0 landed -------------------------------------------------------------
    PP_ISSUE(0)
    PP_ISSUE(1)
    if (!LATE) PP_ISSUE(2)
    if (LATE) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (grp == 1) __builtin_amdgcn_s_barrier();  // stagger

    for (int u = 0; u < nt; ++u) {
        const unsigned char* slot = smem + (u & 3) * SLOT;
        // LOAD segment
        bf16x8 xf[4][2], wf[2][2];
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            wf[nb][0] = *reinterpret_cast<const bf16x8*>(slot + wb0 + nb * 2048);
            wf[nb][1] = *reinterpret_cast<const bf16x8*>(slot + wb1 + nb * 2048);
        }
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
            xf[mb][0] = *reinterpret_cast<const bf16x8*>(slot + xb0 + mb * 2048);
            xf[mb][1] = *reinterpret_cast<const bf16x8*>(slot + xb1 + mb * 2048);
        }
        if (LATE) {
            PP_ISSUE(u + 2)  // overwrites step u-2's slot
            if (grp == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");  // step u+1 landed (this wave's pieces)
        } else {
            PP_ISSUE(u + 3)  // overwrites step u-1's slot: every wave finished reading it before the barrier it just passed
            if (grp == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // step u+1 landed (this wave's pieces)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");              // fragments in registers before the slot can be refilled
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (LATE) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // MFMA segment
        __builtin_amdgcn_s_setprio(1);
        if (FP8) {
            typedef int v8i __attribute__((ext_vector_type(8)));
            typedef int v4i __attribute__((ext_vector_type(4)));
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int mb = 0; mb < 4; ++mb) {
                    const v4i w0 = __builtin_bit_cast(v4i, wf[nb][0]), w1 = __builtin_bit_cast(v4i, wf[nb][1]);
                    const v4i x0 = __builtin_bit_cast(v4i, xf[mb][0]), x1 = __builtin_bit_cast(v4i, xf[mb][1]);
                    const v8i wa = {w0[0], w0[1], w0[2], w0[3], w1[0], w1[1], w1[2], w1[3]};
                    const v8i xa = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
                    acc[nb][mb] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wa, xa, acc[nb][mb], 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
                }
        } else {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                    for (int mb = 0; mb < 4; ++mb)
                        acc[nb][mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[nb][ks], xf[mb][ks], acc[nb][mb], 0, 0, 0);
        }
        __builtin_amdgcn_s_setprio(0);
        if (grp == 0) {  // step u+1 landed (this wave's pieces)
            if (LATE) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    }
#undef PP_ISSUE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // tail re-reads must have landed before the ring is reused
    if (grp == 0) __builtin_amdgcn_s_barrier();       // pairs with the trailing barrier of the staggered group
    __builtin_amdgcn_s_barrier();

    fvk::gemm_tile_epilogue<EPI, FP8, true>(a, acc, smem, wave, lane, m0, n0);
#endif  // __HIP_DEVICE_COMPILE__
}

template <int EPI, bool FP8 = false, bool LATE = false>
int launch(const GemmArgs& a, int batch, hipStream_t s) {
    static FvkLdsConfigured configured;
    if (int rc = fvk_config_lds(configured, (const void*)gemm_pp_kernel<EPI, FP8, LATE>, LDS_BYTES, "fvk_gemm_bf16 (pp)")) return rc;
    hipLaunchKernelGGL((gemm_pp_kernel<EPI, FP8, LATE>), dim3(a.ntm * a.ntn, batch), dim3(512), LDS_BYTES, s, a);
    FVK_LAUNCH_CHECK();
    return FVK_OK;
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

namespace fvk {

bool gemm_pp_eligible(const GemmArgs& a) {
    if (a.M <= 128 || a.K % TK || a.N % 8 || a.lda % 8 || a.ldc % 8) return false;
    if (!aligned16(a.x) || !aligned16(a.w) || !aligned16(a.out)) return false;
    if (a.bias && (reinterpret_cast<uintptr_t>(a.bias) & 7)) return false;
    if (a.residual && !aligned16(a.residual)) return false;
    if (a.gate && !aligned16(a.gate)) return false;
    if ((a.x_bstride | a.w_bstride | a.out_bstride) % 8) return false;
    // 32-bit buffer offsets within one 256-row panel
    if (255L * a.lda * 2 + (long)a.K * 2 > 0x7fffffffL || 255L * a.K * 2 + (long)a.K * 2 > 0x7fffffffL) return false;
    return true;
}

int gemm_pp_fp8_launch(GemmArgs a, int epilogue, hipStream_t s) {
    a.ntm = (a.M + TM - 1) / TM;
    a.ntn = (a.N + TN - 1) / TN;
    switch (epilogue) {
        case FVK_EPI_NONE: return launch<FVK_EPI_NONE, true>(a, 1, s);
        case FVK_EPI_GELU_TANH: return launch<FVK_EPI_GELU_TANH, true>(a, 1, s);
        case FVK_EPI_SILU: return launch<FVK_EPI_SILU, true>(a, 1, s);
        default: return launch<FVK_EPI_RESIDUAL_GATE, true>(a, 1, s);
    }
}

int gemm_pp_launch(GemmArgs a, int epilogue, int batch, hipStream_t s) {
    a.ntm = (a.M + TM - 1) / TM;
    a.ntn = (a.N + TN - 1) / TN;
#if FVK_VARIANTS
    if (fvk::tunable(fvk::TUNE_GEMM_IMPL) == 3) {
        switch (epilogue) {
            case FVK_EPI_NONE: return launch<FVK_EPI_NONE, false, true>(a, batch, s);
            case FVK_EPI_GELU_TANH: return launch<FVK_EPI_GELU_TANH, false, true>(a, batch, s);
            case FVK_EPI_SILU: return launch<FVK_EPI_SILU, false, true>(a, batch, s);
            case FVK_EPI_DIV: return launch<FVK_EPI_DIV, false, true>(a, batch, s);
            default: return launch<FVK_EPI_RESIDUAL_GATE, false, true>(a, batch, s);
        }
    }
#endif
    switch (epilogue) {
        case FVK_EPI_NONE: return launch<FVK_EPI_NONE>(a, batch, s);
        case FVK_EPI_GELU_TANH: return launch<FVK_EPI_GELU_TANH>(a, batch, s);
        case FVK_EPI_SILU: return launch<FVK_EPI_SILU>(a, batch, s);
        case FVK_EPI_DIV: return launch<FVK_EPI_DIV>(a, batch, s);
        default: return launch<FVK_EPI_RESIDUAL_GATE>(a, batch, s);
    }
}

}  // namespace fvk
