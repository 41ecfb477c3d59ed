// EXPERIMENT RECORD (not built into libfvk_amd.so): a "stream" variant of the 256x256 GEMM — every wave runs one continuous
// software-pipelined MFMA / ds_read / LDS-DMA stream with one barrier per K-step instead of gemm_pp.hip's staggered MFMA / load
// segments.  Measured on MI355X (scripts/gemm_ab.py, same box): TK=32 x 4-slot ring (this file) = gemm_pp within +-3 %;
// a TK=64 x 2-slot form +3..+7 % on QKV / out / FFN-out, -4 % on FFN-in.  Timing ablations (FVK_ST_ABL): removing the LDS-DMA
// instructions +25..33 % (same at a quarter of the bytes => issue cost ~55 cycles per buffer_load...lds, not bandwidth);
// removing the fragment reads +14 %; removing the vmcnt wait +4..8 %.  s_memtime probe (FVK_ST_PROBE): 1024 pipe-cycles of MFMA per
// step take ~1230 cycles of stream + ~270 cycles of wait/barrier/restart.  Kept for the next round's work on a loader-wave design.
// bf16 GEMM for the token-axis projections of the DiT block (QKV / out / FFN / cross-attn), gfx950 only — "stream" kernel.
//   out[M,N] = epilogue( x[M,K] · w[N,K]^T + bias )      (same contract as gemm_bf16.hip / gemm_pp.hip; see include/fvk_amd.h)
//
// Design: one 512-thread workgroup per CU owns a 256(M) x 256(N) output tile; K is walked in steps of 32 through a 4-slot LDS ring.
//   * s_memtime probes of the ping-pong kernels showed that a lock-step MFMA-segment / load-segment alternation pays ~300-450 cycles
//     of hand-off per segment.  Here every wave runs ONE continuous software-pipelined stream: per K-step 16 MFMAs
//     (v_mfma_f32_32x32x16_bf16) and their 12 ds_read_b128 fragment reads, each k16 group of 6 reads issued one group (8 MFMAs) ahead
//     — ACROSS the K-step barrier too, because a step's tiles are landed and visible one barrier before they are used.
//     sched_barrier(0) after every {MFMA, read, DMA} unit pins that source order (left alone hipcc emits read / wait / MFMA triplets).
//     The two waves of a SIMD are not phase-locked; whichever has an MFMA ready feeds the matrix pipe.
//   * LDS-DMA (buffer_load ... lds, no staging VGPRs, no ds_write) fills the ring three K-steps ahead; counted s_waitcnt vmcnt(4)
//     before the ONE raw s_barrier per K-step (never __syncthreads, which would drain the queue).  Timing ablations: the DMA costs its
//     issuing wave ~55 cycles per instruction regardless of size, during which that wave issues no MFMA — so a wave's 4 pieces ride
//     in the MFMA gaps, and the two waves of a SIMD issue theirs in different halves of the step (waves 0-3 first, 4-7 second) so that
//     one of them always has MFMAs to issue.
//   * LDS rows are 64 B (4 chunks of 16 B); the chunk index is XOR-ed with (row>>2)&3 on the per-lane SOURCE address and on the read
//     address: conflict-free 32-row fragment reads.
//   * Operands are swapped (A = w rows, B = x rows) and the epilogue is gemm_pp.hip's: accumulators -> bf16(+bias) -> wave-private LDS
//     bounce -> whole 128-B output rows with the activation / gated residual applied on the way out.  Same rounding points.
//   * Workgroup ids are remapped so each XCD (private L2) owns a contiguous range of tiles, walked in groups of 8 m-tiles.
#include "gemm_common.h"

namespace {

constexpr int TM = 256, TN = 256, TK = 32;
constexpr int REGION = TM * TK * 2;        // 16 KiB: the x rows of a slot; the w rows follow
constexpr int SLOT = 2 * REGION;           // 32 KiB
constexpr int NSLOT = 4;
constexpr int EPI_PITCH = 144;             // bytes per staged output row (64 bf16 + 16 B pad)
constexpr int EPI_WAVE = 128 * EPI_PITCH;  // 18 432 B per wave
constexpr int LDS_BYTES = 8 * EPI_WAVE;    // 147 456 B  (>= NSLOT * SLOT = 131 072)
static_assert(LDS_BYTES >= NSLOT * SLOT, "epilogue staging must cover the ring");

using fvk::GemmArgs;

__device__ __forceinline__ float gelu_tanh_fast(float x) {
    // 0.5 x (1 + tanh(u)) == x / (1 + exp(-2u)),  u = sqrt(2/pi) (x + 0.044715 x^3); exp via v_exp_f32 (exp2).
    const float c0 = -2.0f * 0.7978845608028654f * 1.4426950408889634f;
    const float c1 = c0 * 0.044715f;
    const float t = __builtin_amdgcn_exp2f(x * __builtin_fmaf(x * x, c1, c0));
    return x * __builtin_amdgcn_rcpf(1.0f + t);
}

template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm_st_kernel(GemmArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef __attribute__((address_space(3))) void lds_void;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
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

    // ---- LDS-DMA staging: waves 0-3 stage the x panel, waves 4-7 the w panel; 4 pieces (16 rows x 64 B) per wave per K-step
    const int grp = wave >> 2;
    const bool stage_w = wave >= 4;
    const int srow0 = (wave & 3) * 64;
    const bf16_t* sbase = stage_w ? a.w + (long)n0 * a.K : a.x + (long)m0 * a.lda;
    const long sld = stage_w ? (long)a.K : a.lda;
    int srows = stage_w ? a.N - n0 : a.M - m0;
    srows = srows > 256 ? 256 : srows;
    const int nrec = (int)((((long)srows - 1) * sld + a.K) * 2);  // rows past the panel's valid rows read as zeros
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)sbase, 0, nrec, 0x00020000);
    int voff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = srow0 + 16 * i + (lane >> 2);
        const int c = (lane & 3) ^ ((lane >> 4) & 3);  // source chunk that lands at LDS chunk position lane&3 of this row
        voff[i] = (int)((long)row * sld * 2) + c * 16;
    }
    const int dma_dst = (stage_w ? REGION : 0) + srow0 * 64;  // + slot*SLOT + i*1024 (+ lane*16 by the hardware)
    const int nt = a.K / TK;

#define ST_ISSUE1(TILE, I)                                                                                          \
    {                                                                                                               \
        const int t_ = (TILE);                                                                                      \
        const int so_ = __builtin_amdgcn_readfirstlane(t_ < nt ? t_ * (TK * 2) : 0); /* tail: harmless re-read */   \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void*)(smem + (t_ & 3) * SLOT + dma_dst + (I) * 1024), 16, voff[I], so_, 0, 0); \
    }

    // ---- fragment read offsets (bytes within a slot): row r, k-chunk c at r*64 + ((c ^ ((r>>2)&3)) << 4) ------------
    const int sw = (l31 >> 2) & 3;
    const int fo0 = l31 * 64 + ((hi ^ sw) << 4), fo1 = l31 * 64 + (((2 + hi) ^ sw) << 4);
    const int xbase = wm * 128 * 64, wbase = REGION + wn * 64 * 64;

    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragment R (0,1 = w rows nb; 2..5 = x rows mb) of k16 group C (0/1) of the slot at SLOTP
    bf16x8 wf[2][2], xf[2][4];
#define ST_READ1(SLOTP, C, BUF, R)                                                                                  \
    {                                                                                                               \
        if ((R) < 2) wf[BUF][R] = *reinterpret_cast<const bf16x8*>((SLOTP) + wbase + ((C) ? fo1 : fo0) + (R) * 2048); \
        else xf[BUF][(R) - 2] = *reinterpret_cast<const bf16x8*>((SLOTP) + xbase + ((C) ? fo1 : fo0) + ((R) - 2) * 2048); \
    }

    // ---- prologue: three steps in flight, steps 0 and 1 landed and visible; first fragment group in registers ---------
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) ST_ISSUE1(t, i)
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int r = 0; r < 6; ++r) ST_READ1(smem, 0, 0, r)
    __builtin_amdgcn_sched_barrier(0);

#ifdef FVK_ST_PROBE  // timing probe build: workgroup 0's waves sum s_memtime deltas between stream boundaries; written to a.gate as uint64
    unsigned long long pacc_[4] = {0, 0, 0, 0}, plast_ = 0;
#define ST_STAMP(K)                                                               \
    if (blockIdx.x == 0 && blockIdx.y == 0 && u >= 8) {                           \
        const unsigned long long now_ = __builtin_readcyclecounter();            \
        if ((K) != 0 || u > 8) pacc_[K] += now_ - plast_;                         \
        plast_ = now_;                                                            \
    }
#else
#define ST_STAMP(K)
#endif
    for (int u = 0; u < nt; ++u) {
        ST_STAMP(0)
        const unsigned char* slot = smem + (u & 3) * SLOT;
        const unsigned char* slot_n = smem + ((u + 1) & 3) * SLOT;  // landed and visible since the previous barrier
        __builtin_amdgcn_s_setprio(1);
        // Source order == issue order (sched_barrier(0) after every unit).  Step u+3 overwrites the slot of step u-1, which every wave
        // finished reading before the barrier at the end of step u-1.
#pragma unroll
        for (int i = 0; i < 8; ++i) {  // k16 group 0 of step u (buffer 0); prefetch group 1 into buffer 1
            acc[i >> 2][i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[0][i >> 2], xf[0][i & 3], acc[i >> 2][i & 3], 0, 0, 0);
            if (i < 6) ST_READ1(slot, 1, 1, i)
            if ((i & 1) && grp == 0) ST_ISSUE1(u + 3, i >> 1)
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {  // k16 group 1 (buffer 1); prefetch group 0 of step u+1 into buffer 0
            acc[i >> 2][i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[1][i >> 2], xf[1][i & 3], acc[i >> 2][i & 3], 0, 0, 0);
            if (i < 6) ST_READ1(slot_n, 0, 0, i)
            if ((i & 1) && grp == 1) ST_ISSUE1(u + 3, i >> 1)
            __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_s_setprio(0);
        ST_STAMP(1)
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");  // this wave's pieces of step u+2 have landed (step u+3 stays in flight)
        ST_STAMP(2)
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();                      // ... everyone's have, and everyone has finished reading step u
        __builtin_amdgcn_sched_barrier(0);
        ST_STAMP(3)
    }
#ifdef FVK_ST_PROBE
    if (blockIdx.x == 0 && blockIdx.y == 0 && lane == 0 && a.gate)
        for (int i = 0; i < 4; ++i) reinterpret_cast<unsigned long long*>(const_cast<float*>(a.gate))[wave * 4 + i] = pacc_[i];
#endif
#undef ST_ISSUE1
#undef ST_READ1
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // tail re-reads / prefetches done before the ring becomes epilogue staging
    __builtin_amdgcn_s_barrier();

    // ---- epilogue: acc[nb][mb][r] = D[n = nb*32 + (r&3) + 8(r>>2) + 4hi][m = mb*32 + l31] ----------------------------
    unsigned char* st = smem + wave * EPI_WAVE;
    const int ncol0 = n0 + wn * 64;
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
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) {
                bf16x4 y;
#pragma unroll
                for (int e = 0; e < 4; ++e) y[e] = (bf16_t)(acc[nb][mb][4 * g + e] + b4[e]);
                *reinterpret_cast<bf16x4*>(st + (mb * 32 + l31) * EPI_PITCH + nl * 2) = y;
            }
        }
    // the staging region is private to this wave: program order + the compiler's lgkmcnt wait are sufficient
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
                for (int e = 0; e < 8; ++e) y[e] = (bf16_t)__fadd_rn((float)res[e], __fmul_rn((float)y[e], gt[e]));
            }
            st_bf16x8(a.out + (long)m * a.ldc + n, y);
        }
    }
#endif  // __HIP_DEVICE_COMPILE__
}

template <int EPI>
int launch(const GemmArgs& a, int batch, hipStream_t s) {
    static bool configured = false;
    if (!configured) {
        if (hipFuncSetAttribute((const void*)gemm_st_kernel<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) !=
            hipSuccess) {
            fvk_set_error("fvk_gemm_bf16 (stream): cannot set dynamic LDS size %d", LDS_BYTES);
            return FVK_ERR_LAUNCH;
        }
        configured = true;
    }
    hipLaunchKernelGGL((gemm_st_kernel<EPI>), dim3(a.ntm * a.ntn, batch), dim3(512), LDS_BYTES, s, a);
    FVK_LAUNCH_CHECK();
    return FVK_OK;
}

}  // namespace

namespace fvk {

// eligibility = gemm_pp_eligible plus K % 64 == 0
int gemm_st_launch(GemmArgs a, int epilogue, int batch, hipStream_t s) {
    a.ntm = (a.M + TM - 1) / TM;
    a.ntn = (a.N + TN - 1) / TN;
    switch (epilogue) {
        case FVK_EPI_NONE: return launch<FVK_EPI_NONE>(a, batch, s);
        case FVK_EPI_GELU_TANH: return launch<FVK_EPI_GELU_TANH>(a, batch, s);
        case FVK_EPI_SILU: return launch<FVK_EPI_SILU>(a, batch, s);
        case FVK_EPI_DIV: return launch<FVK_EPI_DIV>(a, batch, s);
        default: return launch<FVK_EPI_RESIDUAL_GATE>(a, batch, s);
    }
}

}  // namespace fvk
