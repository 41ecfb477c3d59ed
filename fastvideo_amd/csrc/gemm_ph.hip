// bf16 GEMM for the token-axis projections of the DiT block, gfx950 only — "phased" kernel: the shipped bf16 path for K % 64 == 0
// (gemm_impl 0; gemm_impl 2 forces gemm_pp.hip for A/B, scripts/probes/gemm_harness.cpp).
//   out[M,N] = epilogue( x[M,K] · w[N,K]^T + bias )      (same contract and rounding points as gemm_bf16.hip / gemm_pp.hip)
//
// Same 256(M) x 256(N) workgroup tile, 8 waves x (128 x 64), swapped operands and epilogue as gemm_pp.hip; what changes is the K walk:
//   * K-step 64: an LDS row is a full 128-B line of the operand (gemm_pp stages 64-B half lines), so an LDS-DMA piece (1 KiB per
//     wave instruction) covers 8 whole cache lines.  Rows are XOR-swizzled in 16-B chunks with (row >> 1) & 7 — applied to the
//     per-lane SOURCE address and to the read address — which makes every 32-row ds_read_b128 fragment read conflict-free.
//   * Two 64-KiB buffers (K-tiles t, t+1), each split into four 16-KiB UNITS by the time they are read, not by row range:
//     X0 = the x rows of every wave's first 64 m (rows {0-63, 128-191}), W0 / W1 = the first / second 32 n of every wave's
//     64 w rows, X1 = the remaining x rows.  A K-tile is four phases of 8 MFMAs (one 64 x 32 quadrant of the wave tile x K=64):
//         phase 4t+0: X0 x W0    4t+1: X0 x W1    4t+2: X1 x W1    4t+3: X1 x W0
//     and phase p's LOAD segment reads exactly ONE unit (4t+0: W0(t), 4t+1: W1(t), 4t+2: X1(t), 4t+3: X0(t+1)) and stages ONE unit
//     by LDS-DMA (2 pieces per wave).  Because a unit's region is free two phases after it was read, the unit read in phase p+8
//     is staged in phase p+2: six phases (~1500 cycles) of flight with 128 KiB of LDS, waited with a COUNTED s_waitcnt vmcnt(10).
//   * The two wave groups (waves w and w+4 share a SIMD) run one barrier apart as in gemm_pp: LOAD | MFMA alternate on every
//     SIMD.  Fragment reads are waited for AFTER the barrier (their latency overlaps the barrier wait).
// Barrier algebra (b(p) = barrier closing group 0's LOAD(p) = barrier in front of group 1's LOAD(p)):
//   WAR: unit read in LOAD(p): group 0 reads retire before b(p)+1, group 1 reads before b(p)+2; restaged in LOAD(p+2), which
//        both groups enter after b(p)+3.
//   RAW: a unit staged in LOAD(q) is read in LOAD(q+6).  Every wave waits for its own pieces (vmcnt(10): five younger units
//        stay in flight) at the end of MFMA(q+5) (group 0) / LOAD(q+5) (group 1) — both in front of the same barrier, which every
//        reader passes before its LOAD(q+6).
#include "gemm_common.h"
#include "gemm_epilogue.h"

namespace {

constexpr int TM = 256, TN = 256, TK = 64;
constexpr int ROWB = TK * 2;     // 128 B per LDS row
constexpr int XREG = TM * ROWB;  // 32 KiB: the x rows of a buffer; the w rows follow
constexpr int BUF = 2 * XREG;    // 64 KiB
constexpr int LDS_BYTES = fvk::EPI_LDS_BYTES;
static_assert(LDS_BYTES >= 2 * BUF, "epilogue staging must cover both buffers");

using fvk::GemmArgs;

// Measured on MI355X (scripts/probes/gemm_harness.cpp, same box, bit-identical outputs; profiles/r01_gemm_ph_ab.md), TFLOP/s
// gemm_pp -> this kernel:  QKV [32760,4608,1536] 1099 -> 1225, out-proj + gated residual [32760,1536,1536] 830 -> 991,
// FFN-in + GELU [32760,8960,1536] 1009 -> 1083, FFN-out + gated residual [32760,1536,8960] 1133 -> 1317, 8192^3 1308 -> 1436.
// VAR: measurement variants (gemm_impl = 4 + 8 * VAR; shipped = 28): bit 0 = no blanket lgkmcnt(0) at the head of an MFMA segment (the compiler's
// per-fragment counted waits instead; reads are retired at the END of the segment), bit 1 = stage before the fragment reads,
// bit 2 = no s_setprio around the MFMAs, bit 3 = prefetching gated-residual epilogue,
// bit 4 = staged units retired every second phase (vmcnt(8) in odd phases only), bits 5-6 = m-tiles per walk group 8 / 4 / 16 / 2.
// A/B: bit 0 +-0 %, bit 1 -1..2 %, bit 2 +1..2 % (+7 % at 4096^3), bit 3 +12 % on the K=1536 gated-residual GEMM (+3 % at K=8960),
// bit 4 +1..3 %, walk group 4 = 8, 16 -2..4 %.  bit 7 = persistent workgroups (min(tiles, 256) workgroups walk the tiles).
template <int EPI, int VAR>
__global__ __launch_bounds__(512, 2) void gemm_ph_kernel(GemmArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef __attribute__((address_space(3))) void lds_void;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;  // 0: leading group, 1: runs one barrier behind
    const int l31 = lane & 31, hi = lane >> 5;
    const int wm = wave & 1, wn = wave >> 1;  // wave tile: rows wm*128.., cols wn*64..

    a.x += blockIdx.y * a.x_bstride;
    a.w += blockIdx.y * a.w_bstride;
    a.out += blockIdx.y * a.out_bstride;
    // VAR bit 7: persistent workgroups — the launch has min(tiles, 256) workgroups and each walks tiles vb, vb + gridDim.x, ...
    // (one launch-time dispatch per CU instead of one per tile); otherwise gridDim.x == tiles and the loop body runs once.
    const int ntiles = a.ntm * a.ntn;
    constexpr bool PERSIST = (VAR & 128) != 0;
    int vb = blockIdx.x;
    do {
    // ---- tile id: XCD-contiguous (block b runs on XCD b % 8), then groups of 8 m-tiles swept along n ----------------
    int tile_id;
    {
        const int nwg = ntiles, bid = vb;
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
        tile_id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    constexpr int GM = ((VAR >> 5) & 3) == 0 ? 8 : ((VAR >> 5) & 3) == 1 ? 4 : ((VAR >> 5) & 3) == 2 ? 16 : 2;
    const int per_group = GM * a.ntn;
    const int gid = tile_id / per_group;
    const int first_m = gid * GM;
    const int gsz = (a.ntm - first_m) < GM ? (a.ntm - first_m) : GM;
    const int in_g = tile_id - gid * per_group;
    const int pid_m = first_m + in_g % gsz, pid_n = in_g / gsz;
    const int m0 = pid_m * TM, n0 = pid_n * TN;

    // ---- LDS-DMA staging: every wave stages 16 consecutive rows (2 pieces of 8 rows x 128 B) of each unit ------------------
    //   X0: rows (wave>>2)*128 + (wave&3)*16    X1: + 64        W0: rows (wave>>1)*64 + (wave&1)*16    W1: + 32
    const int xrow0 = (wave >> 2) * 128 + (wave & 3) * 16;
    const int wrow0 = (wave >> 1) * 64 + (wave & 1) * 16;
    int xvalid = a.M - m0, wvalid = a.N - n0;
    xvalid = xvalid > 256 ? 256 : xvalid;
    wvalid = wvalid > 256 ? 256 : wvalid;
    const long xld = a.lda * 2, wld = (long)a.K * 2;  // row pitch in bytes
    const unsigned char* xbase = (const unsigned char*)a.x + (long)m0 * xld;
    const unsigned char* wbase = (const unsigned char*)a.w + (long)n0 * wld;
    // one descriptor per unit kind, based at the wave's first row of that kind; rows past the operand's valid rows read as zeros
    // (the per-lane offset of such a row is >= num_records)
    auto mk = [](const unsigned char* base, long ld, int row0, int valid, int K) {
        const long nrec = valid > row0 ? ((long)(valid - row0) - 1) * ld + (long)K * 2 : 0;
        return __builtin_amdgcn_make_buffer_rsrc((void*)(base + (long)row0 * ld), 0, (int)nrec, 0x00020000);
    };
    const __amdgpu_buffer_rsrc_t r_x0 = mk(xbase, xld, xrow0, xvalid, a.K);
    const __amdgpu_buffer_rsrc_t r_x1 = mk(xbase, xld, xrow0 + 64, xvalid, a.K);
    const __amdgpu_buffer_rsrc_t r_w0 = mk(wbase, wld, wrow0, wvalid, a.K);
    const __amdgpu_buffer_rsrc_t r_w1 = mk(wbase, wld, wrow0 + 32, wvalid, a.K);
    // lane -> (row r = lane>>3 of the piece, LDS chunk position lane&7); piece i holds rows 8i + r.  LDS chunk position c' of tile
    // row R holds source chunk c' ^ ((R >> 1) & 7); every row0 is a multiple of 16, so (R >> 1) & 7 = 4i + (r >> 1).
    int xv[2], wv[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = lane >> 3;
        const int c = (lane & 7) ^ (4 * i + (r >> 1));
        xv[i] = (int)((long)(8 * i + r) * xld) + c * 16;
        wv[i] = (int)((long)(8 * i + r) * wld) + c * 16;
    }
    const int d_x0 = xrow0 * ROWB, d_x1 = (xrow0 + 64) * ROWB;
    const int d_w0 = XREG + wrow0 * ROWB, d_w1 = XREG + (wrow0 + 32) * ROWB;
    const int nt = a.K / TK;

    // unit k = 4*tile + kind (kind 0: X0, 1: W0, 2: W1, 3: X1 — the order in which a tile's units are read)
#define PH_STAGE(KIND, TILE)                                                                                          \
    {                                                                                                                 \
        const int t_ = (TILE);                                                                                        \
        const int so_ = __builtin_amdgcn_readfirstlane(t_ < nt ? t_ * ROWB : 0); /* tail: harmless re-read */         \
        unsigned char* d_ = smem + (t_ & 1) * BUF + ((KIND) == 0 ? d_x0 : (KIND) == 1 ? d_w0 : (KIND) == 2 ? d_w1 : d_x1); \
        if ((KIND) == 0) {                                                                                            \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r_x0, (lds_void*)(d_), 16, xv[0], so_, 0, 0);                    \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r_x0, (lds_void*)(d_ + 1024), 16, xv[1], so_, 0, 0);            \
        } else if ((KIND) == 1) {                                                                                     \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r_w0, (lds_void*)(d_), 16, wv[0], so_, 0, 0);                    \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r_w0, (lds_void*)(d_ + 1024), 16, wv[1], so_, 0, 0);            \
        } else if ((KIND) == 2) {                                                                                     \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r_w1, (lds_void*)(d_), 16, wv[0], so_, 0, 0);                    \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r_w1, (lds_void*)(d_ + 1024), 16, wv[1], so_, 0, 0);            \
        } else {                                                                                                      \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r_x1, (lds_void*)(d_), 16, xv[0], so_, 0, 0);                    \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r_x1, (lds_void*)(d_ + 1024), 16, xv[1], so_, 0, 0);            \
        }                                                                                                             \
    }

    // ---- fragment read offsets (bytes within a buffer): row R, k16-step ks, half hi -> R*128 + (((2ks + hi) ^ ((R>>1)&7)) << 4) ------
    int xo[4], wo[4];
    {
        const int sw = (l31 >> 1) & 7;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int co = ((2 * ks + hi) ^ sw) << 4;
            xo[ks] = (wm * 128 + l31) * ROWB + co;
            wo[ks] = XREG + (wn * 64 + l31) * ROWB + co;
        }
    }

    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    bf16x8 xa[2][4], xb[2][4], wf0[4], wf1[4];  // X0 / X1 fragments [m-block][ks], W0 / W1 fragments [ks]

#define PH_READ_X(DST, BUFP, MB0)                                                                          \
    _Pragma("unroll") for (int mb_ = 0; mb_ < 2; ++mb_) _Pragma("unroll") for (int ks_ = 0; ks_ < 4; ++ks_) \
        DST[mb_][ks_] = *reinterpret_cast<const bf16x8*>((BUFP) + xo[ks_] + ((MB0) + mb_) * 32 * ROWB);
#define PH_READ_W(DST, BUFP, NB)                   \
    _Pragma("unroll") for (int ks_ = 0; ks_ < 4; ++ks_) \
        DST[ks_] = *reinterpret_cast<const bf16x8*>((BUFP) + wo[ks_] + (NB) * 32 * ROWB);
#define PH_MFMA(NB, MB0, WF, XF)                                                                            \
    _Pragma("unroll") for (int ks_ = 0; ks_ < 4; ++ks_) _Pragma("unroll") for (int mb_ = 0; mb_ < 2; ++mb_)  \
        acc[NB][(MB0) + mb_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(WF[ks_], XF[mb_][ks_], acc[NB][(MB0) + mb_], 0, 0, 0);
// close a LOAD segment (group 1 retires its oldest staged unit here) and open the MFMA segment
#define PH_LOAD_END(ODD)                                                  \
    if (grp == 1) {                                                       \
        if (!(VAR & 16)) asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); \
        else if (ODD) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");    \
    }                                                                     \
    __builtin_amdgcn_sched_barrier(0);                                    \
    __builtin_amdgcn_s_barrier();                                         \
    __builtin_amdgcn_sched_barrier(0);                                    \
    if (!(VAR & 1)) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    \
    __builtin_amdgcn_sched_barrier(0);                                    \
    if (!(VAR & 4)) __builtin_amdgcn_s_setprio(1);
#define PH_MFMA_END(ODD)                                                  \
    if (!(VAR & 4)) __builtin_amdgcn_s_setprio(0);                        \
    if (VAR & 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       \
    if (grp == 0) {                                                       \
        if (!(VAR & 16)) asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); \
        else if (ODD) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");    \
    }                                                                     \
    __builtin_amdgcn_sched_barrier(0);                                    \
    __builtin_amdgcn_s_barrier();                                         \
    __builtin_amdgcn_sched_barrier(0);

    // ---- prologue: units 0..6 in flight (unit k is staged in phase k-7), unit 0 (X0 of tile 0) landed and read ("phase -1") --
    PH_STAGE(0, 0) PH_STAGE(1, 0) PH_STAGE(2, 0) PH_STAGE(3, 0)
    PH_STAGE(0, 1) PH_STAGE(1, 1) PH_STAGE(2, 1)
    asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (grp == 1) __builtin_amdgcn_s_barrier();  // stagger
    __builtin_amdgcn_sched_barrier(0);
    PH_READ_X(xa, smem, 0)
    PH_LOAD_END(1)
    PH_MFMA_END(1)

    for (int t = 0; t < nt; ++t) {
        const unsigned char* cur = smem + (t & 1) * BUF;
        const unsigned char* nxt = smem + ((t + 1) & 1) * BUF;
        // phase 4t+0: read W0(t), stage unit 4t+7 = X1(t+1); X0 x W0
        if (VAR & 2) {
            PH_STAGE(3, t + 1)
            __builtin_amdgcn_sched_barrier(0);
            PH_READ_W(wf0, cur, 0)
        } else {
            PH_READ_W(wf0, cur, 0)
            __builtin_amdgcn_sched_barrier(0);
            PH_STAGE(3, t + 1)
        }
        PH_LOAD_END(0)
        PH_MFMA(0, 0, wf0, xa)
        PH_MFMA_END(0)
        // phase 4t+1: read W1(t), stage unit 4t+8 = X0(t+2); X0 x W1
        if (VAR & 2) {
            PH_STAGE(0, t + 2)
            __builtin_amdgcn_sched_barrier(0);
            PH_READ_W(wf1, cur, 1)
        } else {
            PH_READ_W(wf1, cur, 1)
            __builtin_amdgcn_sched_barrier(0);
            PH_STAGE(0, t + 2)
        }
        PH_LOAD_END(1)
        PH_MFMA(1, 0, wf1, xa)
        PH_MFMA_END(1)
        // phase 4t+2: read X1(t), stage unit 4t+9 = W0(t+2); X1 x W1
        if (VAR & 2) {
            PH_STAGE(1, t + 2)
            __builtin_amdgcn_sched_barrier(0);
            PH_READ_X(xb, cur, 2)
        } else {
            PH_READ_X(xb, cur, 2)
            __builtin_amdgcn_sched_barrier(0);
            PH_STAGE(1, t + 2)
        }
        PH_LOAD_END(0)
        PH_MFMA(1, 2, wf1, xb)
        PH_MFMA_END(0)
        // phase 4t+3: read X0(t+1), stage unit 4t+10 = W1(t+2); X1 x W0
        if (VAR & 2) {
            PH_STAGE(2, t + 2)
            __builtin_amdgcn_sched_barrier(0);
            PH_READ_X(xa, nxt, 0)
        } else {
            PH_READ_X(xa, nxt, 0)
            __builtin_amdgcn_sched_barrier(0);
            PH_STAGE(2, t + 2)
        }
        PH_LOAD_END(1)
        PH_MFMA(0, 2, wf0, xb)
        PH_MFMA_END(1)
    }
#undef PH_STAGE
#undef PH_READ_X
#undef PH_READ_W
#undef PH_MFMA
#undef PH_LOAD_END
#undef PH_MFMA_END
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // tail re-reads must have landed before the buffers are reused
    if (grp == 0) __builtin_amdgcn_s_barrier();       // pairs with the trailing barrier of the staggered group
    __builtin_amdgcn_s_barrier();

    fvk::gemm_tile_epilogue<EPI, false, (VAR & 8) != 0>(a, acc, smem, wave, lane, m0, n0);
    if (PERSIST && vb + (int)gridDim.x < ntiles) {  // another tile follows: every wave's staging reads are done before the ring is refilled
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    } while (PERSIST && (vb += gridDim.x) < ntiles);  // tile loop (compiled out unless persistent)
#endif  // __HIP_DEVICE_COMPILE__
}

template <int EPI, int VAR>
int launch(const GemmArgs& a, int batch, hipStream_t s) {
    static FvkLdsConfigured configured;
    if (int rc = fvk_config_lds(configured, (const void*)gemm_ph_kernel<EPI, VAR>, LDS_BYTES, "fvk_gemm_bf16 (ph)")) return rc;
    const int tiles = a.ntm * a.ntn;
    const int grid = (VAR & 128) ? (tiles < 256 ? tiles : 256) : tiles;  // 256 CUs, one resident workgroup each
    hipLaunchKernelGGL((gemm_ph_kernel<EPI, VAR>), dim3(grid, batch), dim3(512), LDS_BYTES, s, a);
    FVK_LAUNCH_CHECK();
    return FVK_OK;
}

}  // namespace

namespace fvk {

bool gemm_ph_eligible(const GemmArgs& a) {
    // on top of gemm_pp_eligible: whole 64-element K-steps, at least two of them, 32-bit offsets inside one 256-row panel
    return a.K % TK == 0 && a.K >= 2 * TK && 255L * a.lda * 2 + (long)a.K * 2 <= 0x7fffffffL;
}

template <int VAR>
int launch_var(const GemmArgs& a, int epilogue, int batch, hipStream_t s) {
    switch (epilogue) {
        case FVK_EPI_NONE: return launch<FVK_EPI_NONE, VAR>(a, batch, s);
        case FVK_EPI_GELU_TANH: return launch<FVK_EPI_GELU_TANH, VAR>(a, batch, s);
        case FVK_EPI_SILU: return launch<FVK_EPI_SILU, VAR>(a, batch, s);
        case FVK_EPI_DIV: return launch<FVK_EPI_DIV, VAR>(a, batch, s);
        default: return launch<FVK_EPI_RESIDUAL_GATE, VAR>(a, batch, s);
    }
}

int gemm_ph_launch(GemmArgs a, int epilogue, int batch, hipStream_t s) {
    a.ntm = (a.M + TM - 1) / TM;
    a.ntn = (a.N + TN - 1) / TN;
    // shipped configuration = VAR 28 (bits 2 + 3 + 4); gemm_impl = 4 + 8 * VAR selects a measurement variant
    const int impl = fvk::tunable(fvk::TUNE_GEMM_IMPL);
    // persistent workgroups (VAR 156) measured within box-to-box noise of VAR 28 (two boxes: out-projection +4.8 % / -2 %, QKV +1.3 % / +5 %,
    // FFN shapes +0.3..1.8 %, single-round shapes -2..5 %): kept as a measurement variant, not shipped
    const int shipped = 28;
#if FVK_VARIANTS
    switch ((impl & 7) == 4 ? impl >> 3 : shipped) {
        case 0: return launch_var<0>(a, epilogue, batch, s);
        case 156: return launch_var<156>(a, epilogue, batch, s);
        default: break;
    }
#endif
    (void)impl;
    return launch_var<shipped>(a, epilogue, batch, s);
}

}  // namespace fvk
