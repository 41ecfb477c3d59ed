// bf16 GEMM for FEW-TILE token-axis problems, gfx950 only — gemm_w1.hip's design on a 256(M) x 128(N) workgroup tile (round 4).
//   out[M,N] = epilogue( x[M,K] · w[N,K]^T + bias )      (same contract and rounding points; BYTE-IDENTICAL results to gemm_w1.hip: every
//   output element accumulates the same 32-k v_mfma_f32_16x16x32_bf16 steps in the same order, then the same epilogue arithmetic)
//
// Why: under sequence parallelism a rank keeps S / P token rows.  At P = 8 the three N = 1536 projections of a Wan2.1-1.3B layer (attention out,
// cross-attention q and out) and FFN-out are 16 x 6 = 96 tiles of 256 x 256 on 256 CUs: 37 % of the chip computes, and K = 1536 leaves nothing to
// split (split-K would move the rounding points).  Measured per-rank compute at P = 8 was 2.04x an eighth of the P = 1 time for those GEMMs
// (profiles/r03_sp_rank_emulation.json).  With 256 x 128 tiles the same problems are 192 workgroups, one per CU in a single round, each with
// half the work.  The kernel is chosen by the host only when the 256 x 256 grid would leave at least half the chip idle, and since the
// arithmetic is the same as gemm_w1's, a rank's results do not depend on which of the two ran (i.e. not on P).
//
// Structure (see gemm_w1.hip for the vocabulary): four waves, one per SIMD, wave tile 128(m) x 64(n) = 128 accumulator AGPRs; K-step 64 with
// 128-B swizzled LDS rows; staging UNITS of 16 KiB by read time — per K-tile X0 / X1 (the first / second 64 rows of each wave's 128) and ONE
// w unit W0 (the tile's 128 w rows).  A K-tile is TWO phases of 32 MFMAs:
//     phase 2t  : X0 x W0(t)   reads X1(t) (8 fragments)                        stages X0(t+3), W0(t+3)   (8 pieces per wave)
//     phase 2t+1: X1 x W0(t)   reads X0(t+1), W0(t+1) (16 fragments)            stages X1(t+3)            (4 pieces per wave)
// Three K-tile buffers of 48 KiB (144 KiB): the units read during phase p were staged during phase p-5 and are waited for at the barrier closing
// phase p-1 (counted wait vmcnt(24): the pieces of the four youngest phases stay in flight, ~2000 cycles of flight); a unit's slot is refilled
// in the phase after the one that read it (WAR: fragment reads are retired with lgkmcnt(0) before every barrier).  One barrier per phase.
// One workgroup per tile (these are single-round problems: no persistent walk), direct epilogue from the accumulators (gemm_w1_epilogue.h).
#include "gemm_common.h"
#include "gemm_epilogue.h"
#include "gemm_w1_epilogue.h"

namespace {

constexpr int TM = 256, TN = 128, TK = 64;
constexpr int ROWB = TK * 2;          // 128 B per LDS row
constexpr int XREG = TM * ROWB;       // 32 KiB: the x rows of a buffer; the 128 w rows (16 KiB) follow
constexpr int BUF = XREG + TN * ROWB; // 48 KiB per K-tile
constexpr int LDS_BYTES = 3 * BUF;    // 144 KiB

using fvk::GemmArgs;

#define W1N_MFMA16(ACC, A, B) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(ACC) : "v"(A), "v"(B))
#define W1N_MFMA16Z(ACC, A, B) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=a"(ACC) : "v"(A), "v"(B))

template <int EPI, bool NT>
__global__ __launch_bounds__(256, 1) void gemm_w1n_kernel(GemmArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef __attribute__((address_space(3))) void lds_void;
    FVK_CLAIM_WHOLE_REGISTER_FILE();
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 1, wn = wave >> 1;  // wave tile: rows wm*128.., cols wn*64..

    a.x += blockIdx.y * a.x_bstride;
    a.w += blockIdx.y * a.w_bstride;
    a.out += blockIdx.y * a.out_bstride;
    // tile id: XCD-contiguous (block b runs on XCD b % 8), m fastest inside groups of 8 m-tiles (the n-tiles of a group share their x panel in L2)
    const int ntiles = a.ntm * a.ntn;
    int m0, n0;
    {
        const int bid = (int)blockIdx.x;
        const int q = ntiles >> 3, r = ntiles & 7, xcd = bid & 7;
        const int tile_id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
        constexpr int GM = 8;
        const int per_group = GM * a.ntn;
        const int gid = tile_id / per_group;
        const int first_m = gid * GM;
        const int gsz = (a.ntm - first_m) < GM ? (a.ntm - first_m) : GM;
        const int in_g = tile_id - gid * per_group;
        m0 = (first_m + in_g % gsz) * TM;
        n0 = (in_g / gsz) * TN;
    }

    // ---- LDS-DMA staging: every wave stages 32 consecutive rows (4 pieces of 8 rows x 128 B) of each unit -----------------------
    //   X0: x rows (wave>>1)*128 + (wave&1)*32   X1: + 64     W0: w rows wave*32
    const int rowx = (wave >> 1) * 128 + (wave & 1) * 32, roww = wave * 32;
    const long xld = a.lda * 2, wld = (long)a.K * 2;  // row pitch in bytes
    auto mk = [&](const unsigned char* base, long ld, int r0, int valid) {  // rows past the operand's valid rows read as zeros
        const long nrec = valid > r0 ? ((long)(valid - r0) - 1) * ld + (long)a.K * 2 : 0;
        return __builtin_amdgcn_make_buffer_rsrc((void*)(base + (long)r0 * ld), 0, (int)nrec, 0x00020000);
    };
    int xvalid = a.M - m0, wvalid = a.N - n0;
    xvalid = xvalid > TM ? TM : xvalid;
    wvalid = wvalid > TN ? TN : wvalid;
    const unsigned char* xbase = (const unsigned char*)a.x + (long)m0 * xld;
    const unsigned char* wbase = (const unsigned char*)a.w + (long)n0 * wld;
    const __amdgpu_buffer_rsrc_t r_x0 = mk(xbase, xld, rowx, xvalid), r_x1 = mk(xbase, xld, rowx + 64, xvalid), r_w0 = mk(wbase, wld, roww, wvalid);
    // lane -> (row r = lane>>3 of the piece, LDS chunk position lane&7); piece i holds rows 8i + r.  x rows: LDS chunk position c' of tile row R holds
    // source chunk c' ^ ((R >> 1) & 7) (the staged row groups start at multiples of 32, so (R >> 1) & 7 = 4(i & 1) + (r >> 1)); w rows:
    // c' ^ (((R >> 1) & 1) | (((R >> 3) & 3) << 1)) — a w fragment reads rows {0-3, 8-11, 16-19, 24-27} (+4) of a 32-row group (the direct
    // epilogue's column order, gemm_w1_epilogue.h), which that swizzle spreads over all 16 (parity, position) slots.  Exactly gemm_w1.hip's layout.
    int xv[4], wv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = lane >> 3;
        const int c = (lane & 7) ^ (4 * (i & 1) + (r >> 1));
        const int cw = (lane & 7) ^ (((r >> 1) & 1) | (i << 1));
        xv[i] = (int)((long)(8 * i + r) * xld) + c * 16;
        wv[i] = (int)((long)(8 * i + r) * wld) + cw * 16;
    }
    const int d_x0 = rowx * ROWB, d_x1 = (rowx + 64) * ROWB, d_w0 = XREG + roww * ROWB;
    const int nt = a.K / TK;  // K-tiles (even: gemm_w1n_eligible)

    // kinds: 0 = X0, 1 = W0, 3 = X1.  One piece (PC) of unit KIND of K-tile TILE into the buffer at byte offset BOFF; K-tiles past the end re-read
    // K-tile 0 (never consumed)
#define W1N_STAGE(KIND, TILE, BOFF, PC)                                                                               \
    {                                                                                                                 \
        const int t_ = (TILE);                                                                                        \
        const int so_ = __builtin_amdgcn_readfirstlane(t_ < nt ? t_ * ROWB : 0);                                      \
        unsigned char* d_ = smem + (BOFF) + ((KIND) == 0 ? d_x0 : (KIND) == 1 ? d_w0 : d_x1) + (PC) * 1024;           \
        if ((KIND) == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(r_x0, (lds_void*)(d_), 16, xv[PC], so_, 0, 0);      \
        else if ((KIND) == 1) __builtin_amdgcn_raw_ptr_buffer_load_lds(r_w0, (lds_void*)(d_), 16, wv[PC], so_, 0, 0); \
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(r_x1, (lds_void*)(d_), 16, xv[PC], so_, 0, 0);                  \
    }
#define W1N_STAGE_UNIT(KIND, TILE, BOFF) W1N_STAGE(KIND, TILE, BOFF, 0) W1N_STAGE(KIND, TILE, BOFF, 1) W1N_STAGE(KIND, TILE, BOFF, 2) W1N_STAGE(KIND, TILE, BOFF, 3)

    // ---- fragment read offsets (bytes within a buffer): 16x16x32 fragments, row R = lane & 15, k32-step ks, quarter q = lane >> 4 -> chunk 4ks + q
    int xo[2], wo[2];
    {
        const int rl = lane & 15;
        const int sw = (rl >> 1) & 7;
        const int rho = 8 * (rl >> 2) + (rl & 3), sww = ((rl & 3) >> 1) | ((rl >> 2) << 1);  // lane row of a w tile within its 32-row group (+4: second tile)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            xo[ks] = (wm * 128 + rl) * ROWB + (((4 * ks + (lane >> 4)) ^ sw) << 4);
            wo[ks] = XREG + (wn * 64 + rho) * ROWB + (((4 * ks + (lane >> 4)) ^ sww) << 4);
        }
    }

    f32x4 acc16[4][8];                  // [nb][mb]: D[n = 16 nb' ...][m = 16 mb + (lane & 15)] (column order: gemm_w1_epilogue.h)
    bf16x8 XA[8], XB[8], WA[8], WB[8];  // fragment sets [ks * 4 + block]

    // fragment J (0..7: block J % 4, k-step J / 4) of a unit at buffer byte offset BOFF: IS_X selects the x / w rows, ROW0 (0 / 64) the unit's first row
#define W1N_READ1(DST, BOFF, IS_X, ROW0, J)                                                                           \
    DST[J] = *reinterpret_cast<const bf16x8*>(smem + (BOFF) + ((IS_X) ? xo[(J) / 4] : wo[(J) / 4]) +                  \
                                              ((IS_X) ? ((ROW0) + ((J) % 4) * 16) : (32 * (((J) % 4) >> 1) + 4 * (((J) % 4) & 1))) * ROWB);
#define W1N_PHASE_END()                                             \
    asm volatile("s_waitcnt vmcnt(24) lgkmcnt(0)" ::: "memory");    \
    __builtin_amdgcn_sched_barrier(0);                              \
    __builtin_amdgcn_s_barrier();                                   \
    __builtin_amdgcn_sched_barrier(0);
    // One phase: 32 MFMAs (ks outer, then 4 x 4 blocks, the w fragment held over 4 MFMAs) on accumulator rows MQ*4.., with SLOT_(i) after MFMA i.
#define W1N_PHASE(ZF, XS, WS, MQ, SLOT_)                                                                \
    _Pragma("unroll") for (int i_ = 0; i_ < 32; ++i_) {                                                 \
        const int ks_ = i_ >> 4, nb_ = (i_ >> 2) & 3, mb_ = i_ & 3;                                     \
        if ((ZF) && ks_ == 0) W1N_MFMA16Z(acc16[nb_][(MQ) * 4 + mb_], WS[ks_ * 4 + nb_], XS[ks_ * 4 + mb_]); \
        else W1N_MFMA16(acc16[nb_][(MQ) * 4 + mb_], WS[ks_ * 4 + nb_], XS[ks_ * 4 + mb_]);              \
        SLOT_(i_)                                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                              \
    }                                                                                                   \
    W1N_PHASE_END()

    // ---- prologue: the three K-tiles' units staged in read order; X0(0), W0(0) landed and read; X1(0) landed ---------------------------
    W1N_STAGE_UNIT(0, 0, 0) W1N_STAGE_UNIT(1, 0, 0) W1N_STAGE_UNIT(3, 0, 0)
    W1N_STAGE_UNIT(0, 1, BUF) W1N_STAGE_UNIT(1, 1, BUF) W1N_STAGE_UNIT(3, 1, BUF)
    W1N_STAGE_UNIT(0, 2, 2 * BUF) W1N_STAGE_UNIT(1, 2, 2 * BUF) W1N_STAGE_UNIT(3, 2, 2 * BUF)
    asm volatile("s_waitcnt vmcnt(28)" ::: "memory");  // 36 pieces issued: the 8 oldest (X0(0), W0(0)) have landed
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 8; ++j) { W1N_READ1(XA, 0, true, 0, j) }
#pragma unroll
    for (int j = 0; j < 8; ++j) { W1N_READ1(WA, 0, false, 0, j) }
    asm volatile("s_waitcnt vmcnt(24) lgkmcnt(0)" ::: "memory");  // X1(0) landed (read in phase 0)
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);

    // buffer byte offsets of K-tiles t, t+1, t+2 (rotating: K-tile tau lives in buffer tau % 3)
    int b0 = 0, b1 = BUF, b2 = 2 * BUF;
    // phase slots.  Even phase (K-tile T_, buffer BC_): 8 fragment reads of X1(T_) after MFMAs 0, 2, .., 14; 8 pieces X0(T_+3), W0(T_+3) after MFMAs 16, 18, .., 30.
    // Odd phase: 16 fragment reads X0(T_+1), W0(T_+1) (buffer BN_) after MFMAs 0..15; 4 pieces X1(T_+3) after MFMAs 18, 22, 26, 30.  Staged K-tile
    // T_+3 goes to buffer BC_ (= (T_+3) % 3), whose X0 / W0 slots were read in the previous phase and whose X1 slot is read in the even phase.
#define W1N_EVEN_SLOT(T_, BC_, i_)                                                                      \
    if ((i_) < 16 && ((i_) & 1) == 0) { W1N_READ1(XB, BC_, true, 64, ((i_) >> 1) & 7) }                 \
    else if ((i_) >= 16 && ((i_) & 1) == 0) {                                                           \
        if ((((i_) - 16) >> 1) < 4) W1N_STAGE(0, (T_) + 3, BC_, (((i_) - 16) >> 1) & 3)                 \
        else W1N_STAGE(1, (T_) + 3, BC_, (((i_) - 24) >> 1) & 3)                                        \
    }
#define W1N_ODD_SLOT(T_, BC_, BN_, WDST, i_)                                                            \
    if ((i_) < 8) { W1N_READ1(XA, BN_, true, 0, (i_) & 7) }                                             \
    else if ((i_) < 16) { W1N_READ1(WDST, BN_, false, 0, ((i_) - 8) & 7) }                              \
    else if (((i_) & 3) == 2) W1N_STAGE(3, (T_) + 3, BC_, (((i_) - 18) >> 2) & 3)

#define W1N_S0(i_) W1N_EVEN_SLOT(t, b0, i_)
#define W1N_S1(i_) W1N_ODD_SLOT(t, b0, b1, WB, i_)
#define W1N_S2(i_) W1N_EVEN_SLOT(t + 1, b1, i_)
#define W1N_S3(i_) W1N_ODD_SLOT(t + 1, b1, b2, WA, i_)
#define W1N_ITER(ZF_)                        \
        W1N_PHASE(ZF_, XA, WA, 0, W1N_S0)    \
        W1N_PHASE(ZF_, XB, WA, 1, W1N_S1)    \
        W1N_PHASE(0, XA, WB, 0, W1N_S2)      \
        W1N_PHASE(0, XB, WB, 1, W1N_S3)      \
        { const int nb0_ = b2; b2 = b1; b1 = b0; b0 = nb0_; }
    // K-tile pairs: K-tile t has W0 in WA, K-tile t+1 in WB.  The first pair is peeled: its first K-tile's MFMAs are each accumulator's first
    // touch and take 0 as C (no zeroing pass).  After a pair the buffers rotate by two: (b0, b1, b2) <- (b2, b0, b1).
    { const int t = 0; W1N_ITER(1) }
    for (int t = 2; t < nt; t += 2) { W1N_ITER(0) }
#undef W1N_ITER
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the tail re-reads have landed before the workgroup's LDS is released
    // the accumulators were written by asm MFMAs: the compiler knows no hazard distance to its own reads of them
    asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) asm volatile("" : "+a"(acc16[i][j]));
    fvk::w1_direct_epilogue<EPI, false, NT, 2>(a, acc16, m0, n0, wm, wn, lane);
#undef W1N_STAGE
#undef W1N_STAGE_UNIT
#undef W1N_READ1
#undef W1N_PHASE
#undef W1N_PHASE_END
#endif  // __HIP_DEVICE_COMPILE__
}

template <int EPI>
int launch(const GemmArgs& a, int batch, hipStream_t s) {
    static FvkLdsConfigured configured;
    if (int rc = fvk_config_lds(configured, (const void*)gemm_w1n_kernel<EPI, false>, LDS_BYTES, "fvk_gemm_bf16 (w1n)")) return rc;
    hipLaunchKernelGGL((gemm_w1n_kernel<EPI, false>), dim3(a.ntm * a.ntn, batch), dim3(256), LDS_BYTES, s, a);
    FVK_LAUNCH_CHECK();
    return FVK_OK;
}

}  // namespace

namespace fvk {

bool gemm_w1n_eligible(const GemmArgs& a, int epilogue) {
    // gemm_w1's operand constraints, a gate no finer than a wave's 128 rows (the direct epilogue holds two gate rows per wave), and a 256 x 256
    // grid that would leave at least half of the 256 CUs without a tile while the 256 x 128 grid still fits one round
    if (!gemm_w1_eligible(a)) return false;
    if (epilogue == FVK_EPI_RESIDUAL_GATE && a.gate && a.rows_per_batch < 128) return false;
    const long t256 = (long)((a.M + 255) / 256) * ((a.N + 255) / 256), t128 = (long)((a.M + 255) / 256) * ((a.N + 127) / 128);
    return t256 <= 128 && t128 > t256 && t128 <= 256;
}

int gemm_w1n_launch(GemmArgs a, int epilogue, int batch, hipStream_t s) {
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
