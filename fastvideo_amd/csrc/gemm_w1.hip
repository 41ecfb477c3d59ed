// bf16 GEMM for the token-axis projections, gfx950 only — "one wave per SIMD" kernel: the shipped bf16 path for K % 128 == 0
// (gemm_impl 0; gemm_impl 228 forces gemm_ph.hip, 5 + 8 * VAR a schedule variant of this file; scripts/gemm_ab.py).
//   out[M,N] = epilogue( x[M,K] · w[N,K]^T + bias )      (same contract and rounding points as gemm_ph.hip)
//
// Same 256(M) x 256(N) workgroup tile, K-step 64, 128-B swizzled LDS rows, unit FIFO and epilogue as gemm_ph.hip.  Two things change:
//   * the wave tile: FOUR waves of 128 x 128 (one per SIMD, the whole 512-entry register file each: 256 accumulator AGPRs + 128
//     fragment VGPRs) instead of eight waves of 128 x 64.  A K-tile then costs every SIMD 32 fragment reads for 2048 MFMA cycles
//     (gemm_ph: 2 x 24), and nothing alternates on a SIMD: the MFMA stream is continuous and the fragment reads / LDS-DMA pieces of the
//     NEXT phases are interleaved into it by hand;
//   * the MFMA shape: v_mfma_f32_16x16x32_bf16 (K = 32 per instruction, 4 accumulator registers) instead of 32x32x16 (K = 16, 16
//     registers).  Same FLOP per cycle, but half the accumulator traffic per FLOP — and MI355X is power-bound: a registers-only MFMA loop
//     sustains 2.40 PFLOP/s in the 16x16x32 shape against 2.13 in 32x32x16 (scripts/probes/mfma_rate.cpp, profiles/r03_mfma_shapes.md).
//     On the same schedule the shape alone is worth +6..8 % here (VAR 3 -> 7).  The summation order inside an instruction differs, so
//     this kernel agrees with gemm_ph.hip to rounding, not byte for byte (VAR 0..3, the 32x32x16 variants, are byte-identical to it).
//   * (VAR bit 3, shipped) the tile walk: 256 PERSISTENT workgroups.  The unit FIFO does not stop at a tile boundary — the last two K-tiles of a
//     tile's loop stage, and its last two phases read, the first units of the workgroup's next tile, so only a workgroup's first tile has a
//     prologue (a cold 128-KiB burst per CU otherwise); the epilogue stores straight from the accumulators (w1_direct_epilogue: no LDS, so it
//     cannot collide with the running stream); and the 256 tiles in flight are one run of consecutive tile ids, 32 per XCD — 40 operand
//     panels from HBM per round instead of the 96 of eight unrelated XCD chunks (worth +10 % at the 14B shapes, which do not fit the MALL).
// Measured (same box, interleaved, TFLOP/s, gemm_ph -> this kernel | vendor library plain epilogue; profiles/r03_gemm_w1_ab2.log):
// QKV [32760,4608,1536] 1155 -> 1210 | 1136, out-proj [32760,1536,1536] 1079 -> 1153 | 1240, FFN-in [32760,8960,1536] 1104 -> 1284 | 1321,
// FFN-out [32760,1536,8960] 1221 -> 1431 | 1514, 8192^3 1396 -> 1575 | 1535, 14B QKV [75600,15360,5120] 1299 -> 1447 | 1494,
// 14B FFN-in [75600,13824,5120] 1294 -> 1455 | 1481, 14B FFN-out [75600,5120,13824] 1293 -> 1525 | 1507.
//
// A K-tile is four phases (one 64 x 64 quadrant of the wave tile x K = 64: 32 MFMAs 16x16x32, or 16 MFMAs 32x32x16):
//     phase 4t+0: X0 x W0    4t+1: X0 x W1    4t+2: X1 x W1    4t+3: X1 x W0          (X0 / X1 = the wave's first / second 64 m, W0 / W1 likewise in n)
// Units in the order their fragments are read:  U(4t) = W0(t), U(4t+1) = W1(t), U(4t+2) = X1(t), U(4t+3) = X0(t+1).   During phase p a wave
//   * runs MFMA(p) from registers,
//   * reads unit U(p+1) (8 fragments = the unit's whole K = 64) into the register set that fell free — X0 / X1 have a set each, the two
//     W sets swap roles every K-tile, hence the loop body of two K-tiles,
//   * stages one unit by LDS-DMA (4 pieces per wave) into a slot whose reads retired before the last barrier.
// Schedule (VAR bit 1, shipped): ONE barrier per two phases.  Phase p stages U(p+7) into the slot of U(p-1).  WAR: U(p-1) was read in phase
// p-2, i.e. in front of the barrier closing the previous phase pair (reads are retired with lgkmcnt(0) there).  RAW: U(p+7) is read in
// phase p+6; every wave waits for its own pieces at the barrier closing the pair before (vmcnt(16): the two younger pairs' pieces stay in
// flight, about 2000-3000 cycles of flight).  Without VAR bit 1: a barrier per phase, phase p stages U(p+8), vmcnt(24).
#include "gemm_common.h"
#include "gemm_epilogue.h"
#include "gemm_w1_epilogue.h"

namespace {

constexpr int TM = 256, TN = 256, TK = 64;
constexpr int ROWB = TK * 2;     // 128 B per LDS row
constexpr int XREG = TM * ROWB;  // 32 KiB: the x rows of a buffer; the w rows follow
constexpr int BUF = 2 * XREG;    // 64 KiB
constexpr int LDS_BYTES = fvk::EPI_LDS_BYTES;
static_assert(LDS_BYTES >= 2 * BUF, "epilogue staging must cover both buffers");

using fvk::GemmArgs;

// W1_STRIP (co-residency study only, scripts/coresidency/build_bug_strips.sh; results are WRONG, the kernel only serves as an AGGRESSOR): bit 0 no
// LDS-DMA staging in the loop, bit 1 no fragment reads in the loop, bit 2 no workgroup barriers in the loop, bit 3 no MFMAs
#ifndef W1_STRIP
#define W1_STRIP 0
#endif
#if (W1_STRIP & 8)
#define W1_MFMA(ACC, A, B) asm volatile("" : "+a"(ACC) : "v"(A), "v"(B))
#define W1_MFMA16(ACC, A, B) asm volatile("" : "+a"(ACC) : "v"(A), "v"(B))
#define W1_MFMA16Z(ACC, A, B) asm volatile("" : "=a"(ACC) : "v"(A), "v"(B))
#elif defined(FVK_W1_BUILTIN_MFMA)  // co-residency study only (scripts/probes/libfvk_bug2.so): the SAME kernel with compiler builtins
#define W1_MFMA(ACC, A, B) ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, ACC, 0, 0, 0)
#define W1_MFMA16(ACC, A, B) ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A, B, ACC, 0, 0, 0)
#define W1_MFMA16Z(ACC, A, B) ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A, B, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0)
#else
#define W1_MFMA(ACC, A, B) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(ACC) : "v"(A), "v"(B))
#define W1_MFMA16(ACC, A, B) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(ACC) : "v"(A), "v"(B))
// first touch of an accumulator in a tile (DIRECT: no zeroing pass — 256 v_accvgpr_write per tile otherwise)
#define W1_MFMA16Z(ACC, A, B) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=a"(ACC) : "v"(A), "v"(B))
#endif
typedef int w1_v8i __attribute__((ext_vector_type(8)));
typedef int w1_v4i __attribute__((ext_vector_type(4)));
// fp8: the two 16-B halves of each operand's 32-B fragment; unit block scales (E8M0 0x7F = 1.0)
#define W1_MFMA_FP8(ACC, A0, A1, B0, B1)                                                                                           \
    {                                                                                                                              \
        const w1_v4i a0_ = __builtin_bit_cast(w1_v4i, A0), a1_ = __builtin_bit_cast(w1_v4i, A1);                                   \
        const w1_v4i b0_ = __builtin_bit_cast(w1_v4i, B0), b1_ = __builtin_bit_cast(w1_v4i, B1);                                   \
        const w1_v8i wa_ = {a0_[0], a0_[1], a0_[2], a0_[3], a1_[0], a1_[1], a1_[2], a1_[3]};                                       \
        const w1_v8i xa_ = {b0_[0], b0_[1], b0_[2], b0_[3], b1_[0], b1_[1], b1_[2], b1_[3]};                                       \
        ACC = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wa_, xa_, ACC, 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);                 \
    }
#define W1_MFMA_FP8Z(ACC, A0, A1, B0, B1)                                                                                          \
    {                                                                                                                              \
        const w1_v4i a0_ = __builtin_bit_cast(w1_v4i, A0), a1_ = __builtin_bit_cast(w1_v4i, A1);                                   \
        const w1_v4i b0_ = __builtin_bit_cast(w1_v4i, B0), b1_ = __builtin_bit_cast(w1_v4i, B1);                                   \
        const w1_v8i wa_ = {a0_[0], a0_[1], a0_[2], a0_[3], a1_[0], a1_[1], a1_[2], a1_[3]};                                       \
        const w1_v8i xa_ = {b0_[0], b0_[1], b0_[2], b0_[3], b1_[0], b1_[1], b1_[2], b1_[3]};                                       \
        const f32x4 z_ = {0.f, 0.f, 0.f, 0.f};                                                                                     \
        ACC = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wa_, xa_, z_, 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);                  \
    }

// FP8 (fvk_gemm_fp8, VAR 15 only): x and w are OCP e4m3 bytes; an LDS row is still 128 B = 128 k, a K-tile is ONE k-step of
// v_mfma_scale_f32_16x16x128_f8f6f4 (unit block scales; 2x the bf16 MFMA rate), a fragment is 32 B per lane = two 16-B chunk reads, and the
// dequantisation scales are applied in the epilogue (gemm_pp.hip's FP8 contract).
template <int EPI, int VAR, bool FP8 = false>
__global__ __launch_bounds__(256, 1) void gemm_w1_kernel(GemmArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef __attribute__((address_space(3))) void lds_void;
    FVK_CLAIM_WHOLE_REGISTER_FILE();
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const int wm = wave & 1, wn = wave >> 1;  // wave tile: rows wm*128.., cols wn*128..

    a.x += blockIdx.y * a.x_bstride;
    a.w += blockIdx.y * a.w_bstride;
    a.out += blockIdx.y * a.out_bstride;
    constexpr bool MI16 = (VAR & 4) != 0;
    constexpr bool DIRECT = MI16 && (VAR & 8) != 0;  // direct epilogue + persistent workgroups (each walks tiles vb, vb + gridDim.x, ...)
    // ---- tile id: XCD-contiguous (block b runs on XCD b % 8), then groups of 8 m-tiles swept along n (as gemm_ph.hip) --------
    const int ntiles = a.ntm * a.ntn;
    // DIRECT: 256 persistent workgroups; in round rho workgroup b (XCD b & 7) takes tile 256 rho + 32 (b & 7) + (b >> 3): the 256 tiles in flight
    // are ONE run of consecutive ids (8 m-tiles x 32 n-tiles: 40 operand panels from HBM per round instead of the 96 of eight unrelated XCD
    // chunks), of which every XCD holds 32 consecutive ones (8 m x 4 n: 12 panels through its L2).  vb = the tile id.
    int vb = DIRECT ? 32 * (int)(blockIdx.x & 7) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    if (DIRECT && vb >= ntiles) return;  // fewer tiles than workgroups (workgroup-uniform, before any barrier)
    int m0, n0;
    auto tile_coords = [&](int bid) {
        const int nwg = ntiles;
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
        const int tile_id = DIRECT ? bid : (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
        constexpr int GM = 8;
        const int per_group = GM * a.ntn;
        const int gid = tile_id / per_group;
        const int first_m = gid * GM;
        const int gsz = (a.ntm - first_m) < GM ? (a.ntm - first_m) : GM;
        const int in_g = tile_id - gid * per_group;
        m0 = (first_m + in_g % gsz) * TM;
        n0 = (in_g / gsz) * TN;
    };
    tile_coords(vb);

    // ---- LDS-DMA staging: every wave stages 32 consecutive rows (4 pieces of 8 rows x 128 B) of each unit -----------------------
    //   X0: rows (wave>>1)*128 + (wave&1)*32    X1: + 64        W0 / W1: the same rows of the w panel
    const int row0 = (wave >> 1) * 128 + (wave & 1) * 32;
    constexpr int ES = FP8 ? 1 : 2;  // bytes per operand element
    static_assert(!FP8 || (VAR & 12) == 12, "the fp8 path exists in the shipped (direct, persistent) configuration only");
    const long xld = a.lda * ES, wld = (long)a.K * ES;  // row pitch in bytes
    // one descriptor per unit kind, based at the wave's first row of that kind; rows past the operand's valid rows read as zeros
    auto mk = [](const unsigned char* base, long ld, int r0, int valid, int K) {
        const long nrec = valid > r0 ? ((long)(valid - r0) - 1) * ld + (long)K * ES : 0;
        return __builtin_amdgcn_make_buffer_rsrc((void*)(base + (long)r0 * ld), 0, (int)nrec, 0x00020000);
    };
    __amdgpu_buffer_rsrc_t r_x0, r_x1, r_w0, r_w1;      // the tile being staged
    __amdgpu_buffer_rsrc_t rn_x0, rn_x1, rn_w0, rn_w1;  // DIRECT: the workgroup's NEXT tile (the staging stream runs on into it)
    auto tile_descriptors = [&](int m0_, int n0_, __amdgpu_buffer_rsrc_t& x0_, __amdgpu_buffer_rsrc_t& x1_, __amdgpu_buffer_rsrc_t& w0_,
                                __amdgpu_buffer_rsrc_t& w1_) {
        int xvalid = a.M - m0_, wvalid = a.N - n0_;
        xvalid = xvalid > 256 ? 256 : xvalid;
        wvalid = wvalid > 256 ? 256 : wvalid;
        const unsigned char* xbase = (const unsigned char*)a.x + (long)m0_ * xld;
        const unsigned char* wbase = (const unsigned char*)a.w + (long)n0_ * wld;
        x0_ = mk(xbase, xld, row0, xvalid, a.K);
        x1_ = mk(xbase, xld, row0 + 64, xvalid, a.K);
        w0_ = mk(wbase, wld, row0, wvalid, a.K);
        w1_ = mk(wbase, wld, row0 + 64, wvalid, a.K);
    };
    tile_descriptors(m0, n0, r_x0, r_x1, r_w0, r_w1);
    int kb = 0;  // DIRECT: byte offset added to a staged K-tile's source offset (-K-extent once the stream has moved on to the next tile)
    // lane -> (row r = lane>>3 of the piece, LDS chunk position lane&7); piece i holds rows 8i + r.  LDS chunk position c' of tile
    // row R holds source chunk c' ^ ((R >> 1) & 7); row0 is a multiple of 32, so (R >> 1) & 7 = 4(i & 1) + (r >> 1).
    // DIRECT: the w rows use c' ^ (((R >> 1) & 1) | (((R >> 3) & 3) << 1)) = c' ^ (((r >> 1) & 1) | (i << 1)) instead: a w fragment then reads rows
    // {0-3, 8-11, 16-19, 24-27} (+4) of a 32-row group (see w1_direct_epilogue), which that swizzle spreads over all 16 (parity, position) slots.
    int xv[4], wv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = lane >> 3;
        const int c = (lane & 7) ^ (4 * (i & 1) + (r >> 1));
        const int cw = DIRECT ? (lane & 7) ^ (((r >> 1) & 1) | (i << 1)) : c;
        xv[i] = (int)((long)(8 * i + r) * xld) + c * 16;
        // V^T form: LDS row R of a w unit takes SOURCE row swap23(R) (bits 2 and 3 of the row index exchanged; row0 is a multiple of 32, so the
        // exchange stays inside the wave's 32 rows) — the swizzle is a function of the LDS row and does not change
        const int rw = a.w_row_perm ? (((8 * i + r) & ~12) | ((r & 4) << 1) | ((i & 1) << 2)) : 8 * i + r;
        wv[i] = (int)((long)rw * wld) + cw * 16;
    }
    const int d_x0 = row0 * ROWB, d_x1 = (row0 + 64) * ROWB;
    const int d_w0 = XREG + row0 * ROWB, d_w1 = XREG + (row0 + 64) * ROWB;
    const int nt = a.K * ES / ROWB;  // K-tiles of 128 B per row: 64 bf16 / 128 fp8 elements

    // kinds: 0 = X0, 1 = W0, 2 = W1, 3 = X1.  One piece (PC) of unit KIND of K-tile TILE; tiles past the end re-read tile 0 (never consumed)
#define W1_STAGE(KIND, TILE, PC)                                                                                      \
    {                                                                                                                 \
        const int t_ = (TILE);                                                                                        \
        const int so_ = __builtin_amdgcn_readfirstlane(DIRECT ? t_ * ROWB + kb : (t_ < nt ? t_ * ROWB : 0));          \
        unsigned char* d_ = smem + (t_ & 1) * BUF + ((KIND) == 0 ? d_x0 : (KIND) == 1 ? d_w0 : (KIND) == 2 ? d_w1 : d_x1) + (PC) * 1024; \
        if ((KIND) == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(r_x0, (lds_void*)(d_), 16, xv[PC], so_, 0, 0);      \
        else if ((KIND) == 1) __builtin_amdgcn_raw_ptr_buffer_load_lds(r_w0, (lds_void*)(d_), 16, wv[PC], so_, 0, 0); \
        else if ((KIND) == 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(r_w1, (lds_void*)(d_), 16, wv[PC], so_, 0, 0); \
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(r_x1, (lds_void*)(d_), 16, xv[PC], so_, 0, 0);                  \
    }
#define W1_STAGE_UNIT(KIND, TILE) W1_STAGE(KIND, TILE, 0) W1_STAGE(KIND, TILE, 1) W1_STAGE(KIND, TILE, 2) W1_STAGE(KIND, TILE, 3)

    // ---- fragment read offsets (bytes within a buffer).  32x32x16 fragments: row R = l31, k16-step ks, half hi -> chunk (2ks + hi);
    // 16x16x32 fragments (MI16): row R = lane & 15, k32-step ks, quarter q = lane >> 4 -> chunk (4ks + q).  LDS chunk = chunk ^ ((R>>1)&7);
    // a ds_read_b128 is served 16 lanes a cycle over 64 banks: 16 consecutive rows of one chunk are conflict-free in both shapes.
    constexpr int NKS = MI16 ? 2 : 4, NBLK = MI16 ? 4 : 2, BLKR = MI16 ? 16 : 32;  // k-steps per K-tile, row blocks per unit, rows per block
    int xo[4], wo[4];
    {
        const int rl = MI16 ? (lane & 15) : l31;
        const int sw = (rl >> 1) & 7;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            // (FP8: `ks` is the 16-B half of the lane's 32-B fragment: chunks 2q, 2q + 1)
            const int co = ((FP8 ? 2 * (lane >> 4) + ks : MI16 ? 4 * ks + (lane >> 4) : 2 * ks + hi) ^ sw) << 4;
            xo[ks] = (wm * 128 + rl) * ROWB + co;
            wo[ks] = XREG + (wn * 128 + rl) * ROWB + co;
            if (DIRECT) {  // lane row l15 of a w tile = row 8(l15 >> 2) + (l15 & 3) of its 32-row group (+4 for the group's second tile, in W1_READ1)
                const int rho = 8 * (rl >> 2) + (rl & 3), sww = ((rl & 3) >> 1) | ((rl >> 2) << 1);
                wo[ks] = XREG + (wn * 128 + rho) * ROWB + (((FP8 ? 2 * (lane >> 4) + ks : 4 * ks + (lane >> 4)) ^ sww) << 4);
            }
        }
    }

    f32x16 acc[MI16 ? 1 : 4][MI16 ? 1 : 4];   // 32x32x16: [nb][mb]
    f32x4 acc16[MI16 ? 8 : 1][MI16 ? 8 : 1];  // 16x16x32: [nb][mb]
    if (MI16 && !DIRECT) {  // (DIRECT: a tile's first MFMA on an accumulator takes 0 as its C operand)
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc16[i & (MI16 ? 7 : 0)][j & (MI16 ? 7 : 0)][r] = 0.f;
    } else if (!MI16) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i & (MI16 ? 0 : 3)][j & (MI16 ? 0 : 3)][r] = 0.f;
    }

    bf16x8 XA[8], XB[8], WA[8], WB[8];  // fragment sets, in the order a phase consumes them: [ks * NBLK + block]

    // fragment J (0..7: block J % NBLK, k-step J / NBLK) of a unit: IS_X selects the x / w rows, ROW0 (0 / 64) the unit's first row in the wave tile
    // (FP8: fragment J = 16-B half J & 1 of block J >> 1)
#define W1_KS(J) (FP8 ? (J) & 1 : (J) / NBLK)
#define W1_BLK(J) (FP8 ? (J) >> 1 : (J) % NBLK)
#define W1_READ1(DST, BUFP, IS_X, ROW0, J) \
    DST[J] = *reinterpret_cast<const bf16x8*>((BUFP) + ((IS_X) ? xo[W1_KS(J)] : wo[W1_KS(J)]) + \
                                              ((ROW0) + ((DIRECT && !(IS_X)) ? 32 * (W1_BLK(J) >> 1) + 4 * (W1_BLK(J) & 1) : W1_BLK(J) * BLKR)) * ROWB);
#define W1_PHASE_END(P)                                                 \
    if (!(VAR & 2)) {                                                   \
        asm volatile("s_waitcnt vmcnt(24) lgkmcnt(0)" ::: "memory");    \
        __builtin_amdgcn_sched_barrier(0);                              \
        if (!(W1_STRIP & 4)) __builtin_amdgcn_s_barrier();              \
        __builtin_amdgcn_sched_barrier(0);                              \
    } else if ((P) & 1) {                                               \
        asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)" ::: "memory");    \
        __builtin_amdgcn_sched_barrier(0);                              \
        if (!(W1_STRIP & 4)) __builtin_amdgcn_s_barrier();              \
        __builtin_amdgcn_sched_barrier(0);                              \
    }
    // One 64 x 64 quadrant x K = 64: 16 MFMAs 32x32x16 (ks outer, then 2 x 2 blocks) or 32 MFMAs 16x16x32 (ks outer, then 4 x 4 blocks, the
    // w fragment held over 4 MFMAs), with the unit read and the unit stage between them at 16 SLOTS.  VAR bit 0: the 8 fragment reads fill
    // slots 0..7 and the 4 pieces slots 8, 10, 12, 14 (else a read in every odd slot, a piece in every fourth); VAR bit 1: one barrier
    // per TWO phases (phase p stages U(p+7) and the wait is vmcnt(16)).  P = phase number within the K-tile; NQ / MQ = the quadrant.
#define W1_PHASE(P, PTV, ZF, XS, WS, NQ, MQ, RDST, RBUF, RIS_X, ROW0)                                   \
    _Pragma("unroll") for (int i_ = 0; i_ < ((MI16 && !FP8) ? 32 : 16); ++i_) {                         \
        if (FP8) {                                                                                      \
            const int nb_ = i_ >> 2, mb_ = i_ & 3;                                                      \
            if (ZF) W1_MFMA_FP8Z(acc16[((NQ) * 4 + nb_) & (MI16 ? 7 : 0)][((MQ) * 4 + mb_) & (MI16 ? 7 : 0)], WS[2 * nb_], WS[2 * nb_ + 1], XS[2 * mb_], XS[2 * mb_ + 1]) \
            else W1_MFMA_FP8(acc16[((NQ) * 4 + nb_) & (MI16 ? 7 : 0)][((MQ) * 4 + mb_) & (MI16 ? 7 : 0)], WS[2 * nb_], WS[2 * nb_ + 1], XS[2 * mb_], XS[2 * mb_ + 1]) \
        } else if (MI16) {                                                                              \
            const int ks_ = i_ >> 4, nb_ = (i_ >> 2) & 3, mb_ = i_ & 3;                                 \
            if ((ZF) && ks_ == 0) W1_MFMA16Z(acc16[((NQ) * 4 + nb_) & (MI16 ? 7 : 0)][((MQ) * 4 + mb_) & (MI16 ? 7 : 0)], WS[ks_ * 4 + nb_], XS[ks_ * 4 + mb_]); \
            else W1_MFMA16(acc16[((NQ) * 4 + nb_) & (MI16 ? 7 : 0)][((MQ) * 4 + mb_) & (MI16 ? 7 : 0)], WS[ks_ * 4 + nb_], XS[ks_ * 4 + mb_]); \
        } else {                                                                                        \
            const int ks_ = i_ >> 2, nb_ = (i_ >> 1) & 1, mb_ = i_ & 1;                                 \
            W1_MFMA(acc[((NQ) * 2 + nb_) & (MI16 ? 0 : 3)][((MQ) * 2 + mb_) & (MI16 ? 0 : 3)], WS[(ks_ * 2 + nb_) & 7], XS[(ks_ * 2 + mb_) & 7]); \
        }                                                                                               \
        constexpr int sk_ = (VAR & 2) ? ((P) == 0 ? 0 : (P) == 1 ? 1 : (P) == 2 ? 2 : 3) : ((P) == 0 ? 1 : (P) == 1 ? 2 : (P) == 2 ? 3 : 0); \
        const int st_ = t + (PTV) + 2 + ((!(VAR & 2) && (P) == 3) ? 1 : 0);                             \
        if (!MI16 || FP8 || (i_ & 1)) {                                                                 \
            const int sl_ = (MI16 && !FP8) ? i_ >> 1 : i_;                                              \
            if (VAR & 1) {                                                                              \
                if (sl_ < 8) { if (!(W1_STRIP & 2)) { W1_READ1(RDST, RBUF, RIS_X, ROW0, sl_ & 7) } }    \
                else if ((sl_ & 1) == 0) { if (!(W1_STRIP & 1)) W1_STAGE(sk_, st_, ((sl_ - 8) >> 1) & 3) } \
            } else {                                                                                    \
                if ((sl_ & 3) == 0) { if (!(W1_STRIP & 1)) W1_STAGE(sk_, st_, sl_ >> 2) }               \
                if (sl_ & 1) { if (!(W1_STRIP & 2)) { W1_READ1(RDST, RBUF, RIS_X, ROW0, sl_ >> 1) } }   \
            }                                                                                           \
        }                                                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                              \
    }                                                                                                   \
    W1_PHASE_END(P)

    // ---- prologue: both K-tiles' units staged in read order, X0(0) and W0(0) landed and read, X0(2) staged ("phase -1") --------
    W1_STAGE_UNIT(0, 0) W1_STAGE_UNIT(1, 0) W1_STAGE_UNIT(2, 0) W1_STAGE_UNIT(3, 0)
    W1_STAGE_UNIT(0, 1) W1_STAGE_UNIT(1, 1) W1_STAGE_UNIT(2, 1) W1_STAGE_UNIT(3, 1)
    bool first_tile = true;
    while (true) {  // tile loop (DIRECT: persistent; otherwise one pass)
    // DIRECT: the unit FIFO does not stop at a tile boundary — the last two K-tiles of a tile's loop stage (and its last two phases read) the
    // first units of the workgroup's NEXT tile, so only the first tile has a prologue; what remains between two tiles is the epilogue itself
    bool more = false;
    int m0n = m0, n0n = n0;
    if (DIRECT) {
        more = vb + 256 < ntiles;  // workgroup-uniform
        if (more) {
            const int m0c = m0, n0c = n0;
            tile_coords(vb + 256);
            m0n = m0; n0n = n0;
            m0 = m0c; n0 = n0c;
        }
        tile_descriptors(m0n, n0n, rn_x0, rn_x1, rn_w0, rn_w1);  // (no next tile: the stream re-reads this tile's first K-tiles, never consumed)
    }
    if (!DIRECT || first_tile) {
    asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 8; ++j) { W1_READ1(XA, smem, true, 0, j) }
#pragma unroll
    for (int j = 0; j < 8; ++j) { W1_READ1(WA, smem, false, 0, j) }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (!(VAR & 2)) {
        W1_STAGE_UNIT(0, 2)
        W1_PHASE_END(1)
    } else {  // phase 0 stages X0(2) itself; U(1), U(2) (read in phases 0, 1) must have landed: all but the 16 youngest pieces
        asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    }
    }  // prologue (first tile)
    first_tile = false;

    // One pair of K-tiles.  In the pair that holds the tile's last K-tiles (t == nt - 2) every staged unit already belongs to the workgroup's NEXT
    // tile (K-tiles nt, nt+1 = its K-tiles 0, 1), and the gated-residual epilogue's residual rows are touched (256 rows x 512 B = 4 cache
    // lines a row; lane = row, one dword per line by LDS-DMA into a 4-KiB dump area behind the ring — no destination register to keep alive,
    // rows / columns outside the matrix fall outside the descriptor and fetch nothing): written a whole layer ago, they are then found in L2 by
    // the epilogue's own loads.  Phases: X0 x W0 (reads W1(t)), X0 x W1 (X1(t)), X1 x W1
    // (X0(t+1)), X1 x W0 (W0(t+1)); then the same on K-tile t+1 with the W sets swapped.  ZF_: the first K-tile's MFMAs take 0 as C.
#define W1_ITER(ZF_) \
        if (DIRECT && t == nt - 2) { \
            r_x0 = rn_x0; r_x1 = rn_x1; r_w0 = rn_w0; r_w1 = rn_w1; \
            kb = -nt * ROWB; \
            if (EPI == FVK_EPI_RESIDUAL_GATE) { \
                int rv_ = a.M - m0, cv_ = a.N - n0; \
                rv_ = rv_ > 256 ? 256 : rv_; cv_ = cv_ > 256 ? 256 : cv_; \
                if (rv_ > 0 && cv_ > 0) { \
                    const __amdgpu_buffer_rsrc_t rr_ = __builtin_amdgcn_make_buffer_rsrc((void*)(a.residual + (long)m0 * a.ldc + n0), 0, \
                                                                                         (int)((((long)rv_ - 1) * a.ldc + cv_) * 2), 0x00020000); \
                    const int vo_ = (wave * 64 + lane) * (int)(a.ldc * 2); \
                    _Pragma("unroll") for (int sg = 0; sg < 4; ++sg) \
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(rr_, (lds_void*)(smem + 2 * BUF + wave * 1024 + sg * 256), 4, vo_, sg * 128, 0, 0); \
                } \
            } \
        } \
        const unsigned char* b0 = smem + (t & 1) * BUF; \
        const unsigned char* b1 = smem + ((t + 1) & 1) * BUF; \
        W1_PHASE(0, 0, ZF_, XA, WA, 0, 0, WB, b0, false, 64) \
        W1_PHASE(1, 0, ZF_, XA, WB, 1, 0, XB, b0, true, 64) \
        W1_PHASE(2, 0, ZF_, XB, WB, 1, 1, XA, b1, true, 0) \
        W1_PHASE(3, 0, ZF_, XB, WA, 0, 1, WB, b1, false, 0) \
        W1_PHASE(0, 1, 0, XA, WB, 0, 0, WA, b1, false, 64) \
        W1_PHASE(1, 1, 0, XA, WA, 1, 0, XB, b1, true, 64) \
        W1_PHASE(2, 1, 0, XB, WA, 1, 1, XA, b0, true, 0) \
        W1_PHASE(3, 1, 0, XB, WB, 0, 1, WA, b0, false, 0)
    // K-tile pairs (nt is even: gemm_w1_eligible); K-tile t: W0 in WA, W1 in WB, K-tile t+1 the other way round.  DIRECT: the first pair is
    // peeled — its first K-tile's MFMAs are each accumulator's first touch in the tile and take 0 as C (no zeroing pass)
    if constexpr (DIRECT) {
        { const int t = 0; W1_ITER(1) }
        for (int t = 2; t < nt; t += 2) { W1_ITER(0) }
    } else {
        for (int t = 0; t < nt; t += 2) { W1_ITER(0) }
    }
#undef W1_ITER
    if constexpr (DIRECT) {
        // the accumulators were written by asm MFMAs: the compiler knows no hazard distance to its own reads of them
        asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) asm volatile("" : "+a"(acc16[i][j]));
        if (!(VAR & 16)) fvk::w1_direct_epilogue<EPI, FP8, (VAR & 128) != 0>(a, acc16, m0, n0, wm, wn, lane);  // VAR bit 4: timing ablation (no epilogue, no output)
        // everything this wave has in flight — the next tile's units and the epilogue's stores (they share vmcnt, and reads / writes need not
        // retire in issue order) — before the counted waits of the next tile's loop, or before the workgroup ends with LDS writes outstanding
#if W1_ABL != 2
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
        if (!more) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        if (!more) break;
        vb += 256;
        m0 = m0n; n0 = n0n;
        kb = 0;  // (r_* already describe this tile)
        continue;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // tail re-reads must have landed before the buffers are reused
    __builtin_amdgcn_s_barrier();
    // the accumulators were written by asm MFMAs: the compiler knows no hazard distance to its own reads of them
    asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");
    // the shared epilogue works on 128(m) x 64(n) wave tiles numbered wm + 2 * (64-column group): this wave is two of them
    if constexpr (MI16) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) asm volatile("" : "+a"(acc16[i][j]));
        fvk::gemm_tile_epilogue<EPI, false, true, true>(a, reinterpret_cast<f32x4(&)[4][8]>(acc16[0]), smem, wm + 2 * (2 * wn), lane, m0, n0);
        fvk::gemm_tile_epilogue<EPI, false, true, true>(a, reinterpret_cast<f32x4(&)[4][8]>(acc16[4]), smem, wm + 2 * (2 * wn + 1), lane, m0, n0);
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) asm volatile("" : "+a"(acc[i][j]));
        fvk::gemm_tile_epilogue<EPI, false, true>(a, reinterpret_cast<f32x16(&)[2][4]>(acc[0]), smem, wm + 2 * (2 * wn), lane, m0, n0);
        fvk::gemm_tile_epilogue<EPI, false, true>(a, reinterpret_cast<f32x16(&)[2][4]>(acc[2]), smem, wm + 2 * (2 * wn + 1), lane, m0, n0);
    }
    break;
    }  // tile loop
#undef W1_STAGE
#undef W1_STAGE_UNIT
#undef W1_READ1
#undef W1_KS
#undef W1_BLK
#undef W1_PHASE
#undef W1_PHASE_END
#endif  // __HIP_DEVICE_COMPILE__
}

template <int EPI, int VAR, bool FP8 = false>
int launch(const GemmArgs& a, int batch, hipStream_t s) {
    static FvkLdsConfigured configured;
    constexpr bool DIRECT = (VAR & 12) == 12;
    constexpr int lds = DIRECT ? 2 * BUF + (EPI == FVK_EPI_RESIDUAL_GATE ? 4096 : 0) : LDS_BYTES;  // the direct epilogue needs no staging region (gated residual: + the touch loads' dump area)
    if (int rc = fvk_config_lds(configured, (const void*)gemm_w1_kernel<EPI, VAR, FP8>, lds, FP8 ? "fvk_gemm_fp8 (w1)" : "fvk_gemm_bf16 (w1)")) return rc;
    const int tiles = a.ntm * a.ntn;
    const int grid = DIRECT ? 256 : tiles;  // DIRECT: persistent, one workgroup per CU (a workgroup without a tile returns at once)
    hipLaunchKernelGGL((gemm_w1_kernel<EPI, VAR, FP8>), dim3(grid, batch), dim3(256), lds, s, a);
    FVK_LAUNCH_CHECK();
    return FVK_OK;
}

}  // namespace

namespace fvk {

bool gemm_w1_eligible(const GemmArgs& a) {
    // on top of gemm_ph_eligible: an even number of 64-element K-steps (the loop body is two K-tiles)
    return gemm_ph_eligible(a) && a.K % (2 * TK) == 0;
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

// V^T = Wv · X^T in the attention kernels' key order (fvk_gemm_vt_bf16): the shipped configuration only (direct epilogue: VAR 15)
int gemm_w1_vt_launch(GemmArgs a, int batch, hipStream_t s) {
    a.ntm = (a.M + TM - 1) / TM;
    a.ntn = (a.N + TN - 1) / TN;
    // V^T [d, S_pad] is a wide output (N = S): streaming stores, as for every N >= 4096 (gemm_w1_launch) — VAR 143
#if FVK_VARIANTS
    if (fvk::tunable(fvk::TUNE_GEMM_IMPL) == 2) return launch<FVK_EPI_VT, 15>(a, batch, s);  // A/B in the step: plain stores
#endif
    return launch<FVK_EPI_VT, 143>(a, batch, s);
}

bool gemm_w1_fp8_eligible(const GemmArgs& a) {
    // whole double K-tiles of 128 fp8 elements, more than half a tile of rows, 32-bit offsets inside one 256-row panel
    return a.K % 256 == 0 && a.M > 128 && a.N % 8 == 0 && a.ldc % 8 == 0 && a.lda % 16 == 0 && 255L * a.lda + a.K <= 0x7fffffffL;
}

template <int VAR>
int launch_fp8_var(const GemmArgs& a, int epilogue, hipStream_t s) {
    switch (epilogue) {
        case FVK_EPI_NONE: return launch<FVK_EPI_NONE, VAR, true>(a, 1, s);
        case FVK_EPI_GELU_TANH: return launch<FVK_EPI_GELU_TANH, VAR, true>(a, 1, s);
        case FVK_EPI_SILU: return launch<FVK_EPI_SILU, VAR, true>(a, 1, s);
        default: return launch<FVK_EPI_RESIDUAL_GATE, VAR, true>(a, 1, s);
    }
}

int gemm_w1_fp8_launch(GemmArgs a, int epilogue, hipStream_t s) {
    a.ntm = (a.M + TM - 1) / TM;
    a.ntn = (a.N + TN - 1) / TN;
    return a.N >= 4096 ? launch_fp8_var<143>(a, epilogue, s) : launch_fp8_var<15>(a, epilogue, s);  // streaming stores for wide outputs, as the bf16 path
}

int gemm_w1_launch(GemmArgs a, int epilogue, int batch, hipStream_t s) {
    a.ntm = (a.M + TM - 1) / TM;
    a.ntn = (a.N + TN - 1) / TN;
    // shipped configuration = VAR 15 (16x16x32 MFMAs, early fragment reads, one barrier per two phases, direct epilogue + persistent
    // workgroups); gemm_impl = 5 + 8 * VAR selects a measurement variant (5 = VAR 0: 32x32x16 MFMAs — byte-identical to gemm_ph.hip;
    // 61 = VAR 7: the shipped arithmetic with the LDS-bounce epilogue and one workgroup per tile — byte-identical to the shipped kernel)
    const int impl = fvk::tunable(fvk::TUNE_GEMM_IMPL);
#if FVK_VARIANTS
    switch ((impl & 7) == 5 ? impl >> 3 : -1) {
        case 0: return launch_var<0>(a, epilogue, batch, s);
        case 1: return launch_var<1>(a, epilogue, batch, s);
        case 2: return launch_var<2>(a, epilogue, batch, s);
        case 3: return launch_var<3>(a, epilogue, batch, s);
        case 4: return launch_var<4>(a, epilogue, batch, s);
        case 5: return launch_var<5>(a, epilogue, batch, s);
        case 6: return launch_var<6>(a, epilogue, batch, s);
        case 7: return launch_var<7>(a, epilogue, batch, s);
        case 31: return launch_var<31>(a, epilogue, batch, s);  // timing ablation: no epilogue
        case 143: return launch_var<143>(a, epilogue, batch, s);  // streaming (nontemporal) output stores
        case 15: return launch_var<15>(a, epilogue, batch, s);
        default: break;
    }
#endif
    (void)impl;
    // a gate finer than 128 rows (per-token modulation): the direct epilogue holds two gate rows per wave, the LDS-bounce variant any number
    if (epilogue == FVK_EPI_RESIDUAL_GATE && a.gate && a.rows_per_batch < 128) return launch_var<7>(a, epilogue, batch, s);
    // streaming output stores where the output is wide (N >= 4096: QKV, FFN-in, every 14B projection): +4 % on QKV, +6-10 % at the 14B shapes
    // (the freshly written tile does not push the operand panels out of L2 / MALL); -1 % where the output is the narrow residual stream
    int nt_min = 4096;
#if FVK_VARIANTS
    if (impl == 3) nt_min = 3072;  // A/B: + the q | k projection (N = 3072 since the V^T GEMM left the fused QKV launch)
    if (impl == 4) nt_min = 0;     // A/B: every output streamed
    if (impl == 6) nt_min = 1 << 30;  // A/B: no streaming stores at all
#endif
    return a.N >= nt_min ? launch_var<143>(a, epilogue, batch, s) : launch_var<15>(a, epilogue, batch, s);
}

}  // namespace fvk
