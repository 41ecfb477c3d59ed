// 3x3-spatial causal convolution of the Wan VAE decoder, "one wave per SIMD" form (round 4) — the shipped kernel for every 3x3x3 / 1x3x3 conv
// with a bf16 output (96 -> 96 at full resolution, 192 -> 192, 384 -> 384, and the 2x-upsampling resample convs 384 -> 192, 192 -> 96, whose
// nearest-exact upsample is folded into the slab staging: 92 % of a decode's conv time); conv_out (3 channels, fp32 planar) and the measurement
// build's A/B kernels stay in vae_conv3.hip.  Same contract, layouts, rounding points and epilogues as vae_conv3.hip (fvk_vae_conv_bf16 /
// fvk_vae_conv_norm_bf16); the MFMA shape differs, so the two agree to rounding (fp32 summation order inside an instruction), not byte for byte.
//
// Why (profiles/r02_vae_conv_pmc.md): the 8-wave kernel runs at an effective 1.52 GHz with the matrix pipe 60 % busy — power-bound,
// and after the matrix pipe the LDS operand stream is the largest consumer: a (dw, 32-channel) group feeds 36 MFMAs 32x32x16 from 30
// ds_read_b128.  Here:
//   * FOUR waves, one per SIMD, each with the whole 512-entry register file: wave tile = 4 pixel rows x 32 columns x 96 output channels
//     (192 accumulator AGPRs).  A (dw, 32-channel chunk) group = 6 weight fragments + 8 pixel fragments -> 48 MFMAs: 14 reads per 393 k MACs
//     instead of 30 per 590 k (-30 % LDS operand bytes per FLOP), and nothing alternates on a SIMD;
//   * v_mfma_f32_16x16x32_bf16 (K = 32 = one channel chunk per instruction, 4 accumulator registers): a fifth less register-file traffic per
//     FLOP than 32x32x16 — on this power-bound part worth +6..8 % in gemm_w1 and attn_w16 (profiles/r03_mfma_shapes.md);
//   * fragments are read TWO groups ahead into three rotating register sets (3 groups per K-step, so every group position owns one set and the
//     loop body is ONE slab = 3 K-steps with every tap offset an instruction immediate); the step barrier sits after the step's FIRST group, so
//     no group ever starts with a cold fragment read (the 8-wave kernel drains the pipe at every step start: s_memtime 3300 cycles per 2304 of
//     MFMA), and the DMA pieces (weights of step u+2, the next halo slab) are issued inside the step's second group, two groups before their wait;
//   * LDS rows stay 64 B (one pixel / one weight row x 32 channels) with the 16-B chunk c of row R stored at position c ^ 2((R >> 2) & 1): a
//     16x16x32 fragment read (lane = row R0 + (lane & 15), chunk lane >> 4) is then bank-conflict-free at EVERY row offset R0 — the dw = 0, 1, 2
//     taps are fragment reads of one staged slab at pixel offsets 0, 1, 2 (brute-forced over the ds_read_b128 lane groups of
//     guides/MI355X_MICROARCH.md; the 8-wave kernel's swizzle (R >> 2) & 3 is 2-way conflicted in this shape).  Halo rows are pitched 40
//     pixels (34 used) so that a row step (2560 B) leaves the swizzle bit alone;
//   * weight rows are staged in MFMA order (LDS row 16 T + i of a dw block = channel 32 (T >> 1) + 8 (i >> 2) + (i & 3) + 4 (T & 1)), which leaves lane
//     (pixel l15, g) with EIGHT consecutive output channels per tile pair: the epilogue stores 16 B per lane straight from the accumulators
//     (bias / residual / fused RMS-norm + SiLU in registers: the row sum of squares is a 24-term lane sum + two cross-lane adds) — no LDS
//     staging passes (the 8-wave kernel's fused norm: three LDS passes over a wave-private tile).
// ref: WanCausalConv3d.forward (fastvideo/models/vaes/wanvae.py:198-207), WanResidualBlock (:418-431, :462), WanRMS_norm (:231-232).
#include "fvk_common.h"
#include "vae_conv3_args.h"

int fvk_vae_conv_tunable();  // vae_conv.hip: the "vae_conv_impl" measurement switch

namespace {

using fvkc3::Conv3Args;
using fvkc3::EPI_BIAS;
using fvkc3::EPI_RESIDUAL;
using fvkc3::OOB;

#define C3W_MFMA(ACC, A, B) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(ACC) : "v"(A), "v"(B))

template <int WNW, int EPI>
__global__ __launch_bounds__(256, 1) void vae_conv3w_kernel(Conv3Args a, int ntiles, int stagger) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int TH = WNW == 1 ? 16 : 8, TW = 32;      // workgroup tile in pixels (4 waves x 4 rows, or 2 row groups x 2 channel halves)
    constexpr int TN = WNW * 96;                        // output channels per workgroup
    constexpr int HH = TH + 2, WW = 40, WWV = TW + 2;   // halo slab: HH rows pitched WW pixels, WWV of them used
    constexpr int ROWB = WW * 64;                       // bytes per halo row (2560: a multiple of 512, so (p >> 2) & 1 is row-invariant)
    constexpr int XPIECES = HH * WW / 16;               // 16-pixel DMA pieces per slab: 45 / 25
    constexpr int XS = (XPIECES + 3) / 4;               // per wave: 12 / 7 (the surplus slots of the later waves are skipped)
    constexpr int XS0 = (XS + 1) / 2, XS1 = XS - XS0;   // issued during dh = 0 / dh = 1 of the previous slab
    constexpr int SLAB = XPIECES * 1024;
    constexpr int WT = TN / 16;                         // 16-row weight tiles per dw block: 6 / 12
    constexpr int WPIECES = 3 * WT;                     // 18 / 36
    constexpr int WS = (WPIECES + 3) / 4;               // per wave: 5 / 9
    constexpr int WSTEP = WPIECES * 1024;
    constexpr int W_BASE = 2 * SLAB;
    constexpr int XCH = W_BASE + 3 * WSTEP;             // WNW = 2: 2 KiB for the partner waves' partial sums of squares (fused norm)
    static_assert(HH * WW % 16 == 0 && XCH + (WNW == 2 ? 2048 : 0) <= 160 * 1024, "LDS budget");
    static_assert(WS + (XS0 > XS1 ? XS0 : XS1) <= 16, "DMA issue slots of a step's second group exhausted");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef __attribute__((address_space(3))) void lds_void;

    FVK_CLAIM_WHOLE_REGISTER_FILE();
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, qk = lane >> 4;
    const int wrow = WNW == 1 ? wave : (wave >> 1), wn = WNW == 1 ? 0 : (wave & 1);  // wave: tile rows 4 wrow .. 4 wrow + 3; channels 96 wn ..

#if FVK_VARIANTS  // measurement build: with EPI_BIAS and a non-null out_f32, every workgroup's wave 0 stamps s_memtime at kernel entry, loop
    // start, loop end and tile end of its FIRST tile into out_f32 (as uint64 [workgroups][4]) — where a tile's time goes (scripts/conv3w_probe.py)
    unsigned long long stamp_[4] = {0, 0, 0, 0};
#define C3W_STAMP(K) if (EPI == EPI_BIAS && a.out_f32) stamp_[K] = __builtin_readcyclecounter();
#else
#define C3W_STAMP(K)
#endif
    C3W_STAMP(0)

    // ---- tiles: PERSISTENT workgroups — workgroup b computes the tiles of linear ids b, b + gridDim.x, ...  The DMA stream does not stop at a
    // tile boundary: the last slab's steps stage the NEXT tile's first slab and weight steps exactly as they would the next slab of the same
    // tile, so only a workgroup's first tile pays the cold prologue burst (every CU fetching 81 KiB at once: 13 k cycles of a 98 k-cycle tile at
    // 96 channels, profiles/r04b_conv3w_tile_probe.log).
    // Linear id -> tile, XCD-AWARE.  Workgroup b runs on XCD b & 7 (round-robin dispatch; gridDim.x is a multiple of 8 whenever a workgroup has
    // more than one tile), and every XCD has its own L2.  An output frame reads three input frames, so the tile (s, t) of spatial position s
    // shares two thirds of its input with (s, t +- 1): the tiles are ordered spatial-major with t FASTEST (m = s T + t), that sequence is cut
    // into 8 contiguous runs, and XCD x walks run x — the 32 tiles an XCD has in flight are then ~2 spatial positions x 16 frames, every input
    // slab is fetched into that XCD's L2 once and hit by the other two frames' tiles.  (The first form — n, w, h, t with t slowest, consecutive
    // ids on different XCDs — fetched every input frame three times: 2.6 x the algorithmic HBM bytes, profiles/r04z_conv3w_traffic.json.)
    struct Tile { int t_out, h0, w0, n0; };
    const int id_base = ntiles >> 3, id_rem = ntiles & 7;
    auto decode = [&](int id) {
        Tile t;
        const int x = id & 7, j = id >> 3;
        const int m = x * id_base + (x < id_rem ? x : id_rem) + j;   // XCD x owns ids x, x + 8, ...: id_base (+ 1 if x < id_rem) of them
        int sp = m / a.T;
        t.t_out = m - sp * a.T;
        const int pid_n = sp % a.ntn; sp /= a.ntn;
        const int tw_i = sp % a.tiles_w;
        const int th_i = sp / a.tiles_w;
        t.h0 = th_i * TH; t.w0 = tw_i * TW; t.n0 = pid_n * TN;
        return t;
    };
    int cur_id = (int)blockIdx.x;
    Tile cur = decode(cur_id);
    if (stagger > 0) {
        // Phase stagger (persistent launches with many tiles per workgroup): all tiles take the same time, so the 256 workgroups would run in
        // lockstep for the whole launch and hit their epilogues together — 25-50 MB of stores in a few microseconds, more than HBM takes, every
        // tile.  Delaying workgroup b by ((b >> 3) & 15) sixteenths of a tile time ONCE spreads the bursts over the tile for the rest of the launch
        // (b & 7 is the XCD: every XCD gets all sixteen phases).
        const unsigned long long t0_ = __builtin_readcyclecounter(), wait_ = (unsigned long long)(((unsigned)blockIdx.x >> 3) & 15u) * (unsigned)stagger;
        while (__builtin_readcyclecounter() - t0_ < wait_) __builtin_amdgcn_s_sleep(16);
    }

    // ---- staging: piece q = wave + 4 i; lane -> LDS row 16 q + (lane >> 2), chunk POSITION lane & 3 = source chunk (lane & 3) ^ 2 ((row >> 2) & 1) ----
    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)a.in, 0, (unsigned)((long)a.ring * a.Hin * a.Win * a.Cin * 2), 0x00020000);
    const int Ktot = a.KT * 9 * a.Cin;
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, (unsigned)((long)a.Cout * Ktot * 2), 0x00020000);
    const int chunk16 = ((lane & 3) ^ (2 * ((lane >> 4) & 1))) * 16;  // ((16 q + (lane >> 2)) >> 2) & 1 = (lane >> 4) & 1
    const int CinB = a.Cin * 2;
    // Per-lane source offsets of the pieces of the tile being STAGED; a tile past the end gets OOB offsets (zero fill, no traffic).  A wave's
    // surplus slot (piece index past the last piece: only the LAST slot of the later waves) re-issues the wave's previous piece — the same
    // bytes to the same place, so the issue stream needs neither a branch nor a scratch region.
    const bool x_dup = wave + 4 * (XS - 1) >= XPIECES, w_dup = wave + 4 * (WS - 1) >= WPIECES;   // wave-uniform
    const int x_last = x_dup ? XS - 2 : XS - 1, w_last = w_dup ? WS - 2 : WS - 1;                // slot whose destination the last slot writes
    unsigned xvo_[XS];
    const int ups = a.Hin != a.H ? 1 : 0;   // (launcher: H == 2 Hin and W == 2 Win, or equal)
    auto set_xvo = [&](const Tile& t, bool live) {
        const int hb = t.h0 - 1, wb = t.w0 - 1;  // input coordinates of slab pixel (0, 0)
#pragma unroll
        for (int i = 0; i < XS; ++i) {
            const int p = (wave + 4 * (i == XS - 1 ? x_last : i)) * 16 + (lane >> 2);
            const int hh = p / WW, ww = p - hh * WW;
            const int h = hb + hh, w = wb + ww;
            const bool ok = live && hh < HH && ww < WWV && (unsigned)h < (unsigned)a.H && (unsigned)w < (unsigned)a.W;
            // the 2x nearest-exact upsample of the decoder's resample convs (ref: WanUpsample, wanvae.py:245-250) folded into the staging: slab pixel
            // (h, w) of the upsampled frame is input pixel (h >> 1, w >> 1) — the slab holds upsampled pixels, the kernel below does not know
            xvo_[i] = ok ? (unsigned)((((h >> ups) * a.Win + (w >> ups)) * CinB)) + chunk16 : OOB;
        }
    };
    unsigned wvo_[WS];   // weight pieces: they depend on the tile's n-tile only; the piece's dw tap offset is folded in
    auto set_wvo = [&](int n0, bool live) {
#pragma unroll
        for (int j = 0; j < WS; ++j) {
            const int q = wave + 4 * (j == WS - 1 ? w_last : j);
            const int dw = q / WT, tq = q - dw * WT;                 // tile tq of the dw block: wave half tq / 6, MFMA tile tq % 6
            const int t6 = tq % 6, i = lane >> 2;
            const int n = n0 + 96 * (tq / 6) + 32 * (t6 >> 1) + 8 * (i >> 2) + (i & 3) + 4 * (t6 & 1);   // MFMA row order (see the header)
            wvo_[j] = (live && n < a.Cout) ? (unsigned)((long)n * Ktot * 2) + (unsigned)(dw * a.Cin * 2) + chunk16 : OOB;
        }
    };
    set_xvo(cur, true);
    set_wvo(cur.n0, true);
    const int cpt = a.Cin / 32;
    const int nslab = a.KT * cpt, nstep = nslab * 3;
    const unsigned frameB = (unsigned)(a.Hin * a.Win * CinB);
    // Scalar issue state, advanced once per slab / per step and across tiles.  Slabs: tile xs_id (frame xs_t), slab xn_s = (xn_dt, xn_cc), source
    // offset x_so, LDS buffer offset x_dst (alternates with every slab, tile boundaries included).  Weights: tile ws_id, step wn_u =
    // (wn_dt, wn_cc, wn_dh), source offset w_so; ring slot = wn_dh (compile-time at every issue site).
    auto slot_of = [&](int t_out, int dt) { int sl = a.ring_start + t_out + dt; return sl >= a.ring ? sl - a.ring : sl; };
    int xs_id = cur_id, xs_t = cur.t_out, xn_dt = 0, xn_cc = 0, xn_s = 0;
    unsigned x_so = (unsigned)slot_of(xs_t, 0) * frameB;
    int x_dst = wave * 1024;                         // + piece slot * 4096; the other buffer: + SLAB
    int ws_id = cur_id, wn_dt = 0, wn_cc = 0, wn_dh = 0, wn_u = 0;
    unsigned w_so = 0;
    const int w_dst = W_BASE + wave * 1024;          // + ring slot * WSTEP + piece slot * 4096
#define C3W_ISSUE_X(I)  /* this wave's piece slot I of slab xn_s of tile xs_id */                                   \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rsrc, (lds_void*)(smem + x_dst + ((I) == XS - 1 ? x_last : (I)) * 4096), 16, xvo_[I], x_so, 0, 0);
#define C3W_ADVANCE_X()                                                                                             \
    {                                                                                                               \
        x_dst = wave * 1024 + (x_dst >= SLAB ? 0 : SLAB);  /* the other slab buffer */                              \
        x_so += 64u;                                                                                                \
        if (++xn_cc == cpt) { xn_cc = 0; ++xn_dt; x_so = (unsigned)slot_of(xs_t, xn_dt) * frameB; }                 \
        if (++xn_s == nslab) { /* the stream moves on to this workgroup's next tile */                              \
            xn_s = 0; xn_dt = 0; xn_cc = 0;                                                                         \
            xs_id += (int)gridDim.x;                                                                                \
            const bool live_ = xs_id < ntiles;                                                                      \
            const Tile nt_ = decode(live_ ? xs_id : 0);                                                             \
            xs_t = nt_.t_out;                                                                                       \
            set_xvo(nt_, live_);                                                                                    \
            x_so = (unsigned)slot_of(xs_t, 0) * frameB;                                                             \
        }                                                                                                           \
    }
#define C3W_ISSUE_W(J, SLOT)  /* this wave's piece slot J of weight step wn_u of tile ws_id into ring slot SLOT */  \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (lds_void*)(smem + w_dst + (SLOT) * WSTEP + ((J) == WS - 1 ? w_last : (J)) * 4096), 16, wvo_[J], w_so, 0, 0);
#define C3W_ADVANCE_W()                                                                                             \
    {                                                                                                               \
        if (++wn_dh == 3) { wn_dh = 0; if (++wn_cc == cpt) { wn_cc = 0; ++wn_dt; } }                                \
        if (++wn_u == nstep) {                                                                                      \
            wn_u = 0; wn_dt = 0; wn_cc = 0; wn_dh = 0;                                                              \
            ws_id += (int)gridDim.x;                                                                                \
            if (a.ntn > 1 || ws_id >= ntiles) set_wvo(decode(ws_id < ntiles ? ws_id : 0).n0, ws_id < ntiles);       \
        }                                                                                                           \
        w_so = (unsigned)((((wn_dt * 3 + wn_dh) * 3) * a.Cin + wn_cc * 32) * 2);                                    \
    }

    // ---- fragment read addresses.  Pixel fragment (row r of the wave's 4, 16-column half h, tap (dh, dw)): slab pixel
    //      p = (4 wrow + r + dh) WW + 16 h + l15 + dw, chunk qk at position qk ^ 2 (((l15 + dw) >> 2) & 1): per-lane base per dw, the rest immediates.
    //      Weight fragment (tile T, tap dw): row 16 (6 wn + T) + l15 of the dw block, same swizzle on l15.
    unsigned xo_[3];
#pragma unroll
    for (int dw = 0; dw < 3; ++dw) {
        const int c = l15 + dw;
        xo_[dw] = (unsigned)((4 * wrow) * ROWB + c * 64 + ((qk ^ (2 * ((c >> 2) & 1))) << 4));
    }
    unsigned wv_[3];  // per ring slot (the tap / tile offsets stay inside the 16-bit instruction immediate)
#pragma unroll
    for (int sl = 0; sl < 3; ++sl) wv_[sl] = (unsigned)(W_BASE + sl * WSTEP + (wn * 6) * 1024 + l15 * 64 + ((qk ^ (2 * ((l15 >> 2) & 1))) << 4));

    f32x4 acc[6][8];   // [weight tile T][pixel block pb = 2 r + h]: D[channel row 4 qk + e of tile T][pixel l15 of block pb]
    bf16x8 WF[3][6], XF[3][8];

    // ---- prologue (the workgroup's FIRST tile only): slab 0, weight steps 0 and 1 in flight ---------------------------------------------
#pragma unroll
    for (int i = 0; i < XS; ++i) C3W_ISSUE_X(i)
    C3W_ADVANCE_X()
#pragma unroll
    for (int j = 0; j < WS; ++j) C3W_ISSUE_W(j, 0)
    C3W_ADVANCE_W()
#pragma unroll
    for (int j = 0; j < WS; ++j) C3W_ISSUE_W(j, 1)
    C3W_ADVANCE_W()

    // fragment k (0..5: weight tile k; 6..13: pixel block k - 6) of the group (slab base XB, ring slot WSLOT, taps DH, DW) into register set RB
#define C3W_READ(RB, K, XB, WSLOT, DH, DW)                                                                                              \
    {                                                                                                                                   \
        if ((K) < 6) WF[RB][(K) < 6 ? (K) : 0] = *reinterpret_cast<const bf16x8*>(smem + wv_[WSLOT] + (DW) * (TN * 64) + (K) * 1024);             \
        else XF[RB][(K) >= 6 ? (K) - 6 : 0] = *reinterpret_cast<const bf16x8*>(smem + (XB)[DW] + ((((K) - 6) >> 1) + (DH)) * ROWB + (((K) - 6) & 1) * 1024); \
    }
    // One group: 48 MFMAs from register set B_ (weight tile outer, pixel block inner: an accumulator is touched once per group); after MFMA
    // 3k the k-th fragment of the group two ahead is read into set RB_; DMA_(k) fills issue slot k (after MFMA 3k + 1), 16 slots.
#define C3W_GROUP(B_, RB_, XB_, WSLOT_, DH_, DW_, DMA_)                                                             \
    _Pragma("unroll") for (int i_ = 0; i_ < 48; ++i_) {                                                             \
        C3W_MFMA(acc[i_ >> 3][i_ & 7], WF[B_][i_ >> 3], XF[B_][i_ & 7]);                                            \
        if (i_ % 3 == 0 && i_ / 3 < 14) C3W_READ(RB_, i_ / 3, XB_, WSLOT_, DH_, DW_)                                \
        if (i_ % 3 == 1) { DMA_(i_ / 3) }                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                                          \
    }
#define C3W_NO_DMA(K_)
    // second group of step DH: weights of step u + 2 (ring slot (DH + 2) % 3: read in step u - 1, which every wave left before this step's
    // barrier), then this wave's share of the next slab (during dh = 0, 1; its buffer was read in the previous slab)
#define C3W_DMA0(K_) if ((K_) < WS) C3W_ISSUE_W((K_) < WS ? (K_) : 0, 2) else if ((K_) - WS < XS0) C3W_ISSUE_X((K_) - WS < XS0 ? (K_) - WS : 0)
#define C3W_DMA1(K_) if ((K_) < WS) C3W_ISSUE_W((K_) < WS ? (K_) : 0, 0) else if ((K_) - WS < XS1) C3W_ISSUE_X((K_) - WS < XS1 ? XS0 + (K_) - WS : 0)
#define C3W_DMA2(K_) if ((K_) < WS) C3W_ISSUE_W((K_) < WS ? (K_) : 0, 1)
    // A K-step u = (slab s, dh): groups dw = 0, 1, 2 from register sets 0, 1, 2.  The step barrier sits after group 0: everything this wave
    // issued in the previous step's second group (weights of step u + 1, slab pieces) has landed by then, and group 1 reads the NEXT step's
    // first fragments behind it.
#define C3W_STEP(DH, DMA_, XNEXT_)                                                                                  \
    C3W_GROUP(0, 2, xc_, DH, DH, 2, C3W_NO_DMA)                                                                     \
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                                \
    __builtin_amdgcn_sched_barrier(0);                                                                              \
    __builtin_amdgcn_s_barrier();                                                                                   \
    __builtin_amdgcn_sched_barrier(0);                                                                              \
    C3W_GROUP(1, 0, XNEXT_, ((DH) + 1) % 3, ((DH) + 1) % 3, 0, DMA_)                                                \
    C3W_ADVANCE_W()                                                                                                 \
    C3W_GROUP(2, 1, XNEXT_, ((DH) + 1) % 3, ((DH) + 1) % 3, 1, C3W_NO_DMA)

    unsigned xc_[3], xn_[3];  // fragment bases in the slab being read / the next one (they swap with every slab, tile boundaries included)
#pragma unroll
    for (int dw = 0; dw < 3; ++dw) { xc_[dw] = xo_[dw]; xn_[dw] = xo_[dw] + SLAB; }
    const int ncol0w = wn * 96;
    const int HW = a.H * a.W;
    bool first = true;

    while (true) {  // tile loop
    // ---- accumulators zeroed (first tile: under the prologue's flight time), the tile's first pieces landed, groups 0 and 1 of step 0 read ------
#pragma unroll
    for (int t = 0; t < 6; ++t)
#pragma unroll
        for (int pb = 0; pb < 8; ++pb)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[t][pb][e] = 0.f;
    if (first) {  // (later tiles: slab 0 and weight step 0 were waited for at the previous tile's last step barrier)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < 14; ++k) C3W_READ(0, k, xc_, 0, 0, 0)
#pragma unroll
    for (int k = 0; k < 14; ++k) C3W_READ(1, k, xc_, 0, 0, 1)
    __builtin_amdgcn_sched_barrier(0);
    if (first) { C3W_STAMP(1) }

    for (int s = 0; s < nslab; ++s) {
        // the per-slot weight bases are loop invariants: left visible, LICM hoists all 54 (slot, dw, tile) fragment addresses into registers of
        // their own (parked in AGPRs, one v_accvgpr_read per read) instead of one base + the instruction's 16-bit immediate
#pragma unroll
        for (int sl = 0; sl < 3; ++sl) asm volatile("" : "+v"(wv_[sl]));
        C3W_STEP(0, C3W_DMA0, xc_)
        C3W_STEP(1, C3W_DMA1, xc_)
        C3W_ADVANCE_X()
        C3W_STEP(2, C3W_DMA2, xn_)
#pragma unroll
        for (int dw = 0; dw < 3; ++dw) { const unsigned t_ = xc_[dw]; xc_[dw] = xn_[dw]; xn_[dw] = t_; }
    }
    // the trailing fragment reads (the next tile's first groups — re-read below once the epilogue has released its registers) have retired
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (first) { C3W_STAMP(2) }
    // Residual epilogue: ALL 24 of the lane's 16-B residual vectors are requested here, in one batch, before anything is stored (96 registers:
    // the fragment sets are dead).  Interleaved with the stores (load, add, store per vector — the first form of this epilogue) every load's
    // wait is a vmcnt(0) that also waits for the store before it to be acknowledged: 24 serialised memory round trips, ~12 us of a ~60-us
    // tile at 96 channels (rocprof: 3.25 vs 2.63 ms per launch).  Pixels outside the image read the tile's first pixel instead (never used).
    // (bias first: its first use then waits for three loads, not for the residual batch behind them)
    float bias8[3][8];
#pragma unroll
    for (int P = 0; P < 3; ++P) {
        const int n = cur.n0 + ncol0w + 32 * P + 8 * qk;
        if (a.bias && n < a.Cout) {
            const bf16x8 bv = ld_bf16x8(a.bias + n);
#pragma unroll
            for (int e = 0; e < 8; ++e) bias8[P][e] = (float)bv[e];
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) bias8[P][e] = 0.f;
        }
    }
    bf16x8 resv[8][3];
    if (EPI == EPI_RESIDUAL) {
#pragma unroll
        for (int pb = 0; pb < 8; ++pb) {
            const int h = cur.h0 + 4 * wrow + (pb >> 1), w = cur.w0 + 16 * (pb & 1) + l15;
            const long hw = (h < a.H && w < a.W) ? (long)h * a.W + w : (long)cur.h0 * a.W + cur.w0;
            const bf16_t* rp = a.residual + (long)cur.t_out * a.res_fs + hw * a.Cout + cur.n0 + ncol0w + 8 * qk;
#pragma unroll
            for (int P = 0; P < 3; ++P) resv[pb][P] = ld_bf16x8(rp + 32 * P);
        }
    }
    // the accumulators were written by asm MFMAs: the compiler knows no hazard distance to its own reads of them
    asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");
#pragma unroll
    for (int t = 0; t < 6; ++t)
#pragma unroll
        for (int pb = 0; pb < 8; ++pb) asm volatile("" : "+a"(acc[t][pb]));

    // ---- epilogue, straight from the accumulators: lane (l15, qk) holds, for pixel l15 of block pb, channels ncol0 + 32 P + 8 qk + 0..7
    //      (tile 2P: + 0..3, tile 2P + 1: + 4..7).  Rounding points as vae_conv3.hip: y = bf16(acc + bias); bf16(residual + y); norm on the bf16 values.
    {
    const int ncol0 = cur.n0 + ncol0w, h0 = cur.h0, w0 = cur.w0, t_out = cur.t_out;
    const bool fused = a.norm_out != nullptr;
    bf16x8 yv[8][3];
    float ss[8];
#pragma unroll
    for (int pb = 0; pb < 8; ++pb) {
        const int h = h0 + 4 * wrow + (pb >> 1), w = w0 + 16 * (pb & 1) + l15;
        const bool inside = h < a.H && w < a.W;
        const long hw = (long)h * a.W + w;
        float sq = 0.f;
#pragma unroll
        for (int P = 0; P < 3; ++P) {
            const int n = ncol0 + 32 * P + 8 * qk;
            bf16x8 y;
#pragma unroll
            for (int e = 0; e < 8; ++e) y[e] = (bf16_t)((e < 4 ? acc[2 * P][pb][e] : acc[2 * P + 1][pb][e - 4]) + bias8[P][e]);
            if (EPI == EPI_RESIDUAL) {
#pragma unroll
                for (int e = 0; e < 8; ++e) y[e] = (bf16_t)((float)resv[pb][P][e] + (float)y[e]);
            }
            if (inside && (!fused || a.write_raw)) st_bf16x8(a.out + (long)t_out * a.out_fs + hw * a.Cout + n, y);   // (n < Cout: whole wave tiles only)
            yv[pb][P] = y;
#pragma unroll
            for (int e = 0; e < 8; ++e) sq += (float)y[e] * (float)y[e];
        }
        // the pixel's 96 channels of this wave live in the four lanes l15 + 16 g
        sq += __shfl_xor(sq, 16, 64);
        sq += __shfl_xor(sq, 32, 64);
        ss[pb] = sq;
    }
#if FVK_VARIANTS
    if (first && EPI == EPI_BIAS && a.out_f32) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the stores have been accepted
        C3W_STAMP(3)
        if (wave == 0 && lane == 0)
            for (int i = 0; i < 4; ++i) reinterpret_cast<unsigned long long*>(a.out_f32)[(long)blockIdx.x * 4 + i] = stamp_[i];
    }
#endif
    if (fused) {
    // ---- fused RMS-norm (+SiLU) into the consumer conv's input ring: ref WanRMS_norm (wanvae.py:231-232) + SiLU (:418-419) on the bf16-rounded
    //      conv output: inv = sqrt(C) / max(||x||_2, 1e-12); out = bf16(silu(x * inv * gamma))
    if (WNW == 2) {
        // the partner wave (same pixels, the other 96 channels) = wave ^ 1: swap partial sums through 2 KiB of LDS of their own (the slab and
        // weight regions already hold the next tile's first pieces).  norm_out is a kernel argument, so all four waves reach the barriers.
        float* xch = reinterpret_cast<float*>(smem + XCH);
        if (qk == 0) {
#pragma unroll
            for (int pb = 0; pb < 8; ++pb) xch[(wave * 8 + pb) * 16 + l15] = ss[pb];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the writes have left this wave before the barrier
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int pb = 0; pb < 8; ++pb) ss[pb] += xch[((wave ^ 1) * 8 + pb) * 16 + l15];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // (the next tile's epilogue writes the same words)
    }
    int slot = a.norm_slot0 + t_out;
    slot = slot >= a.norm_ring ? slot - a.norm_ring : slot;
    slot = slot >= a.norm_ring ? slot - a.norm_ring : slot;
    float gam[3][8];
#pragma unroll
    for (int P = 0; P < 3; ++P) {
        const float* gp = a.norm_gamma + ncol0w + 32 * P + 8 * qk;
        const f32x4 g0 = *reinterpret_cast<const f32x4*>(gp), g1 = *reinterpret_cast<const f32x4*>(gp + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { gam[P][e] = g0[e]; gam[P][4 + e] = g1[e]; }
    }
    const float sqrtC = sqrtf((float)a.Cout);
#pragma unroll
    for (int pb = 0; pb < 8; ++pb) {
        const int h = h0 + 4 * wrow + (pb >> 1), w = w0 + 16 * (pb & 1) + l15;
        if (h < a.H && w < a.W) {
            const float inv = sqrtC / fmaxf(sqrtf(ss[pb]), 1e-12f);
            const long base = ((long)slot * HW + (long)h * a.W + w) * a.Cout + ncol0w + 8 * qk;
#pragma unroll
            for (int P = 0; P < 3; ++P) {
                bf16x8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float r_ = (float)yv[pb][P][e] * inv * gam[P][e];
                    // x * sigmoid(x) with the hardware reciprocal (1 ulp; the result is rounded to bf16), as vae_conv3.hip / vae_norm12_kernel
                    if (a.norm_silu) r_ = r_ * __builtin_amdgcn_rcpf(1.0f + __expf(-r_));
                    o[e] = (bf16_t)r_;
                }
                st_bf16x8(a.norm_out + base + 32 * P, o);
            }
        }
    }
    }  // fused
    }  // epilogue
    first = false;
    cur_id += (int)gridDim.x;
    if (cur_id >= ntiles) break;   // workgroup-uniform
    cur = decode(cur_id);
    }  // tile loop
    // (a tile past the end was staged as zero-fill pieces: retire them before the workgroup's LDS is released)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#undef C3W_STEP
#undef C3W_GROUP
#undef C3W_READ
#undef C3W_ISSUE_X
#undef C3W_ISSUE_W
#undef C3W_ADVANCE_X
#undef C3W_ADVANCE_W
#endif  // __HIP_DEVICE_COMPILE__
}

template <int WNW, int EPI>
int launch3w(Conv3Args a, hipStream_t s) {
    constexpr int TH = WNW == 1 ? 16 : 8, TN = WNW * 96;
    constexpr int LDS = 2 * ((TH + 2) * 40 / 16) * 1024 + 3 * (3 * TN / 16) * 1024 + (WNW == 2 ? 2048 : 0);
    static FvkLdsConfigured configured;
    if (int rc = fvk_config_lds(configured, (const void*)vae_conv3w_kernel<WNW, EPI>, LDS, "fvk_vae_conv_bf16 (3x3, one wave per SIMD)")) return rc;
    a.tiles_h = (a.H + TH - 1) / TH;
    a.tiles_w = (a.W + 31) / 32;
    a.ntn = (a.Cout + TN - 1) / TN;
    const long nwg = (long)a.T * a.tiles_h * a.tiles_w * a.ntn;
    // persistent workgroups: one per CU (the 144-160 KiB of LDS admit one), each walking tiles b, b + grid, ...; "vae_conv_impl" 4 (measurement
    // build) = one workgroup per tile, i.e. every tile pays its own cold prologue (A/B)
    static int n_cu = 0;
    if (n_cu == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        n_cu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    }
    long grid = nwg < n_cu ? nwg : n_cu;
    int stagger = 0;
#if FVK_VARIANTS
    if (fvk_vae_conv_tunable() == 4) grid = nwg;
    // 5: phase stagger on (a sixteenth of an estimated tile time per phase unit: ~2 830 cycles per K-step + ~20 000 of prologue / epilogue),
    // for launches that give every workgroup at least four tiles
    if (fvk_vae_conv_tunable() == 5 && nwg >= 4 * grid) stagger = (a.KT * (a.Cin / 32) * 3 * 2830 + 20000) / 16;
#endif
    hipLaunchKernelGGL((vae_conv3w_kernel<WNW, EPI>), dim3((unsigned)grid), dim3(256), LDS, s, a, (int)nwg, stagger);
    FVK_LAUNCH_CHECK();
    return FVK_OK;
}

}  // namespace

bool fvk_vae_conv3w_launch(Conv3Args a, int epilogue, hipStream_t s, int* rc) {
    // bf16-output convs whose channel count is a whole number of 96-channel wave tiles, the 2x-upsampling resample convs included (every 3x3
    // conv of the Wan decoder but conv_out: Cout = 3, fp32 planar output)
    if (epilogue != EPI_BIAS && epilogue != EPI_RESIDUAL) return false;
    const bool same = a.Hin == a.H && a.Win == a.W, ups2 = a.H == 2 * a.Hin && a.W == 2 * a.Win;
    if (a.Cout % 96 != 0 || !(same || ups2)) return false;
    if (a.norm_out && a.Cout != 96 && a.Cout != 192) return false;
    const bool wide = a.Cout % 192 == 0;
    if (wide) *rc = epilogue == EPI_BIAS ? launch3w<2, EPI_BIAS>(a, s) : launch3w<2, EPI_RESIDUAL>(a, s);
    else *rc = epilogue == EPI_BIAS ? launch3w<1, EPI_BIAS>(a, s) : launch3w<1, EPI_RESIDUAL>(a, s);
    return true;
}
