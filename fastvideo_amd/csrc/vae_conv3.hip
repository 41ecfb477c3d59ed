// 3x3-spatial causal convolution of the Wan VAE decoder with HALO REUSE (the shipped kernel for every 3x3x3 / 1x3x3 conv; the per-tap
// gather kernel in vae_conv.hip keeps the (3,1,1) time_conv).  Same contract, layouts and epilogues as fvk_vae_conv_bf16 there.
//
// Why: the gather kernel fetches the im2col panel once PER TAP — 32-38 LDS-DMA wave-instructions per 96-128 MFMAs — and an LDS-DMA
// instruction costs its issuing wave ~55 cycles whatever its size (timing ablations, scripts/probes/gemm_st_experiment.hip), so the
// N = 96 convs ran at 750-810 TF.  Here a workgroup owns a SPATIAL tile of one output frame (TH x 32 pixels) and, per (input frame
// dt, 32-channel chunk), stages the tile's (TH+2) x 34-pixel halo slab ONCE; all nine spatial taps are fragment reads of that slab at
// shifted pixel addresses.  DMA wave-instructions per 108 MFMAs drop from ~45 to ~12.
//   * LDS: 2 halo slabs (pixel p = hh*WW + ww at p*64 B, 16-B chunk c at c ^ ((p>>2)&3)) + a 3-deep ring of weight steps (one (dt, chunk,
//     dh): 3 dw taps x TN rows x 64 B).  WNW=1: 16 x 32 pixels x 96 channels (132 KiB), WNW=2: 8 x 32 pixels x 192 channels (153 KiB).
//   * A K-"step" = (dt, chunk, dh): 36 MFMAs per wave (3 dw x 2 k16 x 3 n-blocks x 2 pixel rows), fragment reads issued one (dw, k16)
//     group ahead (source order pinned with sched_barrier(0)), this wave's DMA pieces (weights of step u+2; the next slab during dh = 0, 1)
//     riding in the MFMA gaps, counted s_waitcnt vmcnt + ONE raw s_barrier per step.
//   * UPS (nearest-exact 2x upsample folded in): the slab is staged at INPUT resolution ((TH/2+2) x 18 pixels) and the per-lane pixel
//     address is ((h+dh-1)>>1, (w+dw-1)>>1).
// ref: WanCausalConv3d.forward (fastvideo/models/vaes/wanvae.py:198-207), WanResample (:277-284, :247-248), WanResidualBlock (:462),
//      AutoencoderKLWan.decode's clamp (:1210-1211).
#include "fvk_common.h"
#include "vae_conv3_args.h"
#include <type_traits>

int fvk_vae_conv_tunable();  // vae_conv.hip: the "vae_conv_impl" measurement switch

namespace {

using fvkc3::Conv3Args;
using fvkc3::EPI_BIAS;
using fvkc3::EPI_RESIDUAL;
using fvkc3::EPI_FINAL;
using fvkc3::OOB;

template <int N>
__device__ __forceinline__ void wait_vm() {  // counted wait: at most N of this wave's VMEM operations still in flight
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// NB = 32-channel output blocks per wave: 3 everywhere except conv_out (Cout = 3 <= 32, EPI_FINAL), whose NB = 1 instantiation does a third
// of the padded MFMA work and stages a third of the weight rows.
// STAG (WNW = 1, NB = 3: the 96-channel full-resolution stage): the two waves of a SIMD run HALF A STEP APART.  With all eight waves in
// lockstep every step ends with both waves of a SIMD parked at the barrier and the matrix pipe drained (s_memtime: 3300 cycles per step for
// 2304 of MFMA).  Here waves 0-3 (group A) pass the step barrier at the END of their step and waves 4-7 (group B) in the MIDDLE of theirs,
// so one of the two always has MFMAs in flight.  What makes that safe: a 4-deep weight ring (the slot A refills during step u was read in
// step u-2, which B has left by A's barrier u-1), A issues all weight pieces (waited at the end of its next step), B issues all slab
// pieces in the first halves of dh = 0, 1 (the buffer was last read in the previous slab, which A left before the barrier B passed in
// mid-step 3s-1) and retires them before the barrier of mid-step dh = 2 — the one A passes before it starts the next slab.
template <int WNW, int EPI, bool UPS, int NB = 3, bool STAG = false>
__global__ __launch_bounds__(512, 2) void vae_conv3_kernel(Conv3Args a) {
#if defined(__HIP_DEVICE_COMPILE__)
    static_assert(NB == 3 || (NB == 1 && WNW == 1 && EPI == EPI_FINAL && !UPS), "NB = 1 serves the final conv only");
    static_assert(!STAG || (WNW == 1 && NB == 3), "the staggered schedule needs the 4-deep weight ring to fit: 96-channel tiles only");
    constexpr int TH = WNW == 1 ? 16 : 8, TW = 32;
    constexpr int TN = WNW * NB * 32;
    constexpr int HH = UPS ? TH / 2 + 2 : TH + 2, WW = UPS ? TW / 2 + 2 : TW + 2;  // halo slab in INPUT pixels
    constexpr int XPIECES = (HH * WW + 15) / 16;       // 16-pixel DMA pieces per slab
    constexpr int NIW = STAG ? 4 : 8;                  // waves that issue a given kind of piece (STAG: weights = group A, slabs = group B)
    constexpr int XS = (XPIECES + NIW - 1) / NIW;      // per issuing wave (surplus = zero pieces inside the slab's padded tail)
    constexpr int XS0 = (XS + 1) / 2, XS1 = XS - XS0;  // issued during dh = 0 / dh = 1 of the previous slab
    constexpr int SLAB = XPIECES * 1024;               // surplus piece slots (q >= XPIECES) land in the scratch KiB
    constexpr int WPIECES = 3 * TN / 16;               // (dw, 16 n rows) pieces per weight step
    constexpr int WS = (WPIECES + NIW - 1) / NIW;
    constexpr int WSTEP = WPIECES * 1024;
    constexpr int W_BASE = 2 * SLAB;
    constexpr int RW = STAG ? 4 : 3;                   // weight ring depth
    constexpr int SCRATCH = W_BASE + RW * WSTEP;       // 1 KiB landing zone of the surplus (dummy) pieces
    static_assert(!STAG || XS0 <= 6, "group B issues its slab pieces in the first half of a step: six slots");
    constexpr int EPI_PITCH = 208, EPI_WAVE = 64 * EPI_PITCH;
    static_assert(SCRATCH + 1024 <= 160 * 1024 && 8 * EPI_WAVE <= 160 * 1024, "LDS budget");  // the launch allocates the larger of ring / staging
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef __attribute__((address_space(3))) void lds_void;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const int wrow = WNW == 1 ? wave : (wave >> 1), wn = WNW == 1 ? 0 : (wave & 1);  // wave: tile rows 2*wrow, 2*wrow+1; n offset 96*wn
    const int grp = STAG ? (wave >> 2) : 0;   // STAG: waves 0-3 = group A (weights), 4-7 = group B (slabs); w and w+4 share a SIMD
    const int iw = STAG ? (wave & 3) : wave;  // index among the waves that issue this wave's kind of piece

    // tile id (n fastest so that consecutive workgroups share the halo slab in L2)
    int bid = blockIdx.x;
    const int pid_n = bid % a.ntn; bid /= a.ntn;
    const int tw_i = bid % a.tiles_w; bid /= a.tiles_w;
    const int th_i = bid % a.tiles_h;
    const int t_out = bid / a.tiles_h;
    const int h0 = th_i * TH, w0 = tw_i * TW, n0 = pid_n * TN;

    // ---- x slab staging: piece q = wave + 8*i covers slab pixels 16q .. 16q+15; this lane -> pixel 16q + (lane>>2) -----------------
    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)a.in, 0, (unsigned)((long)a.ring * a.Hin * a.Win * a.Cin * 2), 0x00020000);
    const int Ktot = a.KT * 9 * a.Cin;
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, (unsigned)((long)a.Cout * Ktot * 2), 0x00020000);
    const int chunk16 = ((lane & 3) ^ ((lane >> 4) & 3)) * 16;
    const int CinB = a.Cin * 2;
    const int hb = UPS ? (h0 >> 1) - 1 : h0 - 1, wb = UPS ? (w0 >> 1) - 1 : w0 - 1;  // input coordinates of slab pixel (0, 0)
    unsigned xvo_[XS];
#pragma unroll
    for (int i = 0; i < XS; ++i) {
        const int p = (iw + NIW * i) * 16 + (lane >> 2);
        const int hh = p / WW, ww = p - hh * WW;
        const int h = hb + hh, w = wb + ww;
        const bool ok = hh < HH && (unsigned)h < (unsigned)a.Hin && (unsigned)w < (unsigned)a.Win;
        xvo_[i] = ok ? (unsigned)((h * a.Win + w) * CinB) + chunk16 : OOB;
    }
    // ---- weight staging: piece q = wave + 8*j -> dw = q / (TN/16), rows n0 + (q % (TN/16))*16 + (lane>>2) ------------------------
    unsigned wvo_[WS];
    int wdw_[WS];
#pragma unroll
    for (int j = 0; j < WS; ++j) {
        const int q = iw + NIW * j;
        const int dw = q / (TN / 16), nb16 = q - dw * (TN / 16);
        const int n = n0 + nb16 * 16 + (lane >> 2);
        wvo_[j] = (q < WPIECES && n < a.Cout) ? (unsigned)((long)n * Ktot * 2) + chunk16 : OOB;
        wdw_[j] = q < WPIECES ? dw : 0;
    }
    const int cpt = a.Cin / 32;
    const int nslab = a.KT * cpt, nstep = nslab * 3;
    const unsigned frameB = (unsigned)(a.Hin * a.Win * CinB);

    // Scalar issue state, advanced incrementally (no integer divisions inside the MFMA stream):
    //   next slab  s+1 = (xn_dt, xn_cc): soffset xn_so = slot(t_out + dt) * frameB + cc*64, LDS buffer (s+1)&1
    //   weight step u+2 = (wn_dt, wn_cc, wn_dh): K byte offset wn_ko of tap (dt, dh, dw = 0), chunk cc; + dw*Cin*2 per piece
    auto slot_of = [&](int dt) { int sl = a.ring_start + t_out + dt; return sl >= a.ring ? sl - a.ring : sl; };
    int xn_dt = 0, xn_cc = 0, xn_s = 0;
    int wn_dt = 0, wn_cc = 0, wn_dh = 0, wn_u = 0;
    const unsigned dwB = (unsigned)(a.Cin * 2);
#define C3_ISSUE_X(I)  /* piece I of slab xn_s */                                                                   \
    {                                                                                                               \
        const bool live_ = xn_s < nslab;                                                                            \
        const unsigned so_ = __builtin_amdgcn_readfirstlane(live_ ? (unsigned)slot_of(xn_dt) * frameB + (unsigned)xn_cc * 64u : 0u); \
        const int q_ = iw + NIW * (I);                                                                              \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rsrc, (lds_void*)(smem + (q_ < XPIECES ? (xn_s & 1) * SLAB + q_ * 1024 : SCRATCH)), 16, \
                                                 live_ ? xvo_[I] : OOB, so_, 0, 0);                                 \
    }
#define C3_ADVANCE_X() { ++xn_s; if (++xn_cc == cpt) { xn_cc = 0; ++xn_dt; } }
#define C3_ISSUE_W(J)  /* piece J of weight step wn_u */                                                            \
    {                                                                                                               \
        const bool live_ = wn_u < nstep;                                                                            \
        const unsigned so_ = __builtin_amdgcn_readfirstlane(                                                        \
            live_ ? (unsigned)((((wn_dt * 3 + wn_dh) * 3) * a.Cin + wn_cc * 32) * 2) + (unsigned)wdw_[J] * dwB : 0u); \
        const int q_ = iw + NIW * (J);                                                                              \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (lds_void*)(smem + (q_ < WPIECES ? W_BASE + wn_slot * WSTEP + q_ * 1024 : SCRATCH)), 16, \
                                                 live_ ? wvo_[J] : OOB, so_, 0, 0);                                 \
    }
#define C3_ADVANCE_W() { ++wn_u; wn_slot = wn_slot == RW - 1 ? 0 : wn_slot + 1; if (++wn_dh == 3) { wn_dh = 0; if (++wn_cc == cpt) { wn_cc = 0; ++wn_dt; } } }
    int wn_slot = 0;

    // ---- per-lane slab pixel of output pixel (row 2*wrow + mb, column l31) under tap (dh, dw): p = rowt[mb][dh] + colt[dw] -----------
    int rowt_[2][3], colt_[3];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int dh = 0; dh < 3; ++dh) {
            const int r = 2 * wrow + mb + dh;  // halo row in OUTPUT resolution (0 = h0 - 1)
            rowt_[mb][dh] = (UPS ? ((h0 + r - 1) >> 1) - hb : r) * WW;
        }
#pragma unroll
    for (int dw = 0; dw < 3; ++dw) colt_[dw] = UPS ? ((w0 + l31 + dw - 1) >> 1) - wb : l31 + dw;
    const int wfo = (wn * (NB * 32) + l31) * 64;  // weight fragment row within a (dw) block
    const int wsw = (l31 >> 2) & 3;

    f32x16 acc[NB][2];
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // ---- prologue: slab 0 and weight steps 0, 1 in flight; slab 0 + step 0 landed ----------------------------------------------------
    if (!STAG || grp == 1) {
#pragma unroll
        for (int i = 0; i < XS; ++i) C3_ISSUE_X(i)
    }
    C3_ADVANCE_X()
    if (!STAG || grp == 0) {
#pragma unroll
        for (int j = 0; j < WS; ++j) C3_ISSUE_W(j)
    }
    C3_ADVANCE_W()
    if (!STAG || grp == 0) {
#pragma unroll
        for (int j = 0; j < WS; ++j) C3_ISSUE_W(j)
    }
    C3_ADVANCE_W()
    if (!STAG || grp == 0) wait_vm<WS>();  // STAG: group A holds only weight pieces: step 0 landed, step 1 in flight
    else wait_vm<0>();                     // STAG: group B holds only slab 0
    __builtin_amdgcn_s_barrier();

#ifdef FVK_C3_PROBE  // timing probe build: workgroup 0 sums s_memtime deltas per step phase into out_f32 (as uint64) — EPI_BIAS launches only
    unsigned long long pacc_[4] = {0, 0, 0, 0}, plast_ = 0;
#define C3_STAMP(K)                                                               \
    if (EPI == EPI_BIAS && a.out_f32 && blockIdx.x == 0) {                        \
        const unsigned long long now_ = __builtin_readcyclecounter();            \
        if ((K) != 0 || u > 0) pacc_[K] += now_ - plast_;                         \
        plast_ = now_;                                                            \
    }
#else
#define C3_STAMP(K)
#endif
    int u = 0, rd_slot = 0;
    // The K loop, instantiated once per schedule role so that each role is one straight-line loop (an if / else on the wave group INSIDE the
    // unrolled steps made hipcc spill 120-150 VGPRs): MODE 0 = lockstep, 1 = STAG group A, 2 = STAG group B.
    auto kloop = [&](auto mode_) __attribute__((always_inline)) {
    constexpr int MODE = decltype(mode_)::value;
    for (int s = 0; s < nslab; ++s) {
        const unsigned char* xs = smem + (s & 1) * SLAB;
#pragma unroll
        for (int dh = 0; dh < 3; ++dh, ++u) {
            C3_STAMP(0)
            const unsigned char* wsb = smem + W_BASE + rd_slot * WSTEP;
            bf16x8 wf[2][NB], xf[2][2];
            // fragment group g = (dw, ks): w rows nb at wsb + dw*TN*64 + row*64 + chunk; x pixels mb at xs + p*64 + chunk (both swizzled)
#define C3_READ(G, BUF, R)                                                                                          \
    {                                                                                                               \
        const int dw_ = (G) >> 1, c_ = 2 * ((G) & 1) + hi;                                                          \
        if ((R) < NB) wf[BUF][(R) < NB ? (R) : 0] = *reinterpret_cast<const bf16x8*>(wsb + dw_ * (TN * 64) + wfo + (R) * 2048 + ((c_ ^ wsw) << 4)); \
        else {                                                                                                      \
            const int p_ = rowt_[(R) - NB][dh] + colt_[dw_];                                                        \
            xf[BUF][(R) - NB] = *reinterpret_cast<const bf16x8*>(xs + p_ * 64 + ((c_ ^ ((p_ >> 2) & 3)) << 4));    \
        }                                                                                                           \
    }
            // MODE 0: lockstep (every wave issues its share of both kinds of piece); 1: STAG group A (weights); 2: STAG group B (slabs in the
            // first half of dh = 0, 1; the step barrier in MID-step, after the third fragment group)
#define C3_STREAM(MODE)                                                                                             \
    {                                                                                                               \
        _Pragma("unroll") for (int r = 0; r < NB + 2; ++r) C3_READ(0, 0, r)                                         \
        __builtin_amdgcn_sched_barrier(0);                                                                          \
        __builtin_amdgcn_s_setprio(1);                                                                              \
        _Pragma("unroll") for (int g = 0; g < 6; ++g) {                                                             \
            _Pragma("unroll") for (int i = 0; i < 2 * NB; ++i) {                                                    \
                const int nb = i >> 1, mb = i & 1;                                                                  \
                acc[nb][mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[g & 1][nb], xf[g & 1][mb], acc[nb][mb], 0, 0, 0); \
                if (NB == 3) {                                                                                      \
                    if (g < 5 && i < 5) C3_READ(g + 1, (g + 1) & 1, i)                                              \
                } else if (g < 5) { /* NB = 1: three fragment reads behind two MFMAs */                             \
                    if (i == 0) { C3_READ(g + 1, (g + 1) & 1, 0) C3_READ(g + 1, (g + 1) & 1, 1) }                   \
                    else C3_READ(g + 1, (g + 1) & 1, 2)                                                             \
                }                                                                                                   \
                /* this wave's DMA pieces ride in the gaps: weights of step u+2 (lockstep: its ring slot was read in step u-1), and the   \
                   next slab (its buffer was read in the previous slab) during dh = 0 and 1 so that it has landed by this slab's end */ \
                if (NB == 3 ? (i == 2 || i == 4) : true) {                                                          \
                    const int k = NB == 3 ? 2 * g + (i == 4) : 2 * g + i; /* two issue slots per group, 12 per step */ \
                    if ((MODE) == 0) {                                                                              \
                        if (k < WS) C3_ISSUE_W(k)                                                                   \
                        else if (dh == 0 && k - WS < XS0) C3_ISSUE_X(k - WS)                                        \
                        else if (dh == 1 && k - WS < XS1) C3_ISSUE_X(XS0 + (k - WS))                                \
                    } else if ((MODE) == 1) {                                                                       \
                        if (k < WS) C3_ISSUE_W(k)                                                                   \
                    } else {                                                                                        \
                        if (dh == 0 && k < XS0) C3_ISSUE_X(k)                                                       \
                        else if (dh == 1 && k < XS1) C3_ISSUE_X(XS0 + k)                                            \
                    }                                                                                               \
                }                                                                                                   \
                __builtin_amdgcn_sched_barrier(0);                                                                  \
            }                                                                                                       \
            if ((MODE) == 2 && g == 2) { /* group B's step barrier: the slab pieces of dh = 0, 1 retire before the one of dh = 2 */ \
                __builtin_amdgcn_s_setprio(0);                                                                      \
                if (dh == 2) wait_vm<0>();                                                                          \
                __builtin_amdgcn_sched_barrier(0);                                                                  \
                __builtin_amdgcn_s_barrier();                                                                       \
                __builtin_amdgcn_sched_barrier(0);                                                                  \
                __builtin_amdgcn_s_setprio(1);                                                                      \
            }                                                                                                       \
        }                                                                                                           \
        __builtin_amdgcn_s_setprio(0);                                                                              \
    }
            C3_STREAM(MODE)
#undef C3_STREAM
            C3_ADVANCE_W()
            if (dh == 1) C3_ADVANCE_X()
            rd_slot = rd_slot == RW - 1 ? 0 : rd_slot + 1;
#undef C3_READ
            // everything issued before this step has landed (only this step's own pieces may still be in flight)
            C3_STAMP(1)
            static_assert(STAG || WS + XS0 <= 12, "DMA issue slots per step exhausted");
            if (MODE == 0) {
                if (dh == 0) wait_vm<WS + XS0>();
                else if (dh == 1) wait_vm<WS + XS1>();
                else wait_vm<WS>();
            } else if (MODE == 1) {
                wait_vm<WS>();  // group A: the weights issued in the previous step (for step u+1) have landed; this step's own fly on
            }
            C3_STAMP(2)
            if (MODE != 2) {  // (group B passed this step's barrier in mid-step)
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
            }
            C3_STAMP(3)
        }
    }
    };  // kloop
    if (!STAG) kloop(std::integral_constant<int, 0>{});
    else if (grp == 0) kloop(std::integral_constant<int, 1>{});
    else kloop(std::integral_constant<int, 2>{});
#undef C3_ISSUE_X
#undef C3_ISSUE_W
#undef C3_ADVANCE_X
#undef C3_ADVANCE_W
#ifdef FVK_C3_PROBE
    if (EPI == EPI_BIAS && a.out_f32 && blockIdx.x == 0 && lane == 0)
        for (int i = 0; i < 4; ++i) reinterpret_cast<unsigned long long*>(a.out_f32)[wave * 4 + i] = pacc_[i];
#endif
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // tail (dummy) pieces landed before LDS becomes epilogue staging
    __builtin_amdgcn_s_barrier();

    // ---- epilogue: acc[nb][mb][r] = D[n = nb*32 + (r&3) + 8(r>>2) + 4hi][pixel (row 2*wrow + mb, col l31)] ---------------------------
    const int ncol0 = n0 + wn * 96;
    const int HW = a.H * a.W;
    if (EPI == EPI_FINAL) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int n = ncol0 + nb * 32 + 8 * g + 4 * hi + e;
                    if (n < a.Cout) {
                        const float bv = a.bias ? (float)a.bias[n] : 0.f;
#pragma unroll
                        for (int mb = 0; mb < 2; ++mb) {
                            const int h = h0 + 2 * wrow + mb, w = w0 + l31;
                            if (h < a.H && w < a.W) {
                                const float v = acc[nb][mb][4 * g + e] + bv;
                                a.out_f32[(long)n * a.plane_stride + (long)t_out * HW + h * a.W + w] = fminf(fmaxf(v, -1.0f), 1.0f);
                            }
                        }
                    }
                }
        return;
    }
    if constexpr (NB == 3) {
    unsigned char* st = smem + wave * EPI_WAVE;
#pragma unroll
    for (int nb = 0; nb < 3; ++nb)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int nl = nb * 32 + 8 * g + 4 * hi;
            float b4[4] = {0.f, 0.f, 0.f, 0.f};
            if (a.bias) {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (ncol0 + nl + e < a.Cout) b4[e] = (float)a.bias[ncol0 + nl + e];
            }
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
                bf16x4 y;
#pragma unroll
                for (int e = 0; e < 4; ++e) y[e] = (bf16_t)(acc[nb][mb][4 * g + e] + b4[e]);
                *reinterpret_cast<bf16x4*>(st + (mb * 32 + l31) * EPI_PITCH + nl * 2) = y;
            }
        }
    if (a.norm_out) {
        // ---- fused RMS-norm (+SiLU): ref WanRMS_norm (wanvae.py:231-232) + SiLU (:418-419), same arithmetic as vae_norm12_kernel on the bf16-rounded
        // conv output: inv = sqrt(C) / max(||x||_2, 1e-12); out = bf16(silu(x * inv * gamma)).  The staged tile is wave-private and a wave's LDS
        // operations execute in order, so the passes below need no barrier.
        // pass A: residual add (rounded to bf16 like the unfused epilogue) back into the staging tile (+ the raw store, unless dropped)
        if (EPI == EPI_RESIDUAL || a.write_raw) {
#pragma unroll 4
            for (int it = 0; it < 12; ++it) {
                const int id = it * 64 + lane;
                const int px = id / 12, ch = id - px * 12;
                const int h = h0 + 2 * wrow + (px >> 5), w = w0 + (px & 31), n = ncol0 + ch * 8;
                if (h < a.H && w < a.W) {
                    bf16x8 y = *reinterpret_cast<const bf16x8*>(st + px * EPI_PITCH + ch * 16);
                    const long hw = (long)h * a.W + w;
                    if (EPI == EPI_RESIDUAL) {
                        const bf16x8 res = ld_bf16x8(a.residual + (long)t_out * a.res_fs + hw * a.Cout + n);
#pragma unroll
                        for (int e = 0; e < 8; ++e) y[e] = (bf16_t)((float)res[e] + (float)y[e]);
                        *reinterpret_cast<bf16x8*>(st + px * EPI_PITCH + ch * 16) = y;
                    }
                    if (a.write_raw) st_bf16x8(a.out + (long)t_out * a.out_fs + hw * a.Cout + n, y);
                }
            }
        }
        // pass B: lane p owns staged pixel p: sum of squares over its 96 channels
        float ss = 0.f;
#pragma unroll
        for (int c = 0; c < 12; ++c) {
            const bf16x8 v = *reinterpret_cast<const bf16x8*>(st + lane * EPI_PITCH + c * 16);
#pragma unroll
            for (int e = 0; e < 8; ++e) ss += (float)v[e] * (float)v[e];
        }
        if (WNW == 2) {
            // the partner wave (same pixels, the other 96 channels) = wave ^ 1: swap partial sums through 2 KiB past the staging tiles
            // (uniform branch: norm_out is a kernel argument, so all 8 waves reach the barrier)
            float* xch = reinterpret_cast<float*>(smem + 8 * EPI_WAVE);
            xch[wave * 64 + lane] = ss;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the write has left this wave before the barrier
            __builtin_amdgcn_s_barrier();
            ss += xch[(wave ^ 1) * 64 + lane];
        }
        const float inv = sqrtf((float)a.Cout) / fmaxf(sqrtf(ss), 1e-12f);
        int slot = a.norm_slot0 + t_out;
        slot = slot >= a.norm_ring ? slot - a.norm_ring : slot;
        slot = slot >= a.norm_ring ? slot - a.norm_ring : slot;
        // pass C: normalise, SiLU, store into the consumer's ring
#pragma unroll 4
        for (int it = 0; it < 12; ++it) {
            const int id = it * 64 + lane;
            const int px = id / 12, ch = id - px * 12;
            const float inv_px = __shfl(inv, px, 64);
            const int h = h0 + 2 * wrow + (px >> 5), w = w0 + (px & 31);
            if (h < a.H && w < a.W) {
                const bf16x8 v = *reinterpret_cast<const bf16x8*>(st + px * EPI_PITCH + ch * 16);
                const f32x4 g0 = *reinterpret_cast<const f32x4*>(a.norm_gamma + wn * 96 + ch * 8), g1 = *reinterpret_cast<const f32x4*>(a.norm_gamma + wn * 96 + ch * 8 + 4);
                bf16x8 y;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float r_ = (float)v[e] * inv_px * (e < 4 ? g0[e] : g1[e - 4]);
                    // x * sigmoid(x) with the hardware reciprocal (1 ulp; the result is rounded to bf16): an IEEE division here costs the
                    // epilogue ~10 VALU instructions per element with no MFMA left to hide behind
                    if (a.norm_silu) r_ = r_ * __builtin_amdgcn_rcpf(1.0f + __expf(-r_));
                    y[e] = (bf16_t)r_;
                }
                st_bf16x8(a.norm_out + ((long)slot * HW + (long)h * a.W + w) * a.Cout + wn * 96 + ch * 8, y);
            }
        }
        return;
    }
#pragma unroll 4
    for (int it = 0; it < 12; ++it) {
        const int id = it * 64 + lane;
        const int px = id / 12, ch = id - px * 12;  // staged pixel (mb = px>>5, col = px&31), 16-B chunk
        const int h = h0 + 2 * wrow + (px >> 5), w = w0 + (px & 31), n = ncol0 + ch * 8;
        if (h < a.H && w < a.W && n < a.Cout) {
            bf16x8 y = *reinterpret_cast<const bf16x8*>(st + px * EPI_PITCH + ch * 16);
            const long hw = (long)h * a.W + w;
            if (EPI == EPI_RESIDUAL) {
                const bf16x8 res = ld_bf16x8(a.residual + (long)t_out * a.res_fs + hw * a.Cout + n);
#pragma unroll
                for (int e = 0; e < 8; ++e) y[e] = (bf16_t)((float)res[e] + (float)y[e]);
            }
            st_bf16x8(a.out + (long)t_out * a.out_fs + hw * a.Cout + n, y);
        }
    }
    }  // NB == 3
#endif  // __HIP_DEVICE_COMPILE__
}

template <int WNW, int EPI, bool UPS, int NB = 3, bool STAG = false>
int launch3(Conv3Args a, hipStream_t s) {
    constexpr int TH = WNW == 1 ? 16 : 8;
    constexpr int TN = WNW * NB * 32;
    constexpr int HH = UPS ? TH / 2 + 2 : TH + 2, WW = UPS ? 18 : 34;
    constexpr int RING = 2 * ((HH * WW + 15) / 16) * 1024 + (STAG ? 4 : 3) * (3 * TN / 16) * 1024 + 1024;
    constexpr int LDS = RING > 8 * 64 * 208 ? RING : 8 * 64 * 208;  // epilogue staging reuses the ring
    static FvkLdsConfigured configured;
    if (int rc = fvk_config_lds(configured, (const void*)vae_conv3_kernel<WNW, EPI, UPS, NB, STAG>, LDS, "fvk_vae_conv_bf16 (3x3)")) return rc;
    a.tiles_h = (a.H + TH - 1) / TH;
    a.tiles_w = (a.W + 31) / 32;
    a.ntn = (a.Cout + TN - 1) / TN;
    const long nwg = (long)a.T * a.tiles_h * a.tiles_w * a.ntn;
    hipLaunchKernelGGL((vae_conv3_kernel<WNW, EPI, UPS, NB, STAG>), dim3((unsigned)nwg), dim3(512), LDS, s, a);
    FVK_LAUNCH_CHECK();
    return FVK_OK;
}

template <int WNW, bool STAG>
int launch3_s(const Conv3Args& a, int epi, bool ups, hipStream_t s) {
    if (ups) {
        switch (epi) {
            case EPI_BIAS: return launch3<WNW, EPI_BIAS, true, 3, STAG>(a, s);
            case EPI_RESIDUAL: return launch3<WNW, EPI_RESIDUAL, true, 3, STAG>(a, s);
            default: return launch3<WNW, EPI_FINAL, true, 3, STAG>(a, s);
        }
    }
    switch (epi) {
        case EPI_BIAS: return launch3<WNW, EPI_BIAS, false, 3, STAG>(a, s);
        case EPI_RESIDUAL: return launch3<WNW, EPI_RESIDUAL, false, 3, STAG>(a, s);
        default: return launch3<WNW, EPI_FINAL, false, 3, STAG>(a, s);
    }
}

template <int WNW>
int launch3_e(const Conv3Args& a, int epi, bool ups, hipStream_t s) {
    // 96-channel tiles: the staggered schedule (shipped); "vae_conv_impl" 2 = the lockstep schedule, for A/B
    if constexpr (WNW == 1) {
#if FVK_VARIANTS
        if (fvk_vae_conv_tunable() == 2) return launch3_s<1, false>(a, epi, ups, s);
#endif
        return launch3_s<1, true>(a, epi, ups, s);
    } else {
        return launch3_s<WNW, false>(a, epi, ups, s);
    }
}

}  // namespace

// called by fvk_vae_conv_bf16 (vae_conv.hip) for KH = KW = 3 after its argument checks
int fvk_vae_conv3_launch(const void* in, const void* w, const void* bias, void* out, const void* residual, float* out_f32, int T, int H,
                         int W, int Hin, int Win, int Cin, int Cout, int KT, int ring, int ring_start, long out_fs, long res_fs,
                         long plane_stride, int ups, int epilogue, hipStream_t s, const float* norm_gamma, void* norm_out, int norm_ring,
                         int norm_slot0, int norm_silu, int write_raw) {
    Conv3Args a{};
    a.norm_gamma = norm_gamma; a.norm_out = (bf16_t*)norm_out; a.norm_ring = norm_ring; a.norm_slot0 = norm_slot0; a.norm_silu = norm_silu;
    a.write_raw = write_raw;
    a.in = (const bf16_t*)in; a.w = (const bf16_t*)w; a.bias = (const bf16_t*)bias; a.out = (bf16_t*)out;
    a.residual = (const bf16_t*)residual; a.out_f32 = out_f32;
    a.out_fs = out_fs; a.res_fs = res_fs; a.plane_stride = plane_stride;
    a.T = T; a.H = H; a.W = W; a.Hin = Hin; a.Win = Win; a.Cin = Cin; a.Cout = Cout; a.KT = KT; a.ring = ring; a.ring_start = ring_start;
    // conv_out (96 -> 3 channels, 3 time taps): the rolling three-frame kernel of vae_convout.hip (round 6: every input frame fetched once, the
    // time taps on the MFMA's N axis); "vae_conv_impl" 3 / 4 (measurement build) keep this file's padded-N kernel for A/B
    if (epilogue == EPI_FINAL && !ups && (!FVK_VARIANTS || (fvk_vae_conv_tunable() != 3 && fvk_vae_conv_tunable() != 4))) {
        int rc = FVK_OK;
        if (fvk_vae_convout_launch(a, s, &rc)) return rc;
    }
    if (epilogue == EPI_FINAL && !ups && Cout <= 32) return launch3<1, EPI_FINAL, false, 1>(a, s);  // (other shapes) one 32-channel block per wave
    // bf16-output convs (upsampling or not): the one-wave-per-SIMD kernel on 16x16x32 MFMAs (vae_conv3w.hip, round 4); "vae_conv_impl" 3 (measurement
    // build) keeps this file's 8-wave kernel for A/B and the <= 1-ulp comparison tests
    if (!FVK_VARIANTS || fvk_vae_conv_tunable() != 3) {
        int rc = FVK_OK;
        if (fvk_vae_conv3w_launch(a, epilogue, s, &rc)) return rc;
    }
    const int w96 = (Cout + 95) / 96 * 96, w192 = (Cout + 191) / 192 * 192;
    if (w192 <= w96) return launch3_e<2>(a, epilogue, ups != 0, s);
    return launch3_e<1>(a, epilogue, ups != 0, s);
}
