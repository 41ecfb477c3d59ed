// Causal 3-D / 2-D convolution for the Wan VAE decoder as an implicit GEMM on gfx950 MFMA, channels-last bf16.
//   out[t,h,w,co] = bias[co] + sum_{dt,dh,dw,ci} W[co,(dt,dh,dw,ci)] * in[frame(t+dt), h+dh-KH/2, w+dw-KW/2, ci]
// ref: WanCausalConv3d.forward (fastvideo/models/vaes/wanvae.py:198-207; causal = 2*pt frames of history in front, which
//      the caller provides as the two header frames of the input ring instead of torch.cat([cache_x, x]) + F.pad),
//      nn.Conv2d of WanResample (:277-284) with the nearest-exact 2x upsample (:247-248) folded into the gather
//      (UPS: source pixel = (h+dh-1)>>1, (w+dw-1)>>1 of the half-resolution input; nearest => value-exact),
//      residual add of WanResidualBlock (:462) fused as an epilogue, and the decoder's final
//      `.float().clamp(-1, 1)` + NCTHW layout (:1210-1211) fused into the conv_out epilogue (EPI_FINAL).
//
// GEMM view: M = T*H*W output pixels, N = Cout, K = taps*Cin walked in steps of 32 channels of one tap.  The x ("im2col")
// operand is never materialised: each 16-row x 64-B piece of the K-step panel is fetched global -> LDS by LDS-DMA
// (buffer_load ... lds) with a per-lane SOURCE address that applies the tap shift, the input ring slot and the 2x upsample;
// out-of-image taps use an out-of-range offset, which the buffer hardware returns as zeros (= the zero padding).
// Everything else is the ping-pong structure of gemm_pp.hip: 512-thread workgroup, 4-slot LDS ring filled 3 K-steps ahead and
// retired by counted s_waitcnt vmcnt, XOR-swizzled 64-B rows (conflict-free ds_read_b128), two wave groups staggered by one
// barrier so that one wave of every SIMD is in its MFMA segment while its partner loads fragments / issues DMA.
// Wave tile 64(M) x 96(N): the decoder's channel counts are 96 / 192 / 384, so N tiles of 96 (WNW=1: 512 x 96 workgroup tile)
// or 192 (WNW=2: 256 x 192) waste nothing, where a 256-wide tile would idle 25-62 % of the MFMAs.
#include "gemm_common.h"

namespace {

struct ConvArgs {
    const bf16_t* in;        // [ring, Hin, Win, Cin]
    const bf16_t* w;         // [Cout, KT*KH*KW*Cin]
    const bf16_t* bias;      // [Cout] or null
    bf16_t* out;             // pixel (t,h,w) at out + t*out_fs + (h*W+w)*Cout
    const bf16_t* residual;  // pixel (t,h,w) at residual + t*res_fs + (h*W+w)*Cout
    float* out_f32;          // EPI_FINAL: [Cout, planes...]: out_f32 + co*plane_stride + t*H*W + h*W + w
    long out_fs, res_fs, plane_stride;
    int T, H, W, Hin, Win, Cin, Cout, KT, KH, KW;
    int ring, ring_start;
    int M, ntm, ntn;
};

enum { EPI_BIAS = 0, EPI_RESIDUAL = 1, EPI_FINAL = 2 };

constexpr int TK = 32, NSLOT = 4;

template <int WNW, int EPI, bool UPS>
__global__ __launch_bounds__(512, 2) void vae_conv_kernel(ConvArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int TM = WNW == 1 ? 512 : 256;
    constexpr int TN = WNW == 1 ? 96 : 192;
    constexpr int XP = TM / 16 / 8;            // x pieces per wave per K-step (4 or 2)
    constexpr int WP = (TN / 16 + 7) / 8;      // w piece slots per wave per K-step (1 or 2); surplus slots are dummies
    constexpr int NP = XP + WP;                // DMA wave-instructions per wave per K-step (5 or 4)
    constexpr int XREG = TM * 64;              // bytes of the x rows of a slot
    constexpr int SLOT = (TM + TN) * 64;
    constexpr int SCRATCH = NSLOT * SLOT;      // 1 KiB landing zone of the dummy pieces
    constexpr int EPI_PITCH = 208;             // bytes per staged output row (96 bf16 + 16 B pad)
    constexpr int EPI_WAVE = 64 * EPI_PITCH;
    static_assert(8 * EPI_WAVE <= NSLOT * SLOT, "epilogue staging must fit in the ring");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef __attribute__((address_space(3))) void lds_void;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;
    const int l31 = lane & 31, hi = lane >> 5;
    const int wm = WNW == 1 ? wave : (wave >> 1), wn = WNW == 1 ? 0 : (wave & 1);

    // XCD-contiguous tile order (block b runs on XCD b % 8): n fastest so that the WGs of one XCD share x panels in its L2
    int tile_id;
    {
        const int nwg = gridDim.x, bid = blockIdx.x;
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
        tile_id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int pid_m = tile_id / a.ntn, pid_n = tile_id % a.ntn;
    const int m0 = pid_m * TM, n0 = pid_n * TN;
    const int HW = a.H * a.W;

    // ---- staging state ---------------------------------------------------------------------------------------------------
    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)a.in, 0, (unsigned)((long)a.ring * a.Hin * a.Win * a.Cin * 2), 0x00020000);
    const int Ktot = a.KT * a.KH * a.KW * a.Cin;
    const __amdgpu_buffer_rsrc_t w_rsrc =
        __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, (unsigned)((long)a.Cout * Ktot * 2), 0x00020000);
    constexpr unsigned OOB = 0xFFFFFF00u;
    const int chunk16 = ((lane & 3) ^ ((lane >> 4) & 3)) * 16;  // source chunk that belongs at LDS chunk position lane&3
    // x pieces: piece xp = wave*XP + i covers tile rows xp*16 .. +16; this lane fetches row xp*16 + (lane>>2)
    int ph_[XP], pw_[XP];        // pixel coordinates (h, w); h = -(1<<20) marks a row past M
    int sl0_[XP];                // ring slot of logical input frame t (tap dt adds dt, modulo the ring)
    const int CinB = a.Cin * 2;
    const unsigned frameB = (unsigned)(a.Hin * a.Win * CinB);
#pragma unroll
    for (int i = 0; i < XP; ++i) {
        const int m = m0 + (wave * XP + i) * 16 + (lane >> 2);
        const int t = m / HW, hw = m - t * HW;
        const int h = hw / a.W;
        ph_[i] = m < a.M ? h : -(1 << 20);
        pw_[i] = hw - h * a.W;
        int s = a.ring_start + (m < a.M ? t : 0);
        s = s >= a.ring ? s - a.ring : s;
        sl0_[i] = s;
    }
    // w pieces: slot j -> piece wp = wave + 8*j (rows n0 + wp*16 ..); wp >= TN/16 is a dummy (zeros into the scratch KiB)
    unsigned wvo_[WP];
    int wdst_[WP];
#pragma unroll
    for (int j = 0; j < WP; ++j) {
        const int wp = wave + 8 * j;
        const int n = n0 + wp * 16 + (lane >> 2);
        const bool real = wp < TN / 16;
        wvo_[j] = (real && n < a.Cout) ? (unsigned)((long)n * Ktot * 2) + chunk16 : OOB;
        wdst_[j] = real ? XREG + wp * 1024 : -1;
    }
    const int cpt = a.Cin / TK;                 // K-steps per tap
    const int nk = a.KT * a.KH * a.KW * cpt;
    const int padh = a.KH >> 1, padw = a.KW >> 1;
    const int Hlim = a.H, Wlim = a.W;           // validity is tested at OUTPUT resolution (also in UPS mode)

    // issue-side tap iterator (wave-uniform): step j = (tap, cc); xo_[] is recomputed when the tap changes
    int is_dt = 0, is_dh = 0, is_dw = 0, is_cc = 0, is_j = 0;
    unsigned xo_[XP];
#define CONV_TAP_OFFSETS()                                                                                          \
    {                                                                                                               \
        _Pragma("unroll") for (int i = 0; i < XP; ++i) {                                                            \
            int hs = ph_[i] + is_dh - padh, ws = pw_[i] + is_dw - padw;                                             \
            const bool ok = (unsigned)hs < (unsigned)Hlim && (unsigned)ws < (unsigned)Wlim;                         \
            if (UPS) { hs >>= 1; ws >>= 1; }                                                                        \
            int sl = sl0_[i] + is_dt;                                                                               \
            sl = sl >= a.ring ? sl - a.ring : sl;                                                                   \
            xo_[i] = ok ? (unsigned)sl * frameB + (unsigned)((hs * a.Win + ws) * CinB) + chunk16 : OOB;             \
        }                                                                                                           \
    }
#define CONV_ISSUE()                                                                                                \
    {                                                                                                               \
        const bool live_ = is_j < nk;                                                                               \
        unsigned char* d_ = smem + (is_j & (NSLOT - 1)) * SLOT;                                                     \
        const int cco_ = __builtin_amdgcn_readfirstlane(is_cc * 64);                                                \
        const int wso_ = __builtin_amdgcn_readfirstlane(live_ ? is_j * 64 : 0);                                     \
        _Pragma("unroll") for (int i = 0; i < XP; ++i)                                                              \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rsrc, (lds_void*)(d_ + (wave * XP + i) * 1024), 16,          \
                                                     live_ ? xo_[i] : OOB, cco_, 0, 0);                             \
        _Pragma("unroll") for (int j = 0; j < WP; ++j)                                                              \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (lds_void*)(wdst_[j] >= 0 ? d_ + wdst_[j] : smem + SCRATCH), 16, \
                                                     wvo_[j], wso_, 0, 0);                                          \
        ++is_j;                                                                                                     \
        if (++is_cc == cpt) {                                                                                       \
            is_cc = 0;                                                                                              \
            if (++is_dw == a.KW) { is_dw = 0; if (++is_dh == a.KH) { is_dh = 0; ++is_dt; } }                        \
            CONV_TAP_OFFSETS()                                                                                      \
        }                                                                                                           \
    }

    // ---- fragment read offsets (bytes within a slot): row r, k-chunk c at r*64 + ((c ^ ((r>>2)&3)) << 4) -----------------
    const int sw = (l31 >> 2) & 3;
    const int xb0 = (wm * 64 + l31) * 64 + ((hi ^ sw) << 4);
    const int xb1 = (wm * 64 + l31) * 64 + (((2 + hi) ^ sw) << 4);
    const int wb0 = XREG + (wn * 96 + l31) * 64 + ((hi ^ sw) << 4);
    const int wb1 = XREG + (wn * 96 + l31) * 64 + (((2 + hi) ^ sw) << 4);

    f32x16 acc[3][2];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // ---- prologue: three steps in flight, step 0 landed --------------------------------------------------------------------
    CONV_TAP_OFFSETS()
    CONV_ISSUE()
    CONV_ISSUE()
    CONV_ISSUE()
#define WAIT_2STEPS()                                                         \
    {                                                                         \
        if (NP == 5) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");       \
        else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                \
    }
    WAIT_2STEPS()
    __builtin_amdgcn_s_barrier();
    if (grp == 1) __builtin_amdgcn_s_barrier();  // stagger

    for (int u = 0; u < nk; ++u) {
        const unsigned char* slot = smem + (u & (NSLOT - 1)) * SLOT;
        // LOAD segment
        bf16x8 xf[2][2], wf[3][2];
#pragma unroll
        for (int nb = 0; nb < 3; ++nb) {
            wf[nb][0] = *reinterpret_cast<const bf16x8*>(slot + wb0 + nb * 2048);
            wf[nb][1] = *reinterpret_cast<const bf16x8*>(slot + wb1 + nb * 2048);
        }
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
            xf[mb][0] = *reinterpret_cast<const bf16x8*>(slot + xb0 + mb * 2048);
            xf[mb][1] = *reinterpret_cast<const bf16x8*>(slot + xb1 + mb * 2048);
        }
        CONV_ISSUE()  // step u+3 overwrites step u-1's slot: every wave finished reading it before the barrier it just passed
        if (grp == 1) WAIT_2STEPS()                          // step u+1 landed (this wave's pieces)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // fragments in registers before the slot can be refilled
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // MFMA segment
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int nb = 0; nb < 3; ++nb)
#pragma unroll
                for (int mb = 0; mb < 2; ++mb)
                    acc[nb][mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[nb][ks], xf[mb][ks], acc[nb][mb], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        if (grp == 0) WAIT_2STEPS()
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    }
#undef CONV_ISSUE
#undef CONV_TAP_OFFSETS
#undef WAIT_2STEPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // tail (dummy) DMAs must have landed before the ring is reused
    if (grp == 0) __builtin_amdgcn_s_barrier();       // pairs with the trailing barrier of the staggered group
    __builtin_amdgcn_s_barrier();

    // ---- epilogue: acc[nb][mb][r] = D[n = nb*32 + (r&3) + 8(r>>2) + 4hi][m = mb*32 + l31] ---------------------------------
    const int ncol0 = n0 + wn * 96;
    const int mrow0 = m0 + wm * 64;
    if (EPI == EPI_FINAL) {
        // fp32 planar output straight from the accumulators (no bf16 rounding): lanes of one n own 32 consecutive pixels
#pragma unroll
        for (int nb = 0; nb < 3; ++nb)
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int n = ncol0 + nb * 32 + 8 * g + 4 * hi + e;
                    if (n < a.Cout) {
                        const float bv = a.bias ? (float)a.bias[n] : 0.f;
#pragma unroll
                        for (int mb = 0; mb < 2; ++mb) {
                            const int m = mrow0 + mb * 32 + l31;
                            if (m < a.M) {
                                const float v = acc[nb][mb][4 * g + e] + bv;
                                a.out_f32[(long)n * a.plane_stride + m] = fminf(fmaxf(v, -1.0f), 1.0f);
                            }
                        }
                    }
                }
        return;
    }
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
    // the staging region is private to this wave: program order + the compiler's lgkmcnt wait are sufficient
#pragma unroll 4
    for (int it = 0; it < 12; ++it) {
        const int id = it * 64 + lane;
        const int row = id / 12, ch = id - row * 12;
        const int m = mrow0 + row, n = ncol0 + ch * 8;
        if (m < a.M && n < a.Cout) {
            bf16x8 y = *reinterpret_cast<const bf16x8*>(st + row * EPI_PITCH + ch * 16);
            const int t = m / HW, hw = m - t * HW;
            if (EPI == EPI_RESIDUAL) {
                const bf16x8 res = ld_bf16x8(a.residual + (long)t * a.res_fs + (long)hw * a.Cout + n);
#pragma unroll
                for (int e = 0; e < 8; ++e) y[e] = (bf16_t)((float)res[e] + (float)y[e]);
            }
            st_bf16x8(a.out + (long)t * a.out_fs + (long)hw * a.Cout + n, y);
        }
    }
#endif  // __HIP_DEVICE_COMPILE__
}

template <int WNW, int EPI, bool UPS>
int launch(ConvArgs a, hipStream_t s) {
    constexpr int TM = WNW == 1 ? 512 : 256;
    constexpr int TN = WNW == 1 ? 96 : 192;
    constexpr int LDS = NSLOT * (TM + TN) * 64 + 1024;
    static FvkLdsConfigured configured;
    if (int rc = fvk_config_lds(configured, (const void*)vae_conv_kernel<WNW, EPI, UPS>, LDS, "fvk_vae_conv_bf16")) return rc;
    a.ntm = (a.M + TM - 1) / TM;
    a.ntn = (a.Cout + TN - 1) / TN;
    hipLaunchKernelGGL((vae_conv_kernel<WNW, EPI, UPS>), dim3(a.ntm * a.ntn), dim3(512), LDS, s, a);
    FVK_LAUNCH_CHECK();
    return FVK_OK;
}

template <int WNW, int EPI>
int launch_u(const ConvArgs& a, bool ups, hipStream_t s) {
    return ups ? launch<WNW, EPI, true>(a, s) : launch<WNW, EPI, false>(a, s);
}
template <int WNW>
int launch_e(const ConvArgs& a, int epi, bool ups, hipStream_t s) {
    switch (epi) {
        case EPI_BIAS: return launch_u<WNW, EPI_BIAS>(a, ups, s);
        case EPI_RESIDUAL: return launch_u<WNW, EPI_RESIDUAL>(a, ups, s);
        default: return launch_u<WNW, EPI_FINAL>(a, ups, s);
    }
}

// ---- RMS norm over channels (+ SiLU), channels-last rows --------------------------------------------------------------------
// ref: WanRMS_norm.forward (wanvae.py:231-232): F.normalize(x, dim=C) * sqrt(C) * gamma, then the block's SiLU (:418-419).
// One pixel = C bf16 = C/8 lanes of 16 B inside a group of G = 16 / 32 / 64 lanes (C <= 128 / 256 / 512); fp32 math.
// Output pixel (t, hw) goes to ring slot (slot0 + t) % ring of the consumer conv's input ring.
template <int G>
__global__ __launch_bounds__(256) void vae_norm_kernel(const bf16_t* __restrict__ x, const float* __restrict__ gamma,
                                                       bf16_t* __restrict__ out, long n_pix, int C, int HW, int ring, int slot0,
                                                       int silu) {
    const int lane_g = threadIdx.x & (G - 1);
    const long pix = ((long)blockIdx.x * 256 + threadIdx.x) / G;
    const bool act = pix < n_pix && lane_g * 8 < C;
    float v[8];
    float ss = 0.f;
    if (act) {
        const bf16x8 xv = ld_bf16x8(x + pix * C + lane_g * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            v[e] = (float)xv[e];
            ss += v[e] * v[e];
        }
    }
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
    if (act) {
        const float inv = sqrtf((float)C) / fmaxf(sqrtf(ss), 1e-12f);
        const f32x4 g0 = *reinterpret_cast<const f32x4*>(gamma + lane_g * 8), g1 = *reinterpret_cast<const f32x4*>(gamma + lane_g * 8 + 4);
        bf16x8 y;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float r = v[e] * inv * (e < 4 ? g0[e] : g1[e - 4]);
            if (silu) r = r / (1.0f + __expf(-r));
            y[e] = (bf16_t)r;
        }
        const long t = pix / HW, hw = pix - t * HW;
        int s = slot0 + (int)t;
        s = s >= ring ? s - ring : s;
        s = s >= ring ? s - ring : s;
        st_bf16x8(out + ((long)s * HW + hw) * C + lane_g * 8, y);
    }
}

// Same op for C = 96 / 192 / 384 (the Wan2.1 decoder widths): a pixel is owned by 12 lanes holding C/12 = 8 / 16 / 32 channels each, five
// pixels per wave (60 of 64 lanes busy instead of 48 with power-of-two groups); the 12-lane sum is xor-1, xor-2 (inside aligned quads)
// plus the two other quads of the dozen.
template <int CPL>  // channels per lane: 8, 16 or 32
__global__ __launch_bounds__(256) void vae_norm12_kernel(const bf16_t* __restrict__ x, const float* __restrict__ gamma,
                                                         bf16_t* __restrict__ out, long n_pix, int HW, int ring, int slot0, int silu) {
    constexpr int C = CPL * 12;
    const int lane = threadIdx.x & 63;
    const int sub = lane / 12, ll = lane - sub * 12;            // pixel within the wave (0..4; 5 = idle lanes 60..63), lane within the dozen
    const long wave_g = ((long)blockIdx.x * 256 + threadIdx.x) >> 6;
    const long pix = wave_g * 5 + sub;
    const bool act = sub < 5 && pix < n_pix;
    float v[CPL];
    float ss = 0.f;
    if (act) {
#pragma unroll
        for (int j = 0; j < CPL / 8; ++j) {
            const bf16x8 xv = ld_bf16x8(x + pix * C + ll * CPL + j * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                v[j * 8 + e] = (float)xv[e];
                ss += v[j * 8 + e] * v[j * 8 + e];
            }
        }
    }
    ss += __shfl_xor(ss, 1, 64);
    ss += __shfl_xor(ss, 2, 64);
    {   // the dozen = three aligned quads: add the other two quads' sums
        const int base = sub * 12, q = ll >> 2, r = ll & 3;
        const float s1 = __shfl(ss, base + ((q + 1) % 3) * 4 + r, 64), s2 = __shfl(ss, base + ((q + 2) % 3) * 4 + r, 64);
        ss += s1 + s2;
    }
    if (act) {
        const float inv = sqrtf((float)C) / fmaxf(sqrtf(ss), 1e-12f);
        const long t = pix / HW, hw = pix - t * HW;
        int s = slot0 + (int)t;
        s = s >= ring ? s - ring : s;
        s = s >= ring ? s - ring : s;
        bf16_t* o = out + ((long)s * HW + hw) * C + ll * CPL;
#pragma unroll
        for (int j = 0; j < CPL / 8; ++j) {
            const f32x4 g0 = *reinterpret_cast<const f32x4*>(gamma + ll * CPL + j * 8), g1 = *reinterpret_cast<const f32x4*>(gamma + ll * CPL + j * 8 + 4);
            bf16x8 y;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float r_ = v[j * 8 + e] * inv * (e < 4 ? g0[e] : g1[e - 4]);
                if (silu) r_ = r_ / (1.0f + __expf(-r_));
                y[e] = (bf16_t)r_;
            }
            st_bf16x8(o + j * 8, y);
        }
    }
}

}  // namespace

int fvk_vae_conv3_launch(const void* in, const void* w, const void* bias, void* out, const void* residual, float* out_f32, int T, int H,
                         int W, int Hin, int Win, int Cin, int Cout, int KT, int ring, int ring_start, long out_fs, long res_fs,
                         long plane_stride, int ups, int epilogue, hipStream_t s, const float* norm_gamma = nullptr, void* norm_out = nullptr,
                         int norm_ring = 0, int norm_slot0 = 0, int norm_silu = 0, int write_raw = 1);  // vae_conv3.hip

extern "C" int fvk_vae_conv_bf16(const void* in, const void* w, const void* bias, void* out, const void* residual,
                                 float* out_f32, int T, int H, int W, int Cin, int Cout, int KT, int KH, int KW, int ring,
                                 int ring_start, long out_frame_stride, long res_frame_stride, long plane_stride, int upsample2x,
                                 int epilogue, void* stream) {
    FVK_CHECK(in && w, FVK_ERR_ARG, "fvk_vae_conv_bf16: null pointer");
    FVK_CHECK(T > 0 && H > 0 && W > 0 && Cout > 0, FVK_ERR_ARG, "fvk_vae_conv_bf16: empty shape T=%d H=%d W=%d Cout=%d", T, H, W, Cout);
    FVK_CHECK(Cin > 0 && Cin % 32 == 0, FVK_ERR_ARG, "fvk_vae_conv_bf16: Cin=%d must be a multiple of 32 (pad the channels)", Cin);
    FVK_CHECK((KT == 1 || KT == 3) && (KH == 1 || KH == 3) && KH == KW, FVK_ERR_ARG, "fvk_vae_conv_bf16: kernel %dx%dx%d unsupported",
              KT, KH, KW);
    FVK_CHECK(!upsample2x || (KT == 1 && H % 2 == 0 && W % 2 == 0), FVK_ERR_ARG, "fvk_vae_conv_bf16: upsample2x needs KT=1 and even H, W");
    FVK_CHECK(ring >= T + KT - 1 && ring_start >= 0 && ring_start < ring, FVK_ERR_ARG,
              "fvk_vae_conv_bf16: ring=%d too small for T=%d KT=%d (or bad ring_start=%d)", ring, T, KT, ring_start);
    FVK_CHECK(epilogue >= EPI_BIAS && epilogue <= EPI_FINAL, FVK_ERR_ARG, "fvk_vae_conv_bf16: bad epilogue %d", epilogue);
    FVK_CHECK(epilogue == EPI_FINAL ? (out_f32 != nullptr) : (out != nullptr && Cout % 8 == 0), FVK_ERR_ARG,
              "fvk_vae_conv_bf16: output pointer / Cout=%d (bf16 outputs need Cout %% 8 == 0)", Cout);
    FVK_CHECK(epilogue != EPI_RESIDUAL || residual, FVK_ERR_ARG, "fvk_vae_conv_bf16: residual epilogue without residual");
    const int Hin = upsample2x ? H / 2 : H, Win = upsample2x ? W / 2 : W;
    FVK_CHECK((long)ring * Hin * Win * Cin * 2 < 0xFFFFFF00L && (long)Cout * KT * KH * KW * Cin * 2 < 0xFFFFFF00L && (long)T * H * W < 0x7FFFFFFFL,
              FVK_ERR_ARG, "fvk_vae_conv_bf16: tensor exceeds the 32-bit buffer-offset range");
    // 3x3 spatial taps: halo-reuse kernel (vae_conv3.hip); "vae_conv_impl" = 1 forces this file's per-tap gather kernel (A/B)
    if (KH == 3 && fvk::tunable(fvk::TUNE_VAE_CONV_IMPL) != 1)
        return fvk_vae_conv3_launch(in, w, bias, out, residual, out_f32, T, H, W, Hin, Win, Cin, Cout, KT, ring, ring_start, out_frame_stride,
                                    res_frame_stride, plane_stride, upsample2x, epilogue, (hipStream_t)stream);
    ConvArgs a{};
    a.in = (const bf16_t*)in; a.w = (const bf16_t*)w; a.bias = (const bf16_t*)bias; a.out = (bf16_t*)out;
    a.residual = (const bf16_t*)residual; a.out_f32 = out_f32;
    a.out_fs = out_frame_stride; a.res_fs = res_frame_stride; a.plane_stride = plane_stride;
    a.T = T; a.H = H; a.W = W; a.Hin = Hin; a.Win = Win; a.Cin = Cin; a.Cout = Cout; a.KT = KT; a.KH = KH; a.KW = KW;
    a.ring = ring; a.ring_start = ring_start; a.M = T * H * W;
    // N tile: 96-wide unless the 192-wide tile wastes less (Cout 192 / 384 / 768 ...)
    const int w96 = (Cout + 95) / 96 * 96, w192 = (Cout + 191) / 192 * 192;
    if (w192 <= w96) return launch_e<2>(a, epilogue, upsample2x != 0, (hipStream_t)stream);
    return launch_e<1>(a, epilogue, upsample2x != 0, (hipStream_t)stream);
}

int fvk_vae_conv_tunable() { return fvk::tunable(fvk::TUNE_VAE_CONV_IMPL); }

// 3x3-tap conv with the consumer's RMS-norm (+SiLU) fused into the epilogue (Cout == 96 or 192): see include/fvk_amd.h
extern "C" int fvk_vae_conv_norm_bf16(const void* in, const void* w, const void* bias, void* out, const void* residual, int T, int H, int W,
                                      int Cin, int Cout, int KT, int ring, int ring_start, long out_frame_stride, long res_frame_stride,
                                      int upsample2x, const float* norm_gamma, void* norm_out, int norm_ring, int norm_slot0, int norm_silu,
                                      void* stream) {
    FVK_CHECK(in && w && norm_gamma && norm_out, FVK_ERR_ARG, "fvk_vae_conv_norm_bf16: null pointer");
    FVK_CHECK(T > 0 && H > 0 && W > 0 && (Cout == 96 || Cout == 192), FVK_ERR_ARG,
              "fvk_vae_conv_norm_bf16: Cout=%d (the fused norm serves the 96- and 192-channel stages: one workgroup must hold all channels of a pixel)", Cout);
    FVK_CHECK(Cin > 0 && Cin % 32 == 0 && (KT == 1 || KT == 3), FVK_ERR_ARG, "fvk_vae_conv_norm_bf16: Cin=%d KT=%d", Cin, KT);
    FVK_CHECK(!upsample2x || (KT == 1 && H % 2 == 0 && W % 2 == 0), FVK_ERR_ARG, "fvk_vae_conv_norm_bf16: upsample2x needs KT=1 and even H, W");
    FVK_CHECK(ring >= T + KT - 1 && ring_start >= 0 && ring_start < ring, FVK_ERR_ARG, "fvk_vae_conv_norm_bf16: ring=%d too small (T=%d KT=%d)", ring, T, KT);
    FVK_CHECK(norm_ring >= T && norm_slot0 >= 0 && norm_slot0 < norm_ring, FVK_ERR_ARG, "fvk_vae_conv_norm_bf16: bad norm ring %d / slot %d", norm_ring, norm_slot0);
    const int Hin = upsample2x ? H / 2 : H, Win = upsample2x ? W / 2 : W;
    FVK_CHECK((long)ring * Hin * Win * Cin * 2 < 0xFFFFFF00L && (long)Cout * KT * 9 * Cin * 2 < 0xFFFFFF00L && (long)T * H * W < 0x7FFFFFFFL,
              FVK_ERR_ARG, "fvk_vae_conv_norm_bf16: tensor exceeds the 32-bit buffer-offset range");
    return fvk_vae_conv3_launch(in, w, bias, out, residual, nullptr, T, H, W, Hin, Win, Cin, Cout, KT, ring, ring_start, out_frame_stride,
                                res_frame_stride, 0, upsample2x, residual ? EPI_RESIDUAL : EPI_BIAS, (hipStream_t)stream, norm_gamma, norm_out,
                                norm_ring, norm_slot0, norm_silu, out != nullptr);
}

extern "C" int fvk_vae_rmsnorm_silu_bf16(const void* x, const float* gamma, void* out, long n_pix, int C, int HW, int ring, int slot0,
                                         int silu, void* stream) {
    FVK_CHECK(x && gamma && out, FVK_ERR_ARG, "fvk_vae_rmsnorm_silu_bf16: null pointer");
    FVK_CHECK(C > 0 && C % 8 == 0 && C <= 512, FVK_ERR_ARG, "fvk_vae_rmsnorm_silu_bf16: C=%d must be a multiple of 8, <= 512", C);
    FVK_CHECK(n_pix > 0 && HW > 0 && ring > 0 && slot0 >= 0 && slot0 < ring, FVK_ERR_ARG, "fvk_vae_rmsnorm_silu_bf16: bad shape");
    hipStream_t s = (hipStream_t)stream;
    if (C == 96 || C == 192 || C == 384) {
        const long waves = (n_pix + 4) / 5;
        const unsigned blocks = (unsigned)((waves + 3) / 4);
        if (C == 96) hipLaunchKernelGGL((vae_norm12_kernel<8>), dim3(blocks), dim3(256), 0, s, (const bf16_t*)x, gamma, (bf16_t*)out, n_pix, HW, ring, slot0, silu);
        else if (C == 192) hipLaunchKernelGGL((vae_norm12_kernel<16>), dim3(blocks), dim3(256), 0, s, (const bf16_t*)x, gamma, (bf16_t*)out, n_pix, HW, ring, slot0, silu);
        else hipLaunchKernelGGL((vae_norm12_kernel<32>), dim3(blocks), dim3(256), 0, s, (const bf16_t*)x, gamma, (bf16_t*)out, n_pix, HW, ring, slot0, silu);
        FVK_LAUNCH_CHECK();
        return FVK_OK;
    }
    const int G = C <= 128 ? 16 : (C <= 256 ? 32 : 64);
    const long threads = n_pix * G;
    const unsigned blocks = (unsigned)((threads + 255) / 256);
    if (G == 16) hipLaunchKernelGGL((vae_norm_kernel<16>), dim3(blocks), dim3(256), 0, s, (const bf16_t*)x, gamma, (bf16_t*)out, n_pix, C, HW, ring, slot0, silu);
    else if (G == 32) hipLaunchKernelGGL((vae_norm_kernel<32>), dim3(blocks), dim3(256), 0, s, (const bf16_t*)x, gamma, (bf16_t*)out, n_pix, C, HW, ring, slot0, silu);
    else hipLaunchKernelGGL((vae_norm_kernel<64>), dim3(blocks), dim3(256), 0, s, (const bf16_t*)x, gamma, (bf16_t*)out, n_pix, C, HW, ring, slot0, silu);
    FVK_LAUNCH_CHECK();
    return FVK_OK;
}
