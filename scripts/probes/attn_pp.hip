// Dense flash-style attention forward, head_dim 128, gfx950 — the 8-wave "ping-pong" kernel used for the DiT's
// self-attention (S x S) and cross-attention (S x 512) at full sequence length.  Same mathematics, operand orientation
// and V^T input layout as attn_fwd.hip (S^T = K·Q^T so a softmax row is lane-local; O^T = V^T·P^T with P^T taken straight
// from the packed S^T accumulators; fp32 online softmax in the exp2 domain; P rounded to bf16 before P·V), different
// schedule:
//   * One 512-thread workgroup per CU owns 256 query rows (8 waves x 32 rows); K/V tiles of 64 keys are shared by all 8
//     waves (half the L2->LDS traffic per FLOP of the 4-wave kernel).
//   * Per KV tile a wave runs two segments separated by s_barrier:   M(j): O^T += V^T(j-1)·P^T(j-1) ; S^T(j) = K(j)·Q^T
//                                                                     V(j): online softmax of S^T(j) -> P^T(j) (bf16)
//     i.e. 32 MFMAs (1024 matrix-pipe cycles) against ~1000 cycles of VALU.  Waves 4-7 run ONE barrier behind waves 0-3,
//     and waves w / w+4 share a SIMD, so at any time each SIMD has one wave in its matrix segment and one in its softmax
//     segment: the matrix pipe and the VALU work in parallel instead of alternating (MI355X_MICROARCH "two waves per SIMD").
//   * K and V^T tiles go global -> LDS by LDS-DMA (buffer_load ... lds) into a 3-deep ring, issued two segments ahead and
//     retired with counted s_waitcnt vmcnt(N); the rows are XOR-swizzled (source address + read address) so that every
//     ds_read_b128 fragment read is bank-conflict free and the ring needs no padding: K tile = V^T tile = 16 KiB =
//     16 DMA wave-instructions, 4 per wave per tile — uniform counts, no staging VGPRs, no ds_write.
#include "fvk_common.h"

namespace {

constexpr int K_TILE = 64 * 256;   // 64 keys x 128 d bf16
constexpr int V_TILE = 128 * 128;  // 128 d x 64 keys bf16
constexpr int RING = 3;
constexpr int LDS_BYTES = RING * (K_TILE + V_TILE);  // 98 304

// value of lane^32 combined with this lane's, via v_permlane32_swap (no LDS crossbar round trip as with ds_bpermute)
__device__ __forceinline__ float xhalf_max(float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float xhalf_sum(float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// SPLIT: where a wave issues its 4 DMA pieces of the next tile set: 0 = K pieces at the top of M, V pieces at the top of V;
//        1 = all at the top of V; 2 = all at the top of M.
template <int SPLIT, bool PIPE, int PRIO, int ABL>
__global__ __launch_bounds__(512, 2) void attn_pp_kernel(fvk_attn_args a) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef __attribute__((address_space(3))) void lds_void;
    constexpr int BMQ = 256;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;
    const int l31 = lane & 31, hi = lane >> 5;
    const int nqb = (a.Sq + BMQ - 1) / BMQ;
    const int qb = blockIdx.x % nqb;
    const int h = (blockIdx.x / nqb) % a.H;
    const int b = blockIdx.x / (nqb * a.H);

    const bf16_t* qp = (const bf16_t*)a.q + (long)b * a.q_bs + (long)h * a.q_hs;
    const bf16_t* kp = (const bf16_t*)a.k + (long)b * a.k_bs + (long)h * a.k_hs;
    const bf16_t* vtp = (const bf16_t*)a.vt + ((long)b * a.H + h) * 128L * a.Skv_pad;
    bf16_t* op = (bf16_t*)a.o + (long)b * a.o_bs + (long)h * a.o_hs;
    const int n = (a.Skv + 63) >> 6;  // KV tiles

    // ---- Q fragments (B operand of S^T = K·Q^T): row q0 + l31, d = 16*ks + 8*hi .. +8 ------------------------------------
    const int q0 = qb * BMQ + wave * 32;
    int qrow = q0 + l31;
    const bool q_ok = qrow < a.Sq;
    qrow = q_ok ? qrow : a.Sq - 1;
    bf16x8 qf[8];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) qf[ks] = ld_bf16x8(qp + (long)qrow * a.q_ss + ks * 16 + hi * 8);

    // ---- LDS-DMA: wave w moves K pieces {w, w+8} (4 key rows x 256 B each) and V^T pieces {w, w+8} (8 d rows x 128 B).
    // LDS image: K row r, 16-B chunk c at r*256 + ((c ^ (r&15)) << 4); V^T row r, chunk c at r*128 + ((c ^ ((r>>1)&7)) << 4).
    // The hardware writes lane-linearly, so each lane fetches the SOURCE chunk that belongs at its linear position.
    const __amdgpu_buffer_rsrc_t k_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)kp, 0, (unsigned)((((long)a.Skv - 1) * a.k_ss + 128) * 2), 0x00020000);  // key rows >= Skv read as zeros
    const __amdgpu_buffer_rsrc_t v_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)vtp, 0, (unsigned)(256L * a.Skv_pad), 0x00020000);
    unsigned kvoff[2], vvoff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int t = wave + 8 * i;
        const int kr = 4 * t + (lane >> 4);
        kvoff[i] = (unsigned)(((long)kr * a.k_ss) * 2) + (unsigned)((((lane & 15) ^ (kr & 15))) << 4);
        const int vr = 8 * t + (lane >> 3);
        vvoff[i] = (unsigned)(vr * a.Skv_pad * 2) + (unsigned)(((lane & 7) ^ ((vr >> 1) & 7)) << 4);
    }
    const unsigned k_tile_bytes = (unsigned)(a.k_ss * 2 * 64);  // bytes between consecutive key tiles
    const int kdst = wave * 1024;                               // + ring slot, + i*8192
    const int vdst = RING * K_TILE + wave * 1024;

#define ISSUE_K(T)                                                                                                   \
    {                                                                                                                \
        const int t_ = (T) < n ? (T) : 0; /* past the end: harmless re-read of tile 0 into a free slot */            \
        const unsigned o_ = (unsigned)t_ * k_tile_bytes; /* in voffset: soffset is not bounds-checked */             \
        unsigned char* d_ = smem + ((T) % RING) * K_TILE + kdst;                                                     \
        _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                                \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(k_rsrc, (lds_void*)(d_ + i * 8192), 16, kvoff[i] + o_, 0, 0, 0); \
    }
#define ISSUE_K1(T, I)                                                                                               \
    {                                                                                                                \
        const int t_ = (T) < n ? (T) : 0;                                                                            \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(k_rsrc, (lds_void*)(smem + ((T) % RING) * K_TILE + kdst + (I) * 8192), 16, \
                                                 kvoff[I] + (unsigned)t_ * k_tile_bytes, 0, 0, 0);                   \
    }
#define ISSUE_V1(T, I)                                                                                               \
    {                                                                                                                \
        const int t_ = (T) < n ? (T) : 0;                                                                            \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(v_rsrc, (lds_void*)(smem + ((T) % RING) * V_TILE + vdst + (I) * 8192), 16, \
                                                 vvoff[I], __builtin_amdgcn_readfirstlane(t_ * 128), 0, 0);          \
    }
#define ISSUE_V(T)                                                                                                   \
    {                                                                                                                \
        const int t_ = (T) < n ? (T) : 0;                                                                            \
        const int so_ = __builtin_amdgcn_readfirstlane(t_ * 128);                                                    \
        unsigned char* d_ = smem + ((T) % RING) * V_TILE + vdst;                                                     \
        _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                                \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(v_rsrc, (lds_void*)(d_ + i * 8192), 16, vvoff[i], so_, 0, 0);  \
    }

    // ---- fragment read offsets ---------------------------------------------------------------------------------------------
    int koff[8], voff[4];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) koff[ks] = l31 * 256 + (((2 * ks + hi) ^ (l31 & 15)) << 4);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) voff[kk] = RING * K_TILE + l31 * 128 + (((2 * kk + hi) ^ ((l31 >> 1) & 7)) << 4);

    f32x16 o[4];
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
    f32x16 s[2];
    bf16x8 pf[4];
    float m_run = -1e30f, l_run = 0.f;
    const float c2 = a.scale * 1.4426950408889634f;

    // S^T(j) = K(j)·Q^T : A = K rows (LDS), B = Q rows (registers)
#define QK(J)                                                                                                        \
    {                                                                                                                \
        const unsigned char* kb_ = smem + ((J) % RING) * K_TILE;                                                     \
        _Pragma("unroll") for (int kb = 0; kb < 2; ++kb) _Pragma("unroll") for (int r = 0; r < 16; ++r) s[kb][r] = 0.f; \
        _Pragma("unroll") for (int ks = 0; ks < 8; ++ks) _Pragma("unroll") for (int kb = 0; kb < 2; ++kb) {         \
            const bf16x8 kf = *reinterpret_cast<const bf16x8*>(kb_ + koff[ks] + kb * 8192);                          \
            s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s[kb], 0, 0, 0);                             \
        }                                                                                                            \
    }
    // O^T += V^T(j)·P^T(j) : A = V^T rows (LDS), B = packed P^T (registers)
#define PV(J)                                                                                                        \
    {                                                                                                                \
        const unsigned char* vb_ = smem + ((J) % RING) * V_TILE;                                                     \
        _Pragma("unroll") for (int kk = 0; kk < 4; ++kk) _Pragma("unroll") for (int d = 0; d < 4; ++d) {            \
            const bf16x8 vf = *reinterpret_cast<const bf16x8*>(vb_ + voff[kk] + d * 4096);                           \
            o[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[kk], o[d], 0, 0, 0);                               \
        }                                                                                                            \
    }
    // The whole matrix segment of the steady state as ONE software-pipelined fragment stream: 16 V^T fragments (P·V of tile J-1) then
    // 16 K fragments (Q·K^T of tile J), each ds_read_b128 issued FD MFMAs ahead of its use.  Left to itself hipcc emits
    // read-pair / wait / MFMA-pair, i.e. every second MFMA eats a full LDS round trip (matrix pipe 52 % busy); sched_group_barrier
    // pins the 1 MFMA : 1 read interleave so that its lgkmcnt waits become counted (FD-1 reads stay in flight).
#define MSEG(J)                                                                                                      \
    {                                                                                                                \
        const unsigned char* vb_ = smem + (((J) - 1) % RING) * V_TILE;                                               \
        const unsigned char* kb_ = smem + ((J) % RING) * K_TILE;                                                     \
        constexpr int FD = 6;                                                                                        \
        bf16x8 fr_[FD];                                                                                              \
        _Pragma("unroll") for (int i = 0; i < FD; ++i)                                                               \
            fr_[i] = *reinterpret_cast<const bf16x8*>(vb_ + voff[i >> 2] + (i & 3) * 4096);                          \
        _Pragma("unroll") for (int kb = 0; kb < 2; ++kb) _Pragma("unroll") for (int r = 0; r < 16; ++r) s[kb][r] = 0.f; \
        __builtin_amdgcn_sched_group_barrier(0x100, FD, 0);                                                          \
        _Pragma("unroll") for (int i = 0; i < 32; ++i) {                                                             \
            if (ABL == 3) { if ((i & 7) == 0) { if (i < 16) o[i & 3][0] += (float)fr_[i % FD][0]; else s[i & 1][0] += (float)fr_[i % FD][0]; } } \
            else if (i < 16) o[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr_[i % FD], pf[i >> 2], o[i & 3], 0, 0, 0); \
            else s[i & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr_[i % FD], qf[(i - 16) >> 1], s[i & 1], 0, 0, 0); \
            const int n_ = (ABL >= 4) ? 64 : i + FD; /* ABL 4 (timing only): no fragment reads inside the stream */ \
            if (n_ < 16) fr_[n_ % FD] = *reinterpret_cast<const bf16x8*>(vb_ + voff[n_ >> 2] + (n_ & 3) * 4096);     \
            else if (n_ < 32) fr_[n_ % FD] = *reinterpret_cast<const bf16x8*>(kb_ + koff[(n_ - 16) >> 1] + ((n_ - 16) & 1) * 8192); \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                       \
            if (n_ < 32) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                          \
            if (SPLIT == 3 && ABL != 6) { /* the next tiles' DMA pieces ride in the MFMA gaps (issue slack of the matrix segment) */ \
                if (i == 9) { ISSUE_K1((J) + 2, 0) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); }              \
                if (i == 13) { ISSUE_K1((J) + 2, 1) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); }             \
                if (i == 17) { ISSUE_V1((J) + 1, 0) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); }             \
                if (i == 21) { ISSUE_V1((J) + 1, 1) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); }             \
            }                                                                                                        \
        }                                                                                                            \
    }
    // online softmax of tile J (row q = lane&31; this lane holds 32 of its 64 scores, lane^32 the other 32)
#define SOFTMAX(J)                                                                                                   \
    {                                                                                                                \
        const int valid_ = a.Skv - ((J) << 6);                                                                       \
        if (valid_ < 64) {                                                                                           \
            _Pragma("unroll") for (int kb = 0; kb < 2; ++kb) _Pragma("unroll") for (int r = 0; r < 16; ++r) {       \
                const int key = kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;                                           \
                if (key >= valid_) s[kb][r] = -INFINITY;                                                             \
            }                                                                                                        \
        }                                                                                                            \
        /* 4 independent max / sum chains: only ONE wave per SIMD is in its softmax segment, so VALU latency is hidden by ILP alone */ \
        float mx4[4];                                                                                                \
        _Pragma("unroll") for (int c = 0; c < 4; ++c) {                                                              \
            mx4[c] = fmaxf(s[c >> 1][(c & 1) * 8], s[c >> 1][(c & 1) * 8 + 1]);                                      \
            _Pragma("unroll") for (int r = 2; r < 8; r += 2)                                                         \
                mx4[c] = fmaxf(fmaxf(mx4[c], s[c >> 1][(c & 1) * 8 + r]), s[c >> 1][(c & 1) * 8 + r + 1]);           \
        }                                                                                                            \
        float mx = fmaxf(fmaxf(mx4[0], mx4[1]), fmaxf(mx4[2], mx4[3]));                                              \
        mx = xhalf_max(mx);                                                                                          \
        const float m_new = fmaxf(m_run, mx);                                                                        \
        if (!__all(m_new == m_run)) {                                                                                \
            const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c2);                                        \
            l_run *= alpha;                                                                                          \
            _Pragma("unroll") for (int d = 0; d < 4; ++d) _Pragma("unroll") for (int r = 0; r < 16; ++r) o[d][r] *= alpha; \
            m_run = m_new;                                                                                           \
        }                                                                                                            \
        const float mc = m_run * c2;                                                                                 \
        float ps4[4] = {0.f, 0.f, 0.f, 0.f};                                                                         \
        _Pragma("unroll") for (int kb = 0; kb < 2; ++kb) _Pragma("unroll") for (int r = 0; r < 16; ++r) {           \
            float p;                                                                                                 \
            if (ABL == 1) p = __builtin_fmaf(s[kb][r], c2, -mc); /* timing ablation: no exp */                       \
            else if (ABL == 2 || ABL >= 5) p = s[kb][r];                     /* timing ablation: no fma, no exp */               \
            else p = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kb][r], c2, -mc));                                      \
            s[kb][r] = p;                                                                                            \
            if (ABL != 2 && ABL < 5) ps4[r & 3] += p;                                                                           \
        }                                                                                                            \
        l_run += (ps4[0] + ps4[1]) + (ps4[2] + ps4[3]);                                                              \
        _Pragma("unroll") for (int kk = 0; kk < 4; ++kk) _Pragma("unroll") for (int jj = 0; jj < 8; ++jj)           \
            pf[kk][jj] = (bf16_t)s[kk >> 1][(kk & 1) * 8 + jj];                                                      \
        /* keep the packing inside this (VALU) segment: without a use here it is sunk below the barrier into M */   \
        _Pragma("unroll") for (int kk = 0; kk < 4; ++kk) asm volatile("" : "+v"(pf[kk]));                           \
    }
#define WAIT_SET() /* this wave's pieces of the set issued one iteration ago have landed */                           \
    {                                                                                                                \
        if (SPLIT == 0) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");                                             \
        else if (SPLIT == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                        \
        else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");                                                        \
    }
#define BAR()                                   \
    {                                           \
        __builtin_amdgcn_sched_barrier(0);      \
        __builtin_amdgcn_s_barrier();           \
        __builtin_amdgcn_sched_barrier(0);      \
    }

    // ---- prologue: K(0) landed; set T(0) = {K(1), V(0)} in flight ------------------------------------------------------------
    ISSUE_K(0)
    ISSUE_K(1)
    ISSUE_V(0)
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    BAR()
    if (grp == 1) BAR()  // stagger: waves 4-7 run one barrier behind

    // ---- j = 0 ---------------------------------------------------------------------------------------------------------------
    if (SPLIT != 1) ISSUE_K(2)
    if (SPLIT >= 2) ISSUE_V(1)
    __builtin_amdgcn_s_setprio(1);
    QK(0)
    __builtin_amdgcn_s_setprio(0);
    WAIT_SET()
    BAR()
    if (SPLIT == 1) ISSUE_K(2)
    if (SPLIT < 2) ISSUE_V(1)
    SOFTMAX(0)
    BAR()
    // ---- steady state ----------------------------------------------------------------------------------------------------------
    // ABL 7 (timing probe): every wave of workgroup 0 accumulates the s_memtime deltas between its segment boundaries over tiles
    // 100..355 in registers (no stores inside the loop) and writes the 7 sums to a.lse (as uint64) after the loop.
    unsigned long long acc_[7] = {0, 0, 0, 0, 0, 0, 0}, last_ = 0;
#define STAMP(K)                                                                                                     \
    if (ABL == 7 && blockIdx.x == 0 && j >= 100 && j < 356) {                                                        \
        const unsigned long long now_ = __builtin_readcyclecounter();                                                \
        if ((K) != 0 || j > 100) acc_[K] += now_ - last_;                                                            \
        last_ = now_;                                                                                                \
    }
    for (int j = 1; j < n; ++j) {
        STAMP(0)
        if ((SPLIT == 0 || SPLIT == 2) && ABL != 6) ISSUE_K(j + 2)
        if (SPLIT == 2 && ABL != 6) ISSUE_V(j + 1)
        if (PRIO == 0) __builtin_amdgcn_s_setprio(1);
        if (PIPE) MSEG(j)
        else {
            PV(j - 1)
            QK(j)
        }
        if (PRIO == 0) __builtin_amdgcn_s_setprio(0);
        STAMP(1)
        WAIT_SET()
        STAMP(2)
        BAR()
        STAMP(3)
        if (SPLIT == 1 && ABL != 6) ISSUE_K(j + 2)
        if (SPLIT < 2 && ABL != 6) ISSUE_V(j + 1)
        STAMP(4)
        if (PRIO == 2) __builtin_amdgcn_s_setprio(1);
        SOFTMAX(j)
        if (PRIO == 2) __builtin_amdgcn_s_setprio(0);
        STAMP(5)
        BAR()
        STAMP(6)
    }
    if (ABL == 7 && blockIdx.x == 0 && lane == 0)
        for (int i = 0; i < 7; ++i) reinterpret_cast<unsigned long long*>(a.lse)[wave * 8 + i] = acc_[i];
    // ---- j = n: last P·V --------------------------------------------------------------------------------------------------------
    __builtin_amdgcn_s_setprio(1);
    PV(n - 1)
    __builtin_amdgcn_s_setprio(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // drain the tail re-reads before this workgroup's LDS can be re-assigned
    if (grp == 0) BAR()                               // matches the extra leading barrier of waves 4-7
#undef ISSUE_K
#undef ISSUE_V
#undef ISSUE_K1
#undef ISSUE_V1
#undef QK
#undef MSEG
#undef STAMP
#undef PV
#undef SOFTMAX
#undef WAIT_SET
#undef BAR

    // ---- epilogue ------------------------------------------------------------------------------------------------------------------
    const float l_tot = xhalf_sum(l_run);
    const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
    if (q_ok) {
        bf16_t* orow = op + (long)qrow * a.o_ss;
#pragma unroll
        for (int d = 0; d < 4; ++d)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                bf16x4 v4;
#pragma unroll
                for (int e = 0; e < 4; ++e) v4[e] = (bf16_t)(o[d][g * 4 + e] * inv);
                *reinterpret_cast<bf16x4*>(orow + d * 32 + g * 8 + hi * 4) = v4;
            }
        if (ABL != 7 && a.lse && hi == 0) a.lse[((long)b * a.H + h) * a.Sq + qrow] = m_run * c2 + log2f(l_tot);
    }
#endif  // __HIP_DEVICE_COMPILE__
}

template <int SPLIT, bool PIPE, int PRIO = 0, int ABL = 0>
int launch(const fvk_attn_args* a, hipStream_t s) {
    static FvkLdsConfigured configured;
    if (int rc = fvk_config_lds(configured, (const void*)attn_pp_kernel<SPLIT, PIPE, PRIO, ABL>, LDS_BYTES, "fvk_attn_dense_bf16 (pp)")) return rc;
    const long nblk = (long)((a->Sq + 255) / 256) * a->H * a->B;
    hipLaunchKernelGGL((attn_pp_kernel<SPLIT, PIPE, PRIO, ABL>), dim3((unsigned)nblk), dim3(512), LDS_BYTES, s, *a);
    FVK_LAUNCH_CHECK();
    return FVK_OK;
}

}  // namespace

// variant: 0/1/2 = DMA issue placement (see SPLIT) with the software-pipelined matrix segment; 3 = SPLIT 0 with the
// compiler-scheduled matrix segment (the previous shipped form, kept for A/B)
int fvk_attn_pp_launch(const fvk_attn_args* a, int variant, hipStream_t s) {
    switch (variant) {
        case 1: return launch<1, true>(a, s);
        case 2: return launch<2, true>(a, s);
        case 3: return launch<0, false>(a, s);
        case 4: return launch<1, true, 1>(a, s);  // no priority hints
        case 5: return launch<1, true, 2>(a, s);  // priority on the softmax segment
        case 6: return launch<1, true, 0, 1>(a, s);  // timing ablations (wrong results): no exp
        case 7: return launch<1, true, 0, 2>(a, s);  //   no fma / exp / sum
        case 8: return launch<1, true, 0, 3>(a, s);  //   no MFMA
        case 9: return launch<1, true, 0, 4>(a, s);  //   MFMAs without their LDS fragment reads
        case 10: return launch<1, true, 0, 5>(a, s);  //   MFMAs only: no fragment reads, no softmax arithmetic
        case 11: return launch<1, true, 0, 6>(a, s);  //   ... and no DMA
        case 12: return launch<3, true>(a, s);        // DMA pieces inside the MFMA stream
        case 13: return launch<1, true, 0, 7>(a, s);  // timing probe (a.lse = stamp buffer)
        default: return launch<0, true>(a, s);
    }
}
