// Dense flash-style attention forward, head_dim 128, gfx950 — 8-wave ping-pong kernel with 128-key tiles (the shipped kernel for the
// DiT's full-length self-attention).  Same mathematics, operand orientation and V^T input layout as attn_fwd.hip / attn_pp.hip
// (S^T = K·Q^T so a softmax row is lane-local; O^T = V^T·P^T with P^T taken straight from the packed S^T accumulators; fp32 online
// softmax in the exp2 domain, rescale skipped exactly when no row maximum moved; P rounded to bf16 before P·V).
//
// Why 128 keys per step: s_memtime probes of attn_pp.hip (64-key tiles) showed the softmax segment is NOT the bottleneck (it idles
// ~900 of every 3650 cycles at a barrier); each matrix segment pays ~480 cycles of hand-off (barrier skew + release latency + LDS
// latency of its first fragments) on top of 32 MFMAs.  Doubling the tile doubles the work per hand-off: 64 MFMAs per matrix segment
// (P·V of tile j-1: 32, Q·K^T of tile j: 32, four independent accumulator chains each), 2 barriers per 128 keys.
//   * One 512-thread workgroup per CU owns 256 query rows (8 waves x 32 rows); waves w and w+4 share a SIMD and run one barrier
//     apart, so each SIMD always has one wave in its matrix segment and one in its softmax segment.
//   * K and V^T tiles (32 KiB each) go global -> LDS by LDS-DMA into a 2-deep ring (128 KiB), XOR-swizzled on the source side so that
//     every ds_read_b128 fragment read is conflict-free.  A wave issues its 8 pieces of the next set {K(j+1), V(j)} as soon as the
//     ring slots are free — the leading group inside the first half of its MFMA stream (one piece per 4 MFMAs; pieces issued late would still be in flight at the segment's barrier, which measurably waits for them), the trailing group at the top of its
//     softmax segment — and retires them with one s_waitcnt vmcnt(0) two segments later; raw s_barrier only.
//   * The matrix segment is one software-pipelined fragment stream: each ds_read_b128 is issued 6 MFMAs ahead of its use
//     (sched_group_barrier pins the interleave; left alone hipcc emits read-pair / wait / MFMA-pair).
#include "fvk_common.h"
#include "attn_lists.h"

namespace {

constexpr int KT = 128;             // keys per tile
constexpr int K_TILE = KT * 256;    // 128 keys x 128 d bf16
constexpr int V_TILE = 128 * KT * 2;  // 128 d x 128 keys bf16
constexpr int RING = 2;
constexpr int V_BASE = RING * K_TILE;
constexpr int LDS_BYTES = RING * (K_TILE + V_TILE);  // 131 072

}  // namespace
// LIST mode (fvk_attn_tile_lists_bf16, attn_fwd.hip): every `q_stride` consecutive query rows share ONE list of 64-key KV blocks (the
// sliding-tile window of their tile); a workgroup owns 256 of those rows and walks the list two blocks (= one 128-key tile) per step.
namespace {

__device__ __forceinline__ float xhalf_max(float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float xhalf_sum(float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// VSTREAM: the trailing group issues its 8 V^T pieces per tile inside the second half of its matrix segment and only the 8 K pieces at the
// top of its softmax segment.  s_memtime stamps (profiles/r02_attn_pp2_probe.md): the tile step IS the trailing group's serial chain
// (matrix 2300 + DMA issue 998 + softmax 1902 + ~990 of barrier / loop cycles; the leading group idles 1187 cycles at its second barrier),
// and a DMA piece costs its issuing wave ~62 cycles at the top of the softmax segment but only the MFMA-gap overflow inside the stream.
// ONEBAR: ONE workgroup barrier per 128-key tile step instead of two.  The barrier between a group's matrix segment and its partner's
// softmax segment (leading after-M / trailing after-S) guards no data: the ring slots a matrix segment reads were waited for one barrier
// earlier, and they are refilled only after the OTHER barrier (leading after-S / trailing after-M), which every wave still passes.  It only
// forced the two matrix segments of a SIMD to alternate strictly — each hand-off idling the matrix pipe (s_memtime: the tile step was
// exactly 2 x (matrix segment + hand-off)).  Without it the trailing wave starts its matrix segment when its own softmax is done.
// (VSTREAM needs that barrier — the leading group's P·V must have left the V^T slot — so ONEBAR implies all 16 pieces after the barrier.)
// LDMA (implies ONEBAR): the LEADING group issues every DMA piece and waits for them.  With one barrier per step the tile step is the
// trailing wave's own serial chain (matrix 2300 + vmcnt/barrier 420 + DMA issue 1020 + softmax 1940 + loop 560 = 6360 cycles by s_memtime)
// while the leading wave idles ~1480 cycles at the barrier: the DMA work moves to the wave that has the slack.  Set j+1's ring slots are
// free as soon as the barrier closing step j is passed (both groups' M(j) are behind it), which is exactly where the leading group's next
// matrix segment starts; the pieces ride in that segment's MFMA gaps (LSTREAM) or precede it, and are waited for before the leading
// group's next barrier, ~4 000 cycles later, by which time they have long landed (the trailing group used to wait 200-400 cycles there).
// LIST: the KV tiles come from a per-workgroup list of 64-key blocks (two per 128-key tile step, each with its own valid-key count) instead of
// 0 .. Skv/128: K pieces add their block's row offset, V^T pieces select the block per lane (a piece row spans both halves), the softmax
// masks each half by its block size.  The list lives in LDS (packed per tile: id | size << 24 for both halves) and is read one step
// ahead.  Workgroup ids are dealt so that each XCD (private L2) owns a contiguous run of lists: neighbouring tiles' windows overlap.
// ABL (measurement build only; results are WRONG, timing is what is measured — "where do the cycles / the power go"):
//   bit 0: half the LDS fragment reads (every fragment feeds two MFMAs: what a 64-row wave tile would read);  bit 1: no v_exp (the
//   exponent argument is used as P);  bit 2: no LDS-DMA in the steady state (tiles are re-read from the ring as they are)
template <bool PROBE, bool PRIO, bool VSTREAM = true, bool ONEBAR = false, bool LDMA = false, bool LSTREAM = false, bool LIST = false, int ABL = 0>
__global__ __launch_bounds__(512, 2) void attn_pp2_kernel(fvk_attn_args a, fvk_pp2_lists la) {
    static_assert(!(VSTREAM && ONEBAR), "in-stream V^T pieces rely on the second barrier");
    static_assert(!LDMA || ONEBAR, "leading-group DMA is a one-barrier schedule");
    static_assert(!LIST || (LDMA && LSTREAM && !PROBE), "block lists ride on the shipped schedule only");
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef __attribute__((address_space(3))) void lds_void;
    constexpr int BMQ = 256;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;
    const int l31 = lane & 31, hi = lane >> 5;
    int n;            // KV tiles of this workgroup
    int q_first, h, b;
    int lst_num = 0;  // LIST: 64-key blocks in this workgroup's list
    int32_t* const lst = reinterpret_cast<int32_t*>(smem + LDS_BYTES);  // LIST: [tile][2] packed entries
    if (LIST) {
        // XCD-aware deal: hardware workgroup id x lands on XCD x % 8; XCD c gets the contiguous logical ids [c*q + min(c,r), ...)
        const int nblk = gridDim.x, xq = nblk >> 3, xr = nblk & 7, xcd = blockIdx.x & 7;
        const int bid = la.plain_ids ? (int)blockIdx.x : xcd * xq + (xcd < xr ? xcd : xr) + (blockIdx.x >> 3);
        const int per_head = la.n_lists * la.q_sub;
        const int u = bid % per_head, sub = u % la.q_sub, li = u / la.q_sub;
        h = (bid / per_head) % a.H;
        b = bid / (per_head * a.H);
        q_first = li * la.q_stride + sub * BMQ;
        const long meta = ((long)b * a.H + h) * la.n_lists + li;
        lst_num = la.q2k_num[meta];
        if (la.q_rows_valid && sub * BMQ >= la.q_rows_valid[li]) lst_num = 0;  // only padding rows: nothing to do
        n = (lst_num + 1) >> 1;
        const int32_t* src = la.q2k_idx + meta * la.max_kv;
        for (int t = tid; t < n; t += 512) {
            const int id0 = src[2 * t];
            const bool two = 2 * t + 1 < lst_num;
            const int id1 = two ? src[2 * t + 1] : id0;  // an absent second half re-reads the first block, fully masked
            lst[2 * t] = id0 | (la.kv_block_sizes[id0] << 24);
            lst[2 * t + 1] = id1 | ((two ? la.kv_block_sizes[id1] : 0) << 24);
        }
        __syncthreads();
    } else {
        const int nqb = (a.Sq + BMQ - 1) / BMQ;
        q_first = (blockIdx.x % nqb) * BMQ;
        h = (blockIdx.x / nqb) % a.H;
        b = blockIdx.x / (nqb * a.H);
        n = (a.Skv + KT - 1) / KT;
    }

    const bf16_t* qp = (const bf16_t*)a.q + (long)b * a.q_bs + (long)h * a.q_hs;
    const bf16_t* kp = (const bf16_t*)a.k + (long)b * a.k_bs + (long)h * a.k_hs;
    const bf16_t* vtp = (const bf16_t*)a.vt + ((long)b * a.H + h) * 128L * a.Skv_pad;
    bf16_t* op = (bf16_t*)a.o + (long)b * a.o_bs + (long)h * a.o_hs;

    // ---- Q fragments (B operand of S^T = K·Q^T): row q0 + l31, d = 16*ks + 8*hi .. +8 ------------------------------------
    const int q0 = q_first + wave * 32;
    int qrow = q0 + l31;
    const bool q_ok = qrow < a.Sq;
    qrow = q_ok ? qrow : a.Sq - 1;
    bf16x8 qf[8];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) qf[ks] = ld_bf16x8(qp + (long)qrow * a.q_ss + ks * 16 + hi * 8);

    // ---- LDS-DMA: wave w moves K pieces {w, w+8, w+16, w+24} (4 key rows x 256 B each) and the same pieces of V^T (4 d rows x 256 B).
    // LDS image of both tiles: row r, 16-B chunk c at r*256 + ((c ^ (r&15)) << 4).  The hardware writes lane-linearly, so each lane
    // fetches the SOURCE chunk that belongs at its linear position.  Key rows >= Skv are out of the descriptor's range -> zeros.
    const __amdgpu_buffer_rsrc_t k_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)kp, 0, (unsigned)((((long)a.Skv - 1) * a.k_ss + 128) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t v_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)vtp, 0, (unsigned)(256L * a.Skv_pad), 0x00020000);
    // The TRAILING group (waves 4-7) issues every piece: its softmax segment has ~900 cycles of slack per tile, the leading group's
    // matrix segment has none.  Wave w' = wave-4 moves pieces {w', w'+4, ..., w'+28} of K and of V^T; rows r = 4*piece + (lane>>4),
    // and r & 15 does not depend on the piece index, so one swizzled base per tensor + a scalar stride addresses all eight.
    const int wq = wave & 3;
    const int r0 = 4 * wq + (lane >> 4);
    const unsigned sw0 = (unsigned)(((lane & 15) ^ (r0 & 15)) << 4);
    const unsigned kv0 = (unsigned)(((long)r0 * a.k_ss) * 2) + sw0, vv0 = (unsigned)(r0 * a.Skv_pad * 2) + sw0;
    const unsigned k_pstride = (unsigned)(16 * a.k_ss * 2), v_pstride = (unsigned)(16 * a.Skv_pad * 2);  // 16 rows between a wave's pieces
    const unsigned k_tile_bytes = (unsigned)(a.k_ss * 2 * KT);  // bytes between consecutive key tiles
    const int pdst = wq * 1024;                                 // + ring slot, + i*4096
    // LIST: a 256-B V^T piece row holds 64 keys of block A (source chunks 0-7) and 64 of block B (8-15): this lane's source chunk
    // picks the block, (chunk & 7) the 16 B inside it
    const unsigned csrc = (unsigned)((lane & 15) ^ (r0 & 15));
    const bool vhalf = csrc >= 8;
    const unsigned vvl = (unsigned)(r0 * a.Skv_pad * 2) + ((csrc & 7) << 4);
    const unsigned k_blk_bytes = (unsigned)(a.k_ss * 2 * 64);  // bytes between consecutive 64-key blocks
    // list entries: eA = tile j (its V^T pieces, its softmax mask), eB = tile j+1 (its K pieces), pre = tile j+2 (read one step ahead)
    int eA0 = 0, eA1 = 0, eB0 = 0, eB1 = 0, pre0 = 0, pre1 = 0;
    unsigned kB0 = 0, kB1 = 0, vsel = 0;
#define LIST_LOAD(T, D0, D1) { const int t_ = (T) < n ? (T) : 0; D0 = lst[2 * t_]; D1 = lst[2 * t_ + 1]; }
#define LIST_DERIVE()                                                                                                \
    {                                                                                                                \
        kB0 = (unsigned)(eB0 & 0xffffff) * k_blk_bytes;                                                              \
        kB1 = (unsigned)(eB1 & 0xffffff) * k_blk_bytes;                                                              \
        vsel = (unsigned)((vhalf ? eA1 : eA0) & 0xffffff) * 128u;                                                    \
    }

    // piece I (0..7 = K(T+1), 8..15 = V^T(T)) of the set that tile step T makes room for; tiles past the end re-read tile 0 (harmless)
#define ISSUE_PIECE(T, I)                                                                                            \
    if (!(ABL & 4)) {                                                                                                \
        if (LIST) {                                                                                                  \
            if ((I) < 8)                                                                                             \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(k_rsrc, (lds_void*)(smem + (((T) + 1) & 1) * K_TILE + pdst + (I) * 4096), 16, \
                                                         kv0 + ((I) & 3) * k_pstride + ((I) < 4 ? kB0 : kB1), 0, 0, 0); \
            else                                                                                                     \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(v_rsrc, (lds_void*)(smem + V_BASE + ((T) & 1) * V_TILE + pdst + ((I) - 8) * 4096), 16, \
                                                         vvl + ((I) - 8) * v_pstride + vsel, 0, 0, 0);              \
        } else if ((I) < 8) {                                                                                        \
            const int t_ = (T) + 1 < n ? (T) + 1 : 0;                                                                \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(k_rsrc, (lds_void*)(smem + (((T) + 1) & 1) * K_TILE + pdst + (I) * 4096), 16, \
                                                     kv0 + (I) * k_pstride + (unsigned)t_ * k_tile_bytes, 0, 0, 0);  \
        } else {                                                                                                     \
            const int t_ = (T) < n ? (T) : 0;                                                                        \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(v_rsrc, (lds_void*)(smem + V_BASE + ((T) & 1) * V_TILE + pdst + ((I) - 8) * 4096), 16, \
                                                     vv0 + ((I) - 8) * v_pstride, __builtin_amdgcn_readfirstlane(t_ * (KT * 2)), 0, 0); \
        }                                                                                                            \
    }
#define ISSUE_SET(T) { _Pragma("unroll") for (int i_ = 0; i_ < 16; ++i_) ISSUE_PIECE(T, i_) }
#define ISSUE_K(T) { _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_) ISSUE_PIECE(T, i_) }

    // ---- fragment read offsets (same swizzle for both tiles): row l31 (+32*block), k-chunk 2*step + hi ------------------------
    int foff[8];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) foff[ks] = l31 * 256 + (((2 * ks + hi) ^ (l31 & 15)) << 4);

    f32x16 o[4];
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
    f32x16 s[4];
    bf16x8 pf[8];
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float m_run = -1e30f, l_run = 0.f;
    const float c2 = a.scale * 1.4426950408889634f;

    // fragment i of the matrix segment of tile step J: i < 32: V^T(J-1) block (k-step i>>2, d-block i&3); else K(J) (k-step, key block)
#define FRAG(J, I) \
    (*reinterpret_cast<const bf16x8*>((I) < 32 ? smem + V_BASE + (((J) - 1) & 1) * V_TILE + foff[(I) >> 2] + ((I) & 3) * 8192 \
                                                : smem + ((J) & 1) * K_TILE + foff[((I) - 32) >> 2] + ((I) & 3) * 8192))
    // the matrix segment: FIRST..63; DMA != 0: this wave's 8 pieces of set J ride in the MFMA gaps
#define MSEG(J, FIRST, LAST, DMA)                                                                                    \
    {                                                                                                                \
        constexpr int FD = 6;                                                                                        \
        bf16x8 fr_[FD];                                                                                              \
        _Pragma("unroll") for (int i = 0; i < FD; ++i) fr_[i] = FRAG(J, (FIRST) + i);                                \
        __builtin_amdgcn_sched_group_barrier(0x100, FD, 0);                                                          \
        _Pragma("unroll") for (int i = (FIRST); i < (LAST); ++i) {                                                   \
            const int fs_ = (((ABL & 1) ? (i & ~1) : i) - (FIRST)) % FD; /* ABL bit 0: odd MFMAs re-use the even fragment */ \
            if (i < 32) o[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr_[fs_], pf[i >> 2], o[i & 3], 0, 0, 0); \
            else if (i < 36) s[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr_[fs_], qf[0], zero16, 0, 0, 0); /* S is not live during P·V */ \
            else s[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr_[fs_], qf[(i - 32) >> 2], s[i & 3], 0, 0, 0); \
            const int n_ = (ABL & 1) ? i + FD - 1 : i + FD;                                                          \
            const bool rd_ = n_ < (LAST) && (!(ABL & 1) || (i & 1));                                                 \
            if (rd_) fr_[(n_ - (FIRST)) % FD] = FRAG(J, n_);                                                         \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                       \
            if (rd_) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                              \
            /* DMA (trailing group only): the wave's 8 V^T pieces of set J+1 ride in the gaps of the Q·K^T half (one per 4 MFMAs); their   \
               ring slot held V^T(J-1), which this wave finished reading with MFMA 31 and the leading group a whole barrier interval ago */ \
            if ((DMA) && VSTREAM && grp == 1 && i >= 32 && ((i - 32) & 3) == 0) ISSUE_PIECE((J) + 1, 8 + ((i - 32) >> 2))          \
            /* LSTREAM (leading group): all 16 pieces of set J (its slots were freed by the barrier just passed), one per 4 MFMAs */      \
            if ((DMA) == 2 && LSTREAM && grp == 0 && (i & 3) == 0) ISSUE_PIECE((J), (i >> 2))                                         \
        }                                                                                                            \
    }
    // online softmax of tile J (row q = lane&31; this lane holds 64 of its 128 scores, lane^32 the other 64)
#define SOFTMAX(J)                                                                                                   \
    {                                                                                                                \
        if (LIST) {                                                                                                  \
            const int v0_ = eA0 >> 24, v1_ = eA1 >> 24; /* valid keys of the two 64-key blocks of this tile */          \
            if (v0_ < 64 || v1_ < 64) {                                                                              \
                _Pragma("unroll") for (int kb = 0; kb < 4; ++kb) _Pragma("unroll") for (int r = 0; r < 16; ++r) {   \
                    const int key = (kb & 1) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;                                 \
                    if (key >= (kb < 2 ? v0_ : v1_)) s[kb][r] = -INFINITY;                                           \
                }                                                                                                    \
            }                                                                                                        \
        }                                                                                                            \
        const int valid_ = LIST ? KT : a.Skv - (J) * KT;                                                             \
        if (valid_ < KT) {                                                                                           \
            _Pragma("unroll") for (int kb = 0; kb < 4; ++kb) _Pragma("unroll") for (int r = 0; r < 16; ++r) {       \
                const int key = kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;                                           \
                if (key >= valid_) s[kb][r] = -INFINITY;                                                             \
            }                                                                                                        \
        }                                                                                                            \
        float mx4[4];                                                                                                \
        _Pragma("unroll") for (int kb = 0; kb < 4; ++kb) {                                                           \
            mx4[kb] = fmaxf(s[kb][0], s[kb][1]);                                                                     \
            _Pragma("unroll") for (int r = 2; r < 16; r += 2) mx4[kb] = fmaxf(fmaxf(mx4[kb], s[kb][r]), s[kb][r + 1]); \
        }                                                                                                            \
        const float mx = xhalf_max(fmaxf(fmaxf(mx4[0], mx4[1]), fmaxf(mx4[2], mx4[3])));                             \
        const float m_new = fmaxf(m_run, mx);                                                                        \
        if (!__all(m_new == m_run)) {                                                                                \
            float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c2);                                              \
            asm volatile("s_nop 1" : "+v"(alpha)); /* trans-op result -> VALU read wait state: hipcc pads nothing for inline asm */ \
            l_run *= alpha;                                                                                          \
            /* single-issue v_mul_f32: hipcc SLP-packs these into v_pk_mul_f32, which crawls beside the partner wave's MFMAs  \
               (s_memtime probe: a wave taking this branch finished its segment ~1900 cycles late) */                  \
            _Pragma("unroll") for (int d = 0; d < 4; ++d) _Pragma("unroll") for (int r = 0; r < 16; ++r)             \
                asm volatile("v_mul_f32 %0, %1, %0" : "+v"(o[d][r]) : "v"(alpha));                                   \
            m_run = m_new;                                                                                           \
        }                                                                                                            \
        const float mc = m_run * c2;                                                                                 \
        float ps4[4] = {0.f, 0.f, 0.f, 0.f};                                                                         \
        _Pragma("unroll") for (int kb = 0; kb < 4; ++kb) _Pragma("unroll") for (int r = 0; r < 16; ++r) {           \
            const float e_ = __builtin_fmaf(s[kb][r], c2, -mc);                                                      \
            const float p = (ABL & 2) ? e_ : __builtin_amdgcn_exp2f(e_);                                            \
            s[kb][r] = p;                                                                                            \
            ps4[r & 3] += p;                                                                                         \
        }                                                                                                            \
        l_run += (ps4[0] + ps4[1]) + (ps4[2] + ps4[3]);                                                              \
        _Pragma("unroll") for (int kk = 0; kk < 8; ++kk) _Pragma("unroll") for (int jj = 0; jj < 8; ++jj)           \
            pf[kk][jj] = (bf16_t)s[kk >> 1][(kk & 1) * 8 + jj];                                                      \
        /* keep the packing inside this (VALU) segment: without a use here it is sunk below the barrier into the matrix segment */ \
        _Pragma("unroll") for (int kk = 0; kk < 8; ++kk) asm volatile("" : "+v"(pf[kk]));                           \
    }
#define WAIT_ALL() asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // VSTREAM: everything but the 8 youngest loads (the V^T pieces issued inside the segment that just ended) has landed
#define WAIT_SET() { if (VSTREAM) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
#define BAR()                                   \
    {                                           \
        __builtin_amdgcn_sched_barrier(0);      \
        __builtin_amdgcn_s_barrier();           \
        __builtin_amdgcn_sched_barrier(0);      \
    }

    // PROBE (timing only): every wave of workgroup 0 sums the s_memtime deltas between its segment boundaries over tiles 50..177 in
    // registers and writes the sums to a.lse (as uint64) after the loop.
    unsigned long long acc_[6] = {0, 0, 0, 0, 0, 0}, abs_[6] = {0, 0, 0, 0, 0, 0}, last_ = 0;
#define STAMP(K)                                                                                                     \
    if (PROBE && blockIdx.x == 0 && j >= 50 && j < 178) {                                                            \
        const unsigned long long now_ = __builtin_readcyclecounter();                                                \
        if ((K) != 0 || j > 50) acc_[K] += now_ - last_;                                                             \
        last_ = now_;                                                                                                \
        if (j == 100) abs_[K] = now_;                                                                                \
    }
    if (!LIST || n > 0) {  // an empty list (LIST only; workgroup-uniform): straight to the epilogue, which writes zeros
    if (LDMA) {
        // ---- leading-group DMA, one barrier per tile step --------------------------------------------------------------------------
        // leading : [set j in flight] M(j) . softmax(j) . wait(set j) . BARRIER(j) . issue set j+1 ...
        // trailing:                   ... softmax(j-1) . M(j) . BARRIER(j) . softmax(j) . M(j+1) ...
        if (LIST) {
            LIST_LOAD(0, eA0, eA1)
            LIST_LOAD(1, eB0, eB1)
            LIST_LOAD(2, pre0, pre1)
            eA0 = __builtin_amdgcn_readfirstlane(eA0); eA1 = __builtin_amdgcn_readfirstlane(eA1);
            eB0 = __builtin_amdgcn_readfirstlane(eB0); eB1 = __builtin_amdgcn_readfirstlane(eB1);
            LIST_DERIVE()
        }
        if (grp == 0) {
            const unsigned k00 = LIST ? (unsigned)(eA0 & 0xffffff) * k_blk_bytes : 0u, k01 = LIST ? (unsigned)(eA1 & 0xffffff) * k_blk_bytes : 4u * k_pstride;
#pragma unroll
            for (int i = 0; i < 8; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(k_rsrc, (lds_void*)(smem + pdst + i * 4096), 16,
                                                         kv0 + (i & 3) * k_pstride + (i < 4 ? k00 : k01), 0, 0, 0);
            ISSUE_SET(0)
            asm volatile("s_waitcnt vmcnt(16)" ::: "memory");  // K(0) landed; set 0 = {K(1), V(0)} flies
        }
        BAR()
        if (grp == 1) BAR()  // stagger: pairs with the leading group's barrier after M(0)
        {
            const int j = 0;
            (void)j;
            if (PRIO) __builtin_amdgcn_s_setprio(1);
            MSEG(0, 32, 64, 0)
            if (PRIO) __builtin_amdgcn_s_setprio(0);
            BAR()  // leading: releases the trailing group's M(0); trailing: barrier(0)
            SOFTMAX(0)
            if (grp == 0) {
                WAIT_ALL()  // set 0 landed
                BAR()       // barrier(0)
            }
        }
        for (int j = 1; j < n; ++j) {
            STAMP(0)
            if (LIST) {  // advance the list window: tile j -> eA, tile j+1 -> eB (read from LDS a step ago), tile j+2 -> in flight
                eA0 = eB0; eA1 = eB1;
                eB0 = __builtin_amdgcn_readfirstlane(pre0); eB1 = __builtin_amdgcn_readfirstlane(pre1);
                LIST_LOAD(j + 2, pre0, pre1)
                LIST_DERIVE()
            }
            if (grp == 0 && !LSTREAM) ISSUE_SET(j)  // slots free: both groups' M(j-1) are behind barrier(j-1)
            STAMP(1)
            if (PRIO) __builtin_amdgcn_s_setprio(1);
            MSEG(j, 0, 64, 2)
            if (PRIO) __builtin_amdgcn_s_setprio(0);
            STAMP(2)
            if (grp == 1) BAR()  // barrier(j) for the trailing group
            STAMP(3)
            SOFTMAX(j)
            STAMP(4)
            if (grp == 0) {
                WAIT_ALL()  // set j (issued ~4 000 cycles ago) landed
                BAR()       // barrier(j) for the leading group
            }
            STAMP(5)
        }
    } else {
    // ---- prologue: K(0) landed; set 0 = {K(1), V(0)} in flight (issued by the trailing group) -------------------------------------
    if (grp == 1) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(k_rsrc, (lds_void*)(smem + pdst + i * 4096), 16, kv0 + i * k_pstride, 0, 0, 0);
        ISSUE_SET(0)
        asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    }
    BAR()
    if (grp == 1) BAR()  // stagger: waves 4-7 run one barrier behind

    // ---- j = 0: Q·K^T only ---------------------------------------------------------------------------------------------------
    if (PRIO) __builtin_amdgcn_s_setprio(1);
    MSEG(0, 32, 64, 1)
    if (PRIO) __builtin_amdgcn_s_setprio(0);
    if (grp == 1) WAIT_SET()  // trailing group: set 0 landed before the barrier that precedes the leading group's M(1)
    BAR()                     // (ONEBAR: for the leading group this one still pairs with the trailing group's stagger barrier)
    if (grp == 1) {  // ring slots of set 1 are free: every wave has finished M(0)
        if (VSTREAM) ISSUE_K(1) else ISSUE_SET(1)
    }
    SOFTMAX(0)
    if (!ONEBAR || grp == 0) BAR()
    // ---- steady state ----------------------------------------------------------------------------------------------------------
    for (int j = 1; j < n; ++j) {
        STAMP(0)
        if (PRIO) __builtin_amdgcn_s_setprio(1);
        MSEG(j, 0, 64, 1)
        if (PRIO) __builtin_amdgcn_s_setprio(0);
        STAMP(1)
        if (grp == 1) WAIT_SET()         // set j landed (K(j+1) issued at the top of V(j-1); V^T(j) there or inside M(j-1))
        if (!ONEBAR || grp == 1) BAR()
        STAMP(2)
        if (grp == 1) {
            if (VSTREAM) ISSUE_K(j + 1) else ISSUE_SET(j + 1)
        }
        STAMP(3)
        SOFTMAX(j)
        STAMP(4)
        if (!ONEBAR || grp == 0) BAR()
        STAMP(5)
    }
    }  // !LDMA
    if (PROBE && blockIdx.x == 0 && lane == 0)
        for (int i = 0; i < 6; ++i) {
            reinterpret_cast<unsigned long long*>(a.lse)[wave * 8 + i] = acc_[i];
            reinterpret_cast<unsigned long long*>(a.lse)[64 + wave * 8 + i] = abs_[i];
        }
    // ---- j = n: last P·V ---------------------------------------------------------------------------------------------------------
    if (PRIO) __builtin_amdgcn_s_setprio(1);
    MSEG(n, 0, 32, 0)
    if (PRIO) __builtin_amdgcn_s_setprio(0);
    WAIT_ALL()           // drain the tail re-reads before this workgroup's LDS can be re-assigned
    if (!ONEBAR && grp == 0) BAR()  // matches the extra leading barrier of waves 4-7 (ONEBAR: every barrier is already paired)
    }  // n > 0
#undef LIST_LOAD
#undef LIST_DERIVE
#undef ISSUE_PIECE
#undef ISSUE_SET
#undef ISSUE_K
#undef WAIT_SET
#undef FRAG
#undef MSEG
#undef STAMP
#undef SOFTMAX
#undef WAIT_ALL
#undef BAR

    // ---- epilogue ------------------------------------------------------------------------------------------------------------------
    const float l_tot = xhalf_sum(l_run);
    const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
    int orow_i = qrow;
    if (LIST && la.o_rows && q_ok) orow_i = la.o_rows[qrow];  // un-grouping folded into the store
    if (q_ok && orow_i >= 0) {
        bf16_t* orow = op + (long)orow_i * a.o_ss;
#pragma unroll
        for (int d = 0; d < 4; ++d)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                bf16x4 v4;
#pragma unroll
                for (int e = 0; e < 4; ++e) v4[e] = (bf16_t)(o[d][g * 4 + e] * inv);
                *reinterpret_cast<bf16x4*>(orow + d * 32 + g * 8 + hi * 4) = v4;
            }
        if (!PROBE && a.lse && hi == 0) a.lse[((long)b * a.H + h) * a.Sq + qrow] = m_run * c2 + log2f(l_tot);
    }
#endif  // __HIP_DEVICE_COMPILE__
}

}  // namespace

// 256-row workgroups over shared KV block lists (called by fvk_attn_tile_lists_bf16, attn_fwd.hip, after its argument checks)
int fvk_attn_pp2_lists_launch(const fvk_attn_args* a, const fvk_pp2_lists* la, hipStream_t s) {
    constexpr int LDS_LIST = LDS_BYTES + 2048 * 8;  // + the packed list: up to 2048 tiles = 4096 blocks
    FVK_CHECK(la->max_kv <= 4096, FVK_ERR_ARG, "fvk_attn_tile_lists_bf16: lists of more than 4096 blocks (max_kv=%d) do not fit the LDS copy", la->max_kv);
    static FvkLdsConfigured configured;
    if (int rc = fvk_config_lds(configured, (const void*)attn_pp2_kernel<false, true, false, true, true, true, true>, LDS_LIST, "fvk_attn_tile_lists_bf16")) return rc;
    const long nblk = (long)la->n_lists * la->q_sub * a->H * a->B;
    hipLaunchKernelGGL((attn_pp2_kernel<false, true, false, true, true, true, true>), dim3((unsigned)nblk), dim3(512), LDS_LIST, s, *a, *la);
    FVK_LAUNCH_CHECK();
    return FVK_OK;
}

template <bool PROBE, bool PRIO, bool VSTREAM = true, bool ONEBAR = false, bool LDMA = false, bool LSTREAM = false, int ABL = 0>
static int launch_pp2(const fvk_attn_args* a, hipStream_t s) {
    static FvkLdsConfigured configured;
    if (int rc = fvk_config_lds(configured, (const void*)attn_pp2_kernel<PROBE, PRIO, VSTREAM, ONEBAR, LDMA, LSTREAM, false, ABL>, LDS_BYTES, "fvk_attn_dense_bf16 (pp2)")) return rc;
    const long nblk = (long)((a->Sq + 255) / 256) * a->H * a->B;
    hipLaunchKernelGGL((attn_pp2_kernel<PROBE, PRIO, VSTREAM, ONEBAR, LDMA, LSTREAM, false, ABL>), dim3((unsigned)nblk), dim3(512), LDS_BYTES, s, *a, fvk_pp2_lists{});
    FVK_LAUNCH_CHECK();
    return FVK_OK;
}

// probe != 0: timing probe build (a->lse receives s_memtime sums, see PROBE)
int fvk_attn_pp2_launch(const fvk_attn_args* a, int probe, hipStream_t s) {
#if FVK_VARIANTS
    switch (probe) {
        // 0 (shipped): one barrier per tile step, the LEADING group issues the DMA inside its matrix segment (round 2: -3 % time, -9 % cycles
        // against the round-1 schedule, bit-identical output); 1 = its s_memtime probe; 2 = shipped without s_setprio (A/B).
        // Earlier schedules, kept for A/B (scripts/attn_variants_check.py, attn_impl = 99 + k):
        //   4 / 5  round 1: two barriers, the trailing group issues all 16 pieces at the top of its softmax segment (5 = probe)
        //   12 / 3 two barriers, trailing group's V^T pieces inside its matrix segment (3 = probe)
        //   6 / 7  one barrier, trailing group issues (7 = probe);  8 / 9 one barrier, leading group issues AHEAD of its matrix segment
        case 1: return launch_pp2<true, true, false, true, true, true>(a, s);
        case 2: return launch_pp2<false, false, false, true, true, true>(a, s);
        case 3: return launch_pp2<true, true, true>(a, s);
        case 4: return launch_pp2<false, true, false>(a, s);
        case 5: return launch_pp2<true, true, false>(a, s);
        case 6: return launch_pp2<false, true, false, true>(a, s);
        case 7: return launch_pp2<true, true, false, true>(a, s);
        case 8: return launch_pp2<false, true, false, true, true, false>(a, s);
        case 9: return launch_pp2<true, true, false, true, true, false>(a, s);
        case 12: return launch_pp2<false, true, true>(a, s);
        // timing ablations of the shipped schedule (ABL bits: 1 half the fragment reads, 2 no v_exp, 4 no steady-state DMA): attn_impl 120 + bits
        case 21: return launch_pp2<false, true, false, true, true, true, 1>(a, s);
        case 22: return launch_pp2<false, true, false, true, true, true, 2>(a, s);
        case 23: return launch_pp2<false, true, false, true, true, true, 3>(a, s);
        case 24: return launch_pp2<false, true, false, true, true, true, 4>(a, s);
        case 25: return launch_pp2<false, true, false, true, true, true, 5>(a, s);
        case 27: return launch_pp2<false, true, false, true, true, true, 7>(a, s);
        default: break;
    }
#endif
    (void)probe;
    return launch_pp2<false, true, false, true, true, true>(a, s);  // shipped: one barrier per tile step, leading-group in-stream DMA
}
