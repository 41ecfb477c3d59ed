// Dense flash-style attention forward, head_dim 128, gfx950 — 4 waves x 64 query rows, ONE wave per SIMD, software-pipelined inside the wave.
//
// Why (profiles/r03_attn_ablate.md): the 8-wave ping-pong kernel (attn_pp2.hip) is power-bound, and a timing ablation in which every
// K / V^T fragment read from LDS feeds TWO MFMAs instead of one runs +25 % (1406 -> 1760 TF with the DMA off) — the LDS operand stream
// (one ds_read_b128 per MFMA, the same 64 KiB tile re-read by all 8 waves) is the largest single consumer after the matrix pipe.  A wave
// that owns 64 query rows (two 32-row blocks) reads each fragment once for two MFMAs: half the LDS reads and ds_read issue slots per
// FLOP.  64 rows need the whole 512-entry register file (Q 64 + O 128 + S 2 x 64 + P 2 x 32 + fragments), i.e. one wave per SIMD — so
// the softmax can no longer hide behind a partner wave's matrix segment and is software-pipelined INSIDE the wave instead:
//
//   sub-tile = 64 keys.  iteration t:   MFMA stream:  P·V of sub-tile t-1 (32)  then  Q·K^T of sub-tile t+1 (32)
//                                       VALU stream:  softmax of sub-tile t (exp2 against the fixed reference, row sums, bf16 pack)
//   the three are independent inside the iteration: 64 chunks of { 1 MFMA | 1 score: fma, exp, add, (cvt_pk) }, pinned with
//   sched_barrier — ~3.5 VALU instructions in the shadow of each 32-cycle MFMA.  S is double-buffered (Q·K^T(t+1) writes the other buffer).
//
// Register files, pinned by inline-asm constraints (every MFMA is asm): accumulator file: O 128 + Q fragments 64 (loaded there directly);
// arch VGPRs: S 2 x 64 + P 2 x 32 + 4 fragments in flight + softmax temporaries.  (hipcc selects the AGPR-accumulator form for EVERY MFMA
// builtin in a kernel that may use more than 256 registers — S would pay a v_accvgpr_read per score — and shuffles accumulators between the
// files when builtin and asm forms are mixed.)
//
// Fixed softmax reference: p = exp2((s - m_ref) * c) with m_ref = the exact row max of the FIRST sub-tile, never updated — the hot loop has
// no running max, no rescale of O and no branch (a rescale would have to touch the O accumulators with the VALU, which drags them out of
// the accumulator file).  Mathematically the same softmax (the reference cancels in O / l); P is no longer bounded by 1 but by
// 2^(row max - m_ref), which bf16 (P) and fp32 (l, O) absorb as long as that growth stays below ~2^90: checked ONCE, after the last key —
// a row whose l is not a finite number below 2^90 sends its whole workgroup through an exact, un-pipelined online-softmax pass (rescaling
// O as usual) that overwrites the result.  With RMS-normed q / k (the DiT) or randn inputs the growth is a few units; the recompute is
// there for arbitrary callers and is exercised by the spiked-key tests.  Results agree with attn_pp2 to rounding, not bit for bit.
//
// Staging is attn_pp2's: 128-key stages (K 32 KiB + V^T 32 KiB) by LDS-DMA into a 2-deep XOR-swizzled ring, ONE barrier per stage (a pair of
// iterations): behind the barrier that follows iteration 2j the slots of K(j) and V^T(j-1) are free and {K(j+2), V^T(j+1)} is issued — 16
// pieces per wave riding in the chunks of iteration 2j+1 — and waited for (vmcnt(0)) just before the next barrier, two iterations later.
// Same operand orientation as the other kernels (S^T = K·Q^T: a softmax row is lane-local; O^T = V^T·P^T with P^T taken from the packed
// S^T registers), same V^T input layout (fvk_v_transpose_bf16).
#include "fvk_common.h"
#include "attn_lists.h"

namespace {

constexpr int KT = 128;               // keys per staged tile (two 64-key sub-tiles)
constexpr int K_TILE = KT * 256;      // 128 keys x 128 d bf16
constexpr int V_TILE = 128 * KT * 2;  // 128 d x 128 keys bf16
constexpr int RING = 2;
constexpr int V_BASE = RING * K_TILE;
constexpr int LDS_BYTES = RING * (K_TILE + V_TILE);  // 131 072
constexpr float L_LIMIT = 1.2379400392853803e27f;  // 2^90: a row sum at or above it (or NaN) triggers the exact recompute

typedef __attribute__((address_space(3))) void lds_void;

__device__ __forceinline__ float xhalf_max64(float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float xhalf_sum64(float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

struct W64 {
    // registers of one wave (everything is indexed with compile-time constants after unrolling)
    bf16x8 qf[2][8];     // Q fragments [q block][d step]
    f32x16 o[2][4];      // O^T accumulators [q block][d block]
    f32x16 s[2][2][2];   // S^T [sub-tile parity][q block][32-key block]
    bf16x8 pf[2][2][4];  // P^T, packed [sub-tile parity][q block][16-key step]
    float m_run[2], l_run[2];
    int foff[8];
    float c2;
    // LDS-DMA addressing (attn_pp2's): wave w moves pieces {w, w+4, ..., w+28} of K (4 key rows x 256 B) and of V^T (4 d rows x 256 B)
    __amdgpu_buffer_rsrc_t k_rsrc, v_rsrc;
    unsigned kv0, vv0, k_pstride, v_pstride, k_tile_bytes;
    int pdst, n;
    unsigned char* smem;
    // LIST mode (fvk_attn_tile_lists_bf16): the workgroup's 64-key KV blocks, packed in LDS behind the ring as id | valid keys << 24, two per
    // stage (an absent second block re-reads the first with 0 valid keys).  Sub-tile t IS list entry t.
    const int32_t* lst;
    unsigned k_blk_bytes, vvl;  // bytes between 64-key blocks of K; this lane's V^T source offset without the block part
    bool vhalf;                 // this lane's 16-B chunk of a V^T piece row belongs to the stage's SECOND block
    unsigned kblk0, kblk1, vvs; // this pair's set: K byte offsets of the two blocks of stage j+2 (SGPR), V^T lane offset of stage j+1
    __device__ __forceinline__ int entry(int t) const { return lst[t < 2 * n ? t : 0]; }  // stages past the end re-read block 0 (harmless)
    // the DMA operands of pair j: K(j+2) from entries 2j+4, 2j+5; V^T(j+1) from entries 2j+2, 2j+3
    __device__ __forceinline__ void set_pair(int eK0, int eK1, int eV0, int eV1) {
        kblk0 = (unsigned)(eK0 & 0xffffff) * k_blk_bytes;
        kblk1 = (unsigned)(eK1 & 0xffffff) * k_blk_bytes;
        vvs = vvl + (unsigned)((vhalf ? eV1 : eV0) & 0xffffff) * 128u;
    }

    // piece I (0..7 = K(j+2), 8..15 = V^T(j+1)) of the set issued behind the barrier of pair j; stages past the end re-read stage 0 (harmless).
    // The per-piece and per-stage parts of the source address are wave-uniform: they ride in the SGPR offset, the lane part (kv0 / vv0) is ONE
    // arch VGPR per tensor for all pieces.
    template <bool LIST>
    __device__ __forceinline__ void issue_piece(int I, int j) const {  // I is a compile-time constant after unrolling
        if (LIST) {  // K pieces 0-3 / 4-7 come from the stage's first / second block; a V^T piece row holds 64 keys of each (per-lane block)
            if (I < 8)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(k_rsrc, (lds_void*)(smem + (j & 1) * K_TILE + pdst + I * 4096), 16, kv0,
                                                         __builtin_amdgcn_readfirstlane((I & 3) * k_pstride + (I < 4 ? kblk0 : kblk1)), 0, 0);
            else
                __builtin_amdgcn_raw_ptr_buffer_load_lds(v_rsrc, (lds_void*)(smem + V_BASE + ((j + 1) & 1) * V_TILE + pdst + (I - 8) * 4096), 16, vvs,
                                                         __builtin_amdgcn_readfirstlane((I - 8) * v_pstride), 0, 0);
        } else if (I < 8) {
            const int t_ = j + 2 < n ? j + 2 : 0;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(k_rsrc, (lds_void*)(smem + (j & 1) * K_TILE + pdst + I * 4096), 16, kv0,
                                                     __builtin_amdgcn_readfirstlane(I * k_pstride + (unsigned)t_ * k_tile_bytes), 0, 0);
        } else {
            const int t_ = j + 1 < n ? j + 1 : 0;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(v_rsrc, (lds_void*)(smem + V_BASE + ((j + 1) & 1) * V_TILE + pdst + (I - 8) * 4096), 16, vv0,
                                                     __builtin_amdgcn_readfirstlane((I - 8) * v_pstride + t_ * (KT * 2)), 0, 0);
        }
    }
    // V^T fragment of stage st: k-step kk (16 keys), d-block db;  K fragment of stage st: d-step ks, key block kb (32 keys)
    __device__ __forceinline__ bf16x8 frag_v(int st, int kk, int db) const {
        return *reinterpret_cast<const bf16x8*>(smem + V_BASE + (st & 1) * V_TILE + foff[kk] + db * 8192);
    }
    __device__ __forceinline__ bf16x8 frag_k(int st, int ks, int kb) const {
        return *reinterpret_cast<const bf16x8*>(smem + (st & 1) * K_TILE + foff[ks] + kb * 8192);
    }
    // Every MFMA is inline asm with pinned register files: O accumulators "+a" (accumulator file), S accumulators "+v" (arch VGPRs — the
    // VALU reads them), Q fragments "a", K / V^T / P fragments "v".  For a kernel that may use more than 256 registers hipcc selects the
    // AGPR-accumulator form for EVERY MFMA builtin (S would pay one v_accvgpr_read per score) and, with builtin and asm forms mixed, shuffles
    // the accumulators between the two files inside the loop; with asm only, each value's file is fixed by its constraints.
    // hipcc pads no hazards for asm statements: results are fenced by hand (fence_s / fence_o below) before compiler-generated readers.
#define FVK_MFMA1(ACC, FR, P) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(ACC) : "v"(FR), "v"(P))
#define FVK_MFMA_PV(ACC0, ACC1, FR, P0, P1)                                                                             \
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n\tv_mfma_f32_32x32x16_bf16 %1, %2, %4, %1" : "+a"(ACC0), "+a"(ACC1) : "v"(FR), "v"(P0), "v"(P1))
    // MFMA result -> compiler-generated reader (VALU / v_accvgpr_read) and VALU / v_accvgpr_write -> MFMA operand: 16-pass XDL op = 18 wait states
    template <int PAR>
    __device__ __forceinline__ void fence_s() { asm volatile("s_nop 15\n\ts_nop 3" : "+v"(s[PAR][0][0]), "+v"(s[PAR][0][1]), "+v"(s[PAR][1][0]), "+v"(s[PAR][1][1])); }
    __device__ __forceinline__ void fence_o() {
        asm volatile("s_nop 15\n\ts_nop 3" : "+a"(o[0][0]), "+a"(o[0][1]), "+a"(o[0][2]), "+a"(o[0][3]), "+a"(o[1][0]), "+a"(o[1][1]), "+a"(o[1][2]), "+a"(o[1][3]));
    }
    // P·V of the sub-tile (stage st, half hf) from pf[PAR]: plain form for the tail
    template <int PAR>
    __device__ __forceinline__ void pv_plain(int st, int hf) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int db = 0; db < 4; ++db) {
                const bf16x8 fr = frag_v(st, 4 * hf + kk, db);
                FVK_MFMA_PV(o[0][db], o[1][db], fr, pf[PAR][0][kk], pf[PAR][1][kk]);
            }
    }
    // keys of the sub-tile at or beyond `valid` (0..64) get -inf (last stage only)
    template <int PAR>
    __device__ __forceinline__ void mask_keys(int valid, int hi) {
#pragma unroll
        for (int qb = 0; qb < 2; ++qb)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    if (key >= valid) s[PAR][qb][kb][r] = -INFINITY;
                }
    }
    // row max of the sub-tile (this lane holds 32 of its row's 64 scores, lane^32 the other 32)
    template <int PAR, int QB>
    __device__ __forceinline__ float row_max() const {
        float mx = fmaxf(s[PAR][QB][0][0], s[PAR][QB][1][0]);
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(fmaxf(mx, s[PAR][QB][0][r]), s[PAR][QB][1][r]);
        return xhalf_max64(mx);
    }
    // p = exp2(s*c2 - mc) for values [8*kk .. 8*kk+8) of q block QB (one packed P fragment): the partial row sums go to ps4, P packed to bf16
    template <int PAR, int QB, int KK>
    __device__ __forceinline__ void exp_pack8(float mc, float (&ps4)[4]) {
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
            const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s[PAR][QB][KK >> 1][(KK & 1) * 8 + jj], c2, -mc));
            ps4[jj & 3] += p;
            pf[PAR][QB][KK][jj] = (bf16_t)p;
        }
    }
    template <int PAR, int QB>
    __device__ __forceinline__ float exp_pack(float mc) {
        float ps4[4] = {0.f, 0.f, 0.f, 0.f};
        exp_pack8<PAR, QB, 0>(mc, ps4);
        exp_pack8<PAR, QB, 1>(mc, ps4);
        exp_pack8<PAR, QB, 2>(mc, ps4);
        exp_pack8<PAR, QB, 3>(mc, ps4);
        return (ps4[0] + ps4[1]) + (ps4[2] + ps4[3]);
    }
    // exact online-softmax step (new running max first; O and l rescaled): first sub-tile and the slow path
    template <int PAR, bool RESCALE>
    __device__ __forceinline__ void softmax_exact() {
        float mx[2] = {row_max<PAR, 0>(), row_max<PAR, 1>()};
        if (RESCALE) fence_o();  // the P·V MFMAs just issued have written O before the compiler's v_accvgpr_read
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            const float m_new = fmaxf(m_run[qb], mx[qb]);
            const float alpha = __builtin_amdgcn_exp2f((m_run[qb] - m_new) * c2);
            if (RESCALE) {
#pragma unroll
                for (int d = 0; d < 4; ++d)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[qb][d][r] *= alpha;
            }
            m_run[qb] = m_new;
            const float ps = qb == 0 ? exp_pack<PAR, 0>(m_new * c2) : exp_pack<PAR, 1>(m_new * c2);
            l_run[qb] = l_run[qb] * alpha + ps;
        }
        if (RESCALE) fence_o();  // v_accvgpr_write -> MFMA source C
    }

    // Q·K^T of the sub-tile (ring slot st, half hf) -> s: plain form (exact pass)
    template <int PAR>
    __device__ __forceinline__ void qk_plain(int st, int hf) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const bf16x8 fr = frag_k(st, i >> 1, 2 * hf + (i & 1));
            if (i < 2)
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %2, %3, 0\n\tv_mfma_f32_32x32x16_bf16 %1, %2, %4, 0"
                             : "=&v"(s[PAR][0][i & 1]), "=&v"(s[PAR][1][i & 1]) : "v"(fr), "a"(qf[0][i >> 1]), "a"(qf[1][i >> 1]));
            else
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n\tv_mfma_f32_32x32x16_bf16 %1, %2, %4, %1"
                             : "+v"(s[PAR][0][i & 1]), "+v"(s[PAR][1][i & 1]) : "v"(fr), "a"(qf[0][i >> 1]), "a"(qf[1][i >> 1]));
        }
        fence_s<PAR>();
    }
    // The exact pass (rare: only when a row's fixed-reference sum left the safe range): plain online softmax over all keys with a running
    // max and O rescaled by the VALU, one stage at a time through ring slot 0 (load, wait, barrier, compute — no overlap).  Every wave of
    // the workgroup takes part.  Leaves o, m_run, l_run as the pipelined pass would have.
    template <bool LIST>
    __device__ __forceinline__ void exact_pass(int v_last, int hi) {
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            m_run[qb] = -1e30f;
            l_run[qb] = 0.f;
#pragma unroll
            for (int d = 0; d < 4; ++d)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[qb][d][r] = 0.f;
        }
        fence_o();
        for (int st = 0; st < n; ++st) {
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();  // every wave has finished reading slot 0
            __builtin_amdgcn_sched_barrier(0);
            int e0 = 0, e1 = 0;
            if (LIST) {
                e0 = __builtin_amdgcn_readfirstlane(entry(2 * st));
                e1 = __builtin_amdgcn_readfirstlane(entry(2 * st + 1));
                set_pair(e0, e1, e0, e1);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (LIST) {
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(k_rsrc, (lds_void*)(smem + pdst + i * 4096), 16, kv0,
                                                             __builtin_amdgcn_readfirstlane((i & 3) * k_pstride + (i < 4 ? kblk0 : kblk1)), 0, 0);
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(v_rsrc, (lds_void*)(smem + V_BASE + pdst + i * 4096), 16, vvs,
                                                             __builtin_amdgcn_readfirstlane(i * v_pstride), 0, 0);
                } else {
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(k_rsrc, (lds_void*)(smem + pdst + i * 4096), 16, kv0,
                                                             __builtin_amdgcn_readfirstlane(i * k_pstride + (unsigned)st * k_tile_bytes), 0, 0);
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(v_rsrc, (lds_void*)(smem + V_BASE + pdst + i * 4096), 16, vv0,
                                                             __builtin_amdgcn_readfirstlane(i * v_pstride + st * (KT * 2)), 0, 0);
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            const int valid = st == n - 1 ? v_last : KT;
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {  // (fully unrolled: a runtime index into foff[] would pin the whole register struct to scratch)
                const int vh = LIST ? ((hf ? e1 : e0) >> 24) : (valid - 64 * hf < 64 ? valid - 64 * hf : 64);  // valid keys of this sub-tile
                if (vh > 0) {
                    qk_plain<0>(0, hf);
                    if (vh < 64) mask_keys<0>(vh, hi);
                    softmax_exact<0, true>();
                    pv_plain<0>(0, hf);
                }
            }
        }
        fence_o();
    }

    // ---- iteration t = 2j + 1 + EVEN (pair j): 64 chunks of { 1 MFMA | the softmax of ONE score: fma, exp, add, every second chunk a
    // cvt_pk }, pinned with sched_barrier.  MFMA stream: P·V of sub-tile t-1 (stage j, k-steps 4*EVEN..+3, pf[EVEN]) then Q·K^T of sub-tile
    // t+1 (stage j+1, key blocks 2*EVEN, +1 -> s[EVEN]); each of the 32 fragments (read FD ahead) feeds two consecutive chunks (q block 0,
    // 1).  VALU stream: sub-tile t = s[1-EVEN] -> pf[1-EVEN], against the fixed reference.  EVEN = 0 also carries this wave's 16 DMA
    // pieces of set j ({K(j+2), V^T(j+1)}: their slots were freed by the pair's barrier), one per 4 chunks.
    // MASK: keys >= valid of sub-tile t are masked (last stage).
    // ABL (measurement build; results are wrong, timing is what is measured): bit 0 no DMA pieces in the loop, bit 2 no softmax VALU in the loop
    template <int EVEN, bool MASK, bool PIN, int ABL = 0, bool LIST = false>
    __device__ __forceinline__ void iter(int j, int valid, int hi) {
        constexpr int CUR = 1 - EVEN, FD = 6;
        if (LIST ? valid < 64 : MASK) {  // LIST: any block may be ragged (wave-uniform run-time test at the iteration's start)
            fence_s<CUR>();  // the previous iteration's last Q·K^T MFMAs wrote these registers a few instructions ago
            mask_keys<CUR>(valid, hi);
        }
        const float mc0 = m_run[0] * c2, mc1 = m_run[1] * c2;
        bf16x8 fr[FD];
#pragma unroll
        for (int i = 0; i < FD; ++i) fr[i] = frag_v(j, 4 * EVEN + (i >> 2), i & 3);
        float psA[2] = {0.f, 0.f}, psB[2] = {0.f, 0.f};
        // pipeline prologue of the softmax: exponential of score 0, exponent of score 1
        float p_nx = __builtin_amdgcn_exp2f(__builtin_fmaf(s[CUR][0][0][0], c2, -mc0));
        float e_nx = __builtin_fmaf(s[CUR][0][0][1], c2, -mc0);
#pragma unroll
        for (int c = 0; c < 64; ++c) {
            const int i = c >> 1;  // fragment: i < 16: V^T (k-step i>>2, d-block i&3); else K (d-step (i-16)>>1, key block (i-16)&1)
            if (i < 16) {
                if ((c & 1) == 0) FVK_MFMA1(o[0][i & 3], fr[i % FD], pf[EVEN][0][i >> 2]);
                else FVK_MFMA1(o[1][i & 3], fr[i % FD], pf[EVEN][1][i >> 2]);
            } else {
                const int ks = (i - 16) >> 1, kb = (i - 16) & 1;
                if (ks == 0) {
                    if ((c & 1) == 0) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(s[EVEN][0][kb]) : "v"(fr[i % FD]), "a"(qf[0][ks]));
                    else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(s[EVEN][1][kb]) : "v"(fr[i % FD]), "a"(qf[1][ks]));
                } else {
                    if ((c & 1) == 0) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(s[EVEN][0][kb]) : "v"(fr[i % FD]), "a"(qf[0][ks]));
                    else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(s[EVEN][1][kb]) : "v"(fr[i % FD]), "a"(qf[1][ks]));
                }
            }
            if (PIN) __builtin_amdgcn_sched_barrier(0);  // the MFMA FIRST: everything below runs in its shadow (hoisted above it, it would delay the issue)
            if (c & 1) {
                const int nx = i + FD;
                if (nx < 32) fr[i % FD] = nx < 16 ? frag_v(j, 4 * EVEN + (nx >> 2), nx & 3) : frag_k(j + 1, (nx - 16) >> 1, 2 * EVEN + ((nx - 16) & 1));
            }
            if (EVEN == 0 && (c & 3) == 1 && !(ABL & 1)) issue_piece<LIST>(c >> 2, j);
            if (!(ABL & 4)) {
                // the softmax, software-pipelined over the chunks so that the three VALU instructions of a chunk are INDEPENDENT (one wave per
                // SIMD: nothing else hides the latency of fma -> exp -> add on one score): chunk c finishes score c (row sum, bf16 pack),
                // takes the exponential of score c+1 and forms the exponent of score c+2.  Score x of the lane's 64: q block x>>5, P fragment
                // (x>>3)&3, value x&7.
                {
                    const int qb = c >> 5, kk = (c >> 3) & 3, jj = c & 7;
                    if (qb == 0) psA[c & 1] += p_nx; else psB[c & 1] += p_nx;
                    pf[CUR][qb][kk][jj] = (bf16_t)p_nx;
                    if (jj == 7) asm volatile("" : "+v"(pf[CUR][qb][kk]));
                }
                if (c + 1 < 64) p_nx = __builtin_amdgcn_exp2f(e_nx);
                if (c + 2 < 64) {
                    const int x = c + 2, qb = x >> 5, kk = (x >> 3) & 3, r = (kk & 1) * 8 + (x & 7);
                    e_nx = __builtin_fmaf(qb == 0 ? s[CUR][0][kk >> 1][r] : s[CUR][1][kk >> 1][r], c2, qb == 0 ? -mc0 : -mc1);
                }
                // keep the chunk's work IN the chunk (a use here, before the scheduling barrier)
                asm volatile("" : "+v"(p_nx), "+v"(e_nx));
                if ((c >> 5) == 0) asm volatile("" : "+v"(psA[0]), "+v"(psA[1])); else asm volatile("" : "+v"(psB[0]), "+v"(psB[1]));
            }
            if (PIN) __builtin_amdgcn_sched_barrier(0);
        }
        l_run[0] += psA[0] + psA[1];
        l_run[1] += psB[0] + psB[1];
    }
#undef FVK_MFMA_PV
#undef FVK_MFMA1
};

// SPLIT (fvk_attn_dense_split_bf16): the key axis is cut into `n_split` runs of whole 128-key stages and every (query block, head, batch,
// run) is its own workgroup — for grids too small to fill 256 CUs (sequence-parallel ranks: 192 query-block workgroups at SP = 8).  A
// run's result is written UN-merged: normalised O as fp32 rows into o_part[run][b][h][row][128] and its base-2 LSE into
// lse_part[run][b][h][row]; attn_merge_splits_kernel combines the runs (a run with no keys writes LSE = -inf and is ignored).
// LIST (fvk_attn_tile_lists_bf16): every `q_stride` consecutive query rows share ONE list of 64-key KV blocks (a sliding-tile window); a
// workgroup owns 256 of those rows and sub-tile t of its key walk IS list entry t — K pieces add their block's row offset, a V^T piece row
// takes 64 keys from each of the stage's two blocks (per-lane block choice), every sub-tile is masked by its own block size, output rows may
// be scattered (o_rows).  The list is copied to LDS behind the ring once; entries are read a whole pair ahead.
template <bool PIN, int ABL = 0, bool PLAIN_IDS = false, bool SPLIT = false, bool LIST = false>
__global__ __launch_bounds__(256, 1) void attn_w64_kernel(fvk_attn_args a, int n_split, float* o_part, float* lse_part, fvk_pp2_lists la) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int BMQ = 256;
    FVK_CLAIM_WHOLE_REGISTER_FILE();   // (488 of 512 registers used: nothing real fits beside it anyway; claimed for the rule's sake, fvk_common.h)
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const int nqb = (a.Sq + BMQ - 1) / BMQ;
    // XCD-aware deal: hardware workgroup id x lands on XCD x % 8 (its own L2); XCD c gets the CONTIGUOUS logical ids [c*q + min(c, r), ...), i.e.
    // consecutive query blocks of the same head, which stream the same K / V^T
    const int nblk = gridDim.x, xq = nblk >> 3, xr = nblk & 7, xcd = blockIdx.x & 7;
    const int bid = (PLAIN_IDS || (LIST && la.plain_ids)) ? (int)blockIdx.x : xcd * xq + (xcd < xr ? xcd : xr) + (int)(blockIdx.x >> 3);
    int q_first, h, b, lst_num = 0;
    int32_t* const lst = reinterpret_cast<int32_t*>(smem + LDS_BYTES);  // LIST: [stage][2] packed entries
    if (LIST) {
        const int per_head = la.n_lists * la.q_sub;
        const int u = bid % per_head, sub_ = u % la.q_sub, li = u / la.q_sub;
        h = (bid / per_head) % a.H;
        b = bid / (per_head * a.H);
        q_first = li * la.q_stride + sub_ * BMQ;
        const long meta = ((long)b * a.H + h) * la.n_lists + li;
        lst_num = la.q2k_num[meta];
        if (la.q_rows_valid && sub_ * BMQ >= la.q_rows_valid[li]) lst_num = 0;  // only padding rows: nothing to do
        const int32_t* src = la.q2k_idx + meta * la.max_kv;
        for (int t = tid; t < (lst_num + 1) >> 1; t += 256) {
            const int id0 = src[2 * t];
            const bool two = 2 * t + 1 < lst_num;
            const int id1 = two ? src[2 * t + 1] : id0;  // an absent second block re-reads the first, with no valid key
            lst[2 * t] = id0 | (la.kv_block_sizes[id0] << 24);
            lst[2 * t + 1] = id1 | ((two ? la.kv_block_sizes[id1] : 0) << 24);
        }
        __syncthreads();
    } else {
        q_first = (bid % nqb) * BMQ;
        h = (bid / nqb) % a.H;
        b = (bid / (nqb * a.H)) % a.B;
    }
    const int run = SPLIT ? bid / (nqb * a.H * a.B) : 0;
    // this workgroup's keys: stages [st0, st1) of the (b, h) slice; Skv_w = its key count (the last stage of the last run may be ragged)
    const int n_all = (a.Skv + KT - 1) / KT;
    const int st0 = SPLIT ? (int)((long)n_all * run / n_split) : 0, st1 = SPLIT ? (int)((long)n_all * (run + 1) / n_split) : n_all;
    const int n = LIST ? (lst_num + 1) >> 1 : st1 - st0;  // stages of this workgroup (SPLIT / LIST: may be 0)
    const int Skv_w = LIST ? a.Skv : (st1 * KT < a.Skv ? st1 * KT : a.Skv) - st0 * KT;
    const int v_last = Skv_w - (n - 1) * KT;      // dense: valid keys of the last stage, 1..128 (LIST: unused — every entry has its own size)

    const bf16_t* qp = (const bf16_t*)a.q + (long)b * a.q_bs + (long)h * a.q_hs;
    const bf16_t* kp = (const bf16_t*)a.k + (long)b * a.k_bs + (long)h * a.k_hs + (long)st0 * KT * a.k_ss;
    const bf16_t* vtp = (const bf16_t*)a.vt + ((long)b * a.H + h) * 128L * a.Skv_pad + (long)st0 * KT;
    bf16_t* op = (bf16_t*)a.o + (long)b * a.o_bs + (long)h * a.o_hs;
    if (SPLIT && n <= 0) {  // an empty run: LSE = -inf (weight 0 in the merge); workgroup-uniform
        const int r = q_first + tid;
        if (r < a.Sq) lse_part[(((long)run * a.B + b) * a.H + h) * a.Sq + r] = -INFINITY;
        return;
    }

    if (LIST && n <= 0) {  // an empty list (workgroup-uniform): zero rows, LSE = -inf
        for (int r = tid; r < BMQ; r += 256) {
            const int qr = q_first + r;
            if (qr >= a.Sq) break;
            const int orow_i = la.o_rows ? la.o_rows[qr] : qr;
            if (orow_i < 0) continue;
            bf16_t* orow = op + (long)orow_i * a.o_ss;
            const bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int c = 0; c < 16; ++c) st_bf16x8(orow + c * 8, z);
            if (a.lse) a.lse[((long)b * a.H + h) * a.Sq + qr] = -INFINITY;
        }
        return;
    }

    W64 w;
    w.smem = smem;
    w.lst = lst;
    w.n = n;
    w.c2 = a.scale * 1.4426950408889634f;
    // Q fragments of the wave's two 32-row blocks (B operand of S^T = K·Q^T): row q0 + 32*blk + l31, d = 16*ks + 8*hi .. +8
    const int q0 = q_first + wave * 64;
    int qrow[2];
    bool q_ok[2];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const int r = q0 + 32 * qb + l31;
        q_ok[qb] = r < a.Sq;
        qrow[qb] = q_ok[qb] ? r : a.Sq - 1;
        // loaded STRAIGHT into the accumulator file (gfx950 loads can target AGPRs): a value that is defined in an "a" register and only
        // ever used as an "a" operand stays there; loaded with a plain C++ load it lives in arch VGPRs and is copied (v_accvgpr_write)
        // in front of every MFMA.  The loads are invisible to the compiler's vmcnt bookkeeping: the prologue's s_waitcnt vmcnt(0) covers them.
#pragma unroll
        for (int ks = 0; ks < 8; ++ks)
            asm volatile("global_load_dwordx4 %0, %1, off" : "=a"(w.qf[qb][ks]) : "v"(qp + (long)qrow[qb] * a.q_ss + ks * 16 + hi * 8) : "memory");
    }
    // LDS image of both tiles: row r, 16-B chunk c at r*256 + ((c ^ (r&15)) << 4); the hardware writes lane-linearly, so each lane fetches
    // the SOURCE chunk that belongs at its linear position.  Key rows >= Skv are outside the descriptor's range -> zeros.
    w.k_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)kp, 0, (unsigned)((((long)Skv_w - 1) * a.k_ss + 128) * 2), 0x00020000);
    w.v_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)vtp, 0, (unsigned)(256L * a.Skv_pad - 2L * st0 * KT), 0x00020000);
    const int r0 = 4 * wave + (lane >> 4);
    const unsigned sw0 = (unsigned)(((lane & 15) ^ (r0 & 15)) << 4);
    w.kv0 = (unsigned)(((long)r0 * a.k_ss) * 2) + sw0;
    w.vv0 = (unsigned)(r0 * a.Skv_pad * 2) + sw0;
    w.k_pstride = (unsigned)(16 * a.k_ss * 2);
    w.v_pstride = (unsigned)(16 * a.Skv_pad * 2);
    w.k_tile_bytes = (unsigned)(a.k_ss * 2 * KT);
    w.k_blk_bytes = (unsigned)(a.k_ss * 2 * 64);
    {   // LIST: this lane's source chunk of a V^T piece row picks the block (chunks 8-15 = the stage's second block), (chunk & 7) the 16 B inside it
        const unsigned csrc = (unsigned)((lane & 15) ^ (r0 & 15));
        w.vhalf = csrc >= 8;
        w.vvl = (unsigned)(r0 * a.Skv_pad * 2) + ((csrc & 7) << 4);
    }
    w.pdst = wave * 1024;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) w.foff[ks] = l31 * 256 + (((2 * ks + hi) ^ (l31 & 15)) << 4);
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        w.m_run[qb] = -1e30f;
        w.l_run[qb] = 0.f;
#pragma unroll
        for (int d = 0; d < 4; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) w.o[qb][d][r] = 0.f;
    }

    w.fence_o();  // zero-initialised accumulators (v_accvgpr_write) -> first MFMA
#define WAIT_ALL() asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#define WAIT8() asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // everything but this wave's youngest set (8 pieces) has landed
#define BAR()                                   \
    {                                           \
        __builtin_amdgcn_sched_barrier(0);      \
        __builtin_amdgcn_s_barrier();           \
        __builtin_amdgcn_sched_barrier(0);      \
    }
    // ---- prologue: K(0), V^T(0) -> slot 0, K(1) -> slot 1; Q·K^T(0), the reference + softmax(0), Q·K^T(1) --------------------------------
    // LIST: the entries of the stage whose second block is softmaxed first (A1), of stage j+1 (B0, B1) and of stage j+2 (C0, C1), wave-uniform;
    // P0 / P1 = the entries of stage j+3, read from LDS a whole pair before they are needed
    int eA1 = 0, eB0 = 0, eB1 = 0, eC0 = 0, eC1 = 0, eP0 = 0, eP1 = 0;
    if (LIST) {
        const int e00 = __builtin_amdgcn_readfirstlane(w.entry(0));
        eA1 = __builtin_amdgcn_readfirstlane(w.entry(1));
        eB0 = __builtin_amdgcn_readfirstlane(w.entry(2));
        eB1 = __builtin_amdgcn_readfirstlane(w.entry(3));
        eC0 = __builtin_amdgcn_readfirstlane(w.entry(4));
        eC1 = __builtin_amdgcn_readfirstlane(w.entry(5));
        eP0 = w.entry(6);
        eP1 = w.entry(7);
        w.set_pair(e00, eA1, e00, eA1);  // stage 0 as a "set": K(0) -> slot 0 (J = 0: pieces 0-7), V^T(0) -> slot 0 (J = -1: pieces 8-15)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            w.issue_piece<true>(i, 0);
            w.issue_piece<true>(8 + i, -1);
        }
        w.set_pair(eB0, eB1, 0, 0);      // K(1) -> slot 1 (J = 1)
#pragma unroll
        for (int i = 0; i < 8; ++i) w.issue_piece<true>(i, 1);
    }
#pragma unroll
    for (int i = 0; i < 8 && !LIST; ++i) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(w.k_rsrc, (lds_void*)(smem + w.pdst + i * 4096), 16, w.kv0,
                                                 __builtin_amdgcn_readfirstlane(i * w.k_pstride), 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(w.k_rsrc, (lds_void*)(smem + K_TILE + w.pdst + i * 4096), 16, w.kv0,
                                                 __builtin_amdgcn_readfirstlane(i * w.k_pstride + (n > 1 ? w.k_tile_bytes : 0u)), 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(w.v_rsrc, (lds_void*)(smem + V_BASE + w.pdst + i * 4096), 16, w.vv0,
                                                 __builtin_amdgcn_readfirstlane(i * w.v_pstride), 0, 0);
    }
    WAIT_ALL()
    BAR()
    const int v0 = v_last < 64 ? v_last : 64;  // dense: valid keys of the last stage's two sub-tiles
    int v1 = v_last > 64 ? v_last - 64 : 0;
    w.qk_plain<0>(0, 0);
    if (LIST) {
        const int sz0 = __builtin_amdgcn_readfirstlane(w.entry(0)) >> 24;
        if (sz0 < 64) w.mask_keys<0>(sz0, hi);
    } else if (n == 1) w.mask_keys<0>(v0, hi);
    // the fixed reference: exact row max of the first sub-tile (it has at least one valid key)
    w.m_run[0] = w.row_max<0, 0>();
    w.m_run[1] = w.row_max<0, 1>();
    w.l_run[0] = w.exp_pack<0, 0>(w.m_run[0] * w.c2);
    w.l_run[1] = w.exp_pack<0, 1>(w.m_run[1] * w.c2);
    w.qk_plain<1>(0, 1);
    // ---- pairs j = 0 .. n-2: iterations t = 2j+1 and 2j+2 behind ONE barrier; the last pair masks sub-tile 2n-2 ------------------------------
    for (int j = 0; j + (LIST ? 1 : 2) < n; ++j) {  // straight-line body: a conditional inside would make the register assignment of the two paths meet with copies
        if (!(ABL & 2)) {
            WAIT_ALL()  // this wave's pieces of set j-1 (issued a whole pair ago) have landed
            BAR()       // every wave is past iteration 2j: the slots of K(j) and V^T(j-1) are free, set j-1 is visible
        }
        if (LIST) {
            w.set_pair(eC0, eC1, eB0, eB1);                       // this pair's set: K(j+2), V^T(j+1)
            const int a1 = eA1 >> 24, b0 = eB0 >> 24;             // valid keys of sub-tiles 2j+1 and 2j+2
            eA1 = eB1; eB0 = eC0; eB1 = eC1;                      // the window moves one stage
            eC0 = __builtin_amdgcn_readfirstlane(eP0);
            eC1 = __builtin_amdgcn_readfirstlane(eP1);
            eP0 = w.entry(2 * j + 8);                             // stage j+4: needed as C in the pair after next
            eP1 = w.entry(2 * j + 9);
            w.iter<0, false, PIN, ABL, true>(j, a1, hi);
            w.iter<1, false, PIN, ABL, true>(j, b0, hi);
        } else {
            w.iter<0, false, PIN, ABL>(j, 64, hi);
            w.iter<1, false, PIN, ABL>(j, 64, hi);
        }
    }
    if (!LIST && n >= 2) {
        WAIT_ALL()
        BAR()
        w.iter<0, false, PIN>(n - 2, 64, hi);
        w.iter<1, true, PIN>(n - 2, v0, hi);
    }
    if (LIST) v1 = eA1 >> 24;  // the last stage's second block
    // ---- tail: P·V(2n-2), softmax of sub-tile 2n-1 (second half of the last stage, masked), P·V(2n-1) --------------------------------------
    WAIT_ALL()  // V^T(n-1) (and the harmless re-reads of stage 0) landed
    BAR()
    w.pv_plain<0>(n - 1, 0);
    if (v1 > 0) {  // workgroup-uniform
        w.fence_s<1>();
        w.mask_keys<1>(v1, hi);
        w.l_run[0] += w.exp_pack<1, 0>(w.m_run[0] * w.c2);
        w.l_run[1] += w.exp_pack<1, 1>(w.m_run[1] * w.c2);
        w.pv_plain<1>(n - 1, 1);
    }
    w.fence_o();  // the last MFMAs' results before the epilogue's v_accvgpr_read
    WAIT_ALL()  // the harmless re-reads of stage 0 have landed (the exact pass below re-uses the ring; afterwards the LDS can be re-assigned)
    // ---- epilogue: normalise and store; a row whose fixed-reference sum left the safe range (NaN, infinite or >= 2^90) is redone by the
    // exact pass and stored again — per ROW, so that a row's result never depends on which other rows share its wave or workgroup (the
    // sequence-parallel paths rely on that: a different partition groups the rows differently).  The whole workgroup takes part in the
    // exact pass (staging, barriers); only the flagged rows are overwritten.
    bool redo[2] = {false, false};
#define FVK_STORE_ROWS(ONLY_REDO)                                                                                    \
    _Pragma("unroll") for (int qb = 0; qb < 2; ++qb) {                                                               \
        const float l_tot = xhalf_sum64(w.l_run[qb]);                                                                \
        const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;                                                          \
        if (!(ONLY_REDO)) redo[qb] = !(l_tot < L_LIMIT);                                                             \
        if (q_ok[qb] && (!(ONLY_REDO) || redo[qb])) {                                                                \
            if (SPLIT) {                                                                                             \
                const long prow = (((long)run * a.B + b) * a.H + h) * a.Sq + qrow[qb];                               \
                float* orow = o_part + prow * 128;                                                                   \
                _Pragma("unroll") for (int d = 0; d < 4; ++d) _Pragma("unroll") for (int g = 0; g < 4; ++g) {       \
                    f32x4 v4;                                                                                        \
                    _Pragma("unroll") for (int e = 0; e < 4; ++e) v4[e] = w.o[qb][d][g * 4 + e] * inv;              \
                    *reinterpret_cast<f32x4*>(orow + d * 32 + g * 8 + hi * 4) = v4;                                  \
                }                                                                                                    \
                if (hi == 0) lse_part[prow] = w.m_run[qb] * w.c2 + log2f(l_tot);                                     \
            } else if (!LIST || !la.o_rows || la.o_rows[qrow[qb]] >= 0) { /* LIST: un-grouping folded into the store */    \
                bf16_t* orow = op + (long)((LIST && la.o_rows) ? la.o_rows[qrow[qb]] : qrow[qb]) * a.o_ss;           \
                _Pragma("unroll") for (int d = 0; d < 4; ++d) _Pragma("unroll") for (int g = 0; g < 4; ++g) {       \
                    bf16x4 v4;                                                                                       \
                    _Pragma("unroll") for (int e = 0; e < 4; ++e) v4[e] = (bf16_t)(w.o[qb][d][g * 4 + e] * inv);    \
                    *reinterpret_cast<bf16x4*>(orow + d * 32 + g * 8 + hi * 4) = v4;                                 \
                }                                                                                                    \
                if (a.lse && hi == 0) a.lse[((long)b * a.H + h) * a.Sq + qrow[qb]] = w.m_run[qb] * w.c2 + log2f(l_tot); \
            }                                                                                                        \
        }                                                                                                            \
    }
    FVK_STORE_ROWS(false)
    if (__syncthreads_or(redo[0] || redo[1])) {
        w.exact_pass<LIST>(v_last, hi);
        FVK_STORE_ROWS(true)
    }
#undef FVK_STORE_ROWS
#endif  // __HIP_DEVICE_COMPILE__
}

template <bool PIN, int ABL = 0, bool PLAIN_IDS = false>
int launch_w64(const fvk_attn_args* a, hipStream_t s) {
    static FvkLdsConfigured configured;
    if (int rc = fvk_config_lds(configured, (const void*)attn_w64_kernel<PIN, ABL, PLAIN_IDS>, LDS_BYTES, "fvk_attn_dense_bf16 (w64)")) return rc;
    const long nblk = (long)((a->Sq + 255) / 256) * a->H * a->B;
    hipLaunchKernelGGL((attn_w64_kernel<PIN, ABL, PLAIN_IDS>), dim3((unsigned)nblk), dim3(256), LDS_BYTES, s, *a, 1, (float*)nullptr, (float*)nullptr, fvk_pp2_lists{});
    FVK_LAUNCH_CHECK();
    return FVK_OK;
}


#if FVK_VARIANTS  // (only the measurement build's split-KV launch uses it)
// out[b, row, h, :] = sum_r 2^(lse_r - max) * o_part[r] / sum_r 2^(lse_r - max): one wave per (b, h, row), 2 columns per lane; HBM-bound
// (reads n_split * 512 B, writes 256 B per row).
__global__ __launch_bounds__(256) void attn_merge_splits_kernel(fvk_attn_args a, int n_split, const float* o_part, const float* lse_part) {
    const long row_id = (long)blockIdx.x * 4 + (threadIdx.x >> 6);  // over B * H * Sq
    const long rows = (long)a.B * a.H * a.Sq;
    if (row_id >= rows) return;
    const int lane = threadIdx.x & 63;
    const int r = (int)(row_id % a.Sq), h = (int)((row_id / a.Sq) % a.H), b = (int)(row_id / ((long)a.Sq * a.H));
    float mx = -INFINITY;
    for (int s_ = 0; s_ < n_split; ++s_) mx = fmaxf(mx, lse_part[(long)s_ * rows + row_id]);
    float acc0 = 0.f, acc1 = 0.f, wsum = 0.f;
    for (int s_ = 0; s_ < n_split; ++s_) {
        const float wgt = exp2f(lse_part[(long)s_ * rows + row_id] - mx);  // an empty run: 2^(-inf) = 0
        if (wgt > 0.f) {
            const float2 v = *reinterpret_cast<const float2*>(o_part + ((long)s_ * rows + row_id) * 128 + lane * 2);
            acc0 += wgt * v.x;
            acc1 += wgt * v.y;
            wsum += wgt;
        }
    }
    const float inv = wsum > 0.f ? 1.0f / wsum : 0.f;
    bf16x2 o2;
    o2[0] = (bf16_t)(acc0 * inv);
    o2[1] = (bf16_t)(acc1 * inv);
    *reinterpret_cast<bf16x2*>((bf16_t*)a.o + (long)b * a.o_bs + (long)h * a.o_hs + (long)r * a.o_ss + lane * 2) = o2;
    if (a.lse && lane == 0) a.lse[row_id] = mx + log2f(wsum);
}

#endif

}  // namespace

// variant (measurement build): 2 = hardware workgroup order, 11.. = timing ablations.  (A variant WITHOUT the scheduling barriers was tried and
// removed: free to move the softmax's reads of S next to the asm MFMAs that write it — whose latency the compiler does not know — it
// produced wrong rows at some shapes.)
int fvk_attn_w64_launch(const fvk_attn_args* a, int variant, hipStream_t s) {
#if FVK_VARIANTS
    switch (variant) {
        case 2: return launch_w64<true, 0, true>(a, s);  // hardware workgroup order (A/B of the XCD-contiguous deal)
        // timing ablations (attn_impl 210 + bits: 1 no DMA in the loop, 2 no barrier / wait in the loop, 4 no softmax VALU in the loop)
        case 11: return launch_w64<true, 1>(a, s);
        case 12: return launch_w64<true, 2>(a, s);
        case 13: return launch_w64<true, 3>(a, s);
        case 14: return launch_w64<true, 4>(a, s);
        case 17: return launch_w64<true, 7>(a, s);
        default: break;
    }
#endif
    (void)variant;
    return launch_w64<true>(a, s);
}

#if FVK_VARIANTS  // measurement build: the shipped split-KV form is attn_w16's
// split-KV form (called by fvk_attn_dense_split_bf16, attn_fwd.hip, after its argument checks)
int fvk_attn_w64_split_launch(const fvk_attn_args* a, int n_split, float* o_part, float* lse_part, hipStream_t s) {
    static FvkLdsConfigured configured;
    if (int rc = fvk_config_lds(configured, (const void*)attn_w64_kernel<true, 0, false, true>, LDS_BYTES, "fvk_attn_dense_split_bf16")) return rc;
    const long nblk = (long)((a->Sq + 255) / 256) * a->H * a->B * n_split;
    hipLaunchKernelGGL((attn_w64_kernel<true, 0, false, true>), dim3((unsigned)nblk), dim3(256), LDS_BYTES, s, *a, n_split, o_part, lse_part, fvk_pp2_lists{});
    FVK_LAUNCH_CHECK();
    const long rows = (long)a->B * a->H * a->Sq;
    hipLaunchKernelGGL(attn_merge_splits_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, *a, n_split, (const float*)o_part, (const float*)lse_part);
    FVK_LAUNCH_CHECK();
    return FVK_OK;
}

// measurement build only: attn_fwd.hip (fvk_attn_tile_lists_bf16) says why the list mode is not shipped
// 256-row workgroups over shared KV block lists (called by fvk_attn_tile_lists_bf16, attn_fwd.hip, after its argument checks)
int fvk_attn_w64_lists_launch(const fvk_attn_args* a, const fvk_pp2_lists* la, hipStream_t s) {
    constexpr int LDS_LIST = LDS_BYTES + 2048 * 8;  // + the packed list: up to 2048 stages = 4096 blocks
    FVK_CHECK(la->max_kv <= 4096, FVK_ERR_ARG, "fvk_attn_tile_lists_bf16: lists of more than 4096 blocks (max_kv=%d) do not fit the LDS copy", la->max_kv);
    static FvkLdsConfigured configured;
    if (int rc = fvk_config_lds(configured, (const void*)attn_w64_kernel<true, 0, false, false, true>, LDS_LIST, "fvk_attn_tile_lists_bf16")) return rc;
    const long nblk = (long)la->n_lists * la->q_sub * a->H * a->B;
    hipLaunchKernelGGL((attn_w64_kernel<true, 0, false, false, true>), dim3((unsigned)nblk), dim3(256), LDS_LIST, s, *a, 1, (float*)nullptr, (float*)nullptr, *la);
    FVK_LAUNCH_CHECK();
    return FVK_OK;
}
#endif
