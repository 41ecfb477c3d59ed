// Block-sparse attention forward for 64-row query blocks over lists of 64-key KV blocks (the VSA sparse branch), head_dim 128, gfx950 — round 6.
// replaces: fastvideo-kernel/python/fastvideo_kernel/triton_kernels/block_sparse_attn_triton.py:34-158 (_attn_fwd_sparse), csrc/attention/
// block_sparse_h100.cu; called through fvk_attn_block_sparse_bf16 (q_block 64).
//
// attn_w16's design (one wave per SIMD with the whole register file, v_mfma_f32_16x16x32_bf16 in inline asm, S^T = K·Q^T so that P^T is the
// bf16-packed S^T registers of the same lane, the step software-pipelined INSIDE the wave: MFMA stream P·V(t-1) then Q·K^T(t+1), softmax(t) in
// the gaps, fixed softmax reference with an exact per-row recompute, row sums on the matrix pipe) re-tiled for the reference's 64 x 64 blocks:
//
//   * ONE WAVE = ONE 64-ROW QUERY BLOCK with its own KV list.  The four waves of a workgroup are four consecutive query blocks of one head and
//     share NOTHING: every wave stages its own K / V^T through its own 40 KiB of LDS with its own LDS-DMA and retires it with its own counted
//     vmcnt — there is no barrier in this kernel.  (The round-1 kernel, attn_fwd.hip: 32 query rows per wave, a SERIAL Q·K^T -> softmax -> P·V
//     chain per step, one loader wave beside every compute wave, one workgroup barrier per tile: matrix pipe 35 % busy, 7.2 VALU instructions
//     per score; profiles/r05a_pmc_vsa_block_sparse.json.)  Every LDS fragment feeds FOUR MFMAs (the wave's four 16-row q blocks).
//   * A block's operands travel as four 8-KiB GROUPS in the order the MFMA stream consumes them: Va / Vb = V^T d rows 0-63 / 64-127 (64 rows x
//     128 B), Ka / Kb = keys 0-31 / 32-63 (32 rows x 256 B) — full 128-B / 256-B global segments.  P·V walks d blocks outermost and Q·K^T walks
//     score tiles outermost, so a group is finished after a QUARTER of the iteration and its slot is refilled at once.  The wave's LDS is a ring
//     of FIVE slots: sequence number n = 4 * iteration + {Va, Vb, Ka, Kb} lives in slot n % 5, and the group refilled into a freed slot is n + 5,
//     i.e. every group is requested 1.25 iterations (~2 700 matrix cycles) before its first fragment read; 32 KiB per wave, 128 KiB per CU in
//     flight (the round-1 kernel: 70 KiB).  One 1-KiB piece is issued every fourth MFMA at fixed positions of the unrolled stream, so every
//     wait is `s_waitcnt vmcnt(N)` with a compile-time N (the pieces issued after the awaited group's last one).
//   * KV lists are read with SCALAR loads one pair of iterations ahead (no LDS left for them, and no list-length limit); a block's valid-key
//     count (variable_block_sizes) masks the bf16-packed P with a bitwise AND in front of the P·V that consumes it (27 % of the blocks of the
//     81f x 480p grid are ragged; the AND also removes whatever a padded K row produced, NaN included).
//   * XCD-contiguous workgroup ids (as attn_w16): one XCD walks neighbouring query blocks of one head, in step — L2 hit rate 63 -> 88 %,
//     fabric traffic 11.2 -> 3.5 GB per launch measured on the round-1 kernel with this deal alone (profiles/r06a_vsa_xcd_deal_pmc.json).
//
// Numerics: the reference kernels' (block_sparse_attn_triton.py:124-158): scale 1/sqrt(D) folded into exp2, masked columns arange(64) <
// block_size, P rounded to bf16 (RNE) before P·V, fp32 accumulation.  The softmax reference is the row maximum of the list's FIRST block
// instead of a running maximum (any reference gives the same quotient; a row whose sum leaves [0, 2^90) is redone exactly, per row).
#include "fvk_common.h"

namespace {

constexpr int SLOT = 8192;               // one staged group
constexpr int NSLOT = 5;
constexpr int WAVE_LDS = NSLOT * SLOT;   // 40 960
constexpr int LDS_BYTES = 4 * WAVE_LDS;  // 163 840: the whole CU
constexpr float L_LIMIT = 1.2379400392853803e27f;  // 2^90: a row sum at or above it (or NaN) triggers the exact recompute

typedef __attribute__((address_space(3))) void lds_void;
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float row4_max(float v) {  // over the four lanes {l, l^16, l^32, l^48} that share a query row (cold paths)
    v = fmaxf(v, __shfl_xor(v, 16));
    return fmaxf(v, __shfl_xor(v, 32));
}

template <int N>
__device__ __forceinline__ void wait_vm() {  // at most N of this wave's LDS-DMA pieces still in flight
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// wave-uniform dword read, synchronous (prologue and cold paths): p and idx must be wave-uniform
__device__ __forceinline__ int sload_sync(const int32_t* p, int idx) {
    int v;
    asm volatile("s_nop 4\n\ts_load_dword %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=&s"(v) : "s"(p), "s"(idx * 4) : "memory");
    return v;
}

// what one iteration needs beside the wave's registers (all wave-uniform)
struct IterArgs {
    int sVa, sVb, sKa, sKb, sSp;          // LDS byte offsets of the four groups consumed in this iteration and of the spare slot
    unsigned vb_t, kb_t2, vb_t1;          // source byte offsets: V^T columns of block t and t+1, K rows of block t+2
    int valid_prev;                       // valid keys of block t-1 (masks P(t-1))
};

// ABL (measurement build; results are WRONG, timing is what is measured): bit 0 = only every second LDS-DMA piece is issued (half the L2 -> LDS
// traffic), bit 1 = no softmax VALU in the loop, bit 2 = no counted vmcnt waits in the loop
template <int AUX, int ABL = 0>
struct BS16 {
    bf16x8 qf[4][4];     // Q fragments [q block][k-step of 32 d]   (accumulator file)
    f32x4 o[4][9];       // O^T accumulators [q block][16-row d block]; block 8 = the row sums (A operand all ones)
    f32x4 s[2][4][4];    // S^T [block parity][q block][tile: 2 * group + (a = 0 / b = 1)]
    bf16x8 pf[2][4][2];  // P^T, packed [block parity][q block][32-key group]
    float m_run[4];
    bf16x8 ones;
    int fkl[4], fvl[2];  // per-lane fragment offsets inside a group: K [k-step], V^T [32-key group]
    unsigned kv[4], vv[2];  // per-lane source offsets of a DMA piece: K [(p & 1) + 2 * (p >> 2)], V^T [p & 1]
    __amdgpu_buffer_rsrc_t k_rsrc, v_rsrc;
    unsigned k_pstride, v_pstride;  // bytes between pieces: 4 key rows, 8 d rows
    unsigned char* smem;            // this wave's ring
    float c2;
    int lp;                         // lane part of a score's key index: 16 (g >> 1) + 4 (g & 1)

    // piece p (0..7) of K group G (keys 32 G + 4 p .. + 4) of the block whose rows start kb bytes into the head's K slice -> slot
    // (the instruction's immediate offset moves the LDS destination AND the source address: pieces p & 3 of a half group share one M0 value — one
    // M0 write per four pieces instead of one per piece — and the scalar offset takes the 1024 * (p & 3) bytes back.  The immediate must be a
    // constant for the front end: a four-way switch that folds once the loops are unrolled)
#define FVK_BS_PIECE(RSRC, VOFF, SOFF)                                                                                              \
    {                                                                                                                               \
        lds_void* d_ = (lds_void*)(smem + slot + (p & 4) * 1024);                                                                   \
        const unsigned so_ = __builtin_amdgcn_readfirstlane((SOFF) - (unsigned)(p & 3) * 1024u);                                    \
        switch (p & 3) {                                                                                                            \
            case 0: __builtin_amdgcn_raw_ptr_buffer_load_lds(RSRC, d_, 16, VOFF, so_, 0, AUX); break;                               \
            case 1: __builtin_amdgcn_raw_ptr_buffer_load_lds(RSRC, d_, 16, VOFF, so_, 1024, AUX); break;                            \
            case 2: __builtin_amdgcn_raw_ptr_buffer_load_lds(RSRC, d_, 16, VOFF, so_, 2048, AUX); break;                            \
            default: __builtin_amdgcn_raw_ptr_buffer_load_lds(RSRC, d_, 16, VOFF, so_, 3072, AUX); break;                           \
        }                                                                                                                           \
    }
    __device__ __forceinline__ void issue_k(int G, int p, int slot, unsigned kb) const {
        FVK_BS_PIECE(k_rsrc, kv[(p & 1) + 2 * (p >> 2)], kb + (unsigned)(8 * G + p) * k_pstride)
    }
    // piece p (0..7) of V^T half hf (d rows 64 hf + 8 p .. + 8) of the block whose columns start vb bytes into a V^T row -> slot
    __device__ __forceinline__ void issue_v(int hf, int p, int slot, unsigned vb) const {
        FVK_BS_PIECE(v_rsrc, vv[p & 1], vb + (unsigned)(8 * hf + p) * v_pstride)
    }
    // V^T fragment: 32-key group G, d block db (0..7) — half db >> 2;  K fragment: score tile T = 2 * group + a/b, k-step ks
    __device__ __forceinline__ bf16x8 frag_v(int G, int db, int sVa, int sVb) const {
        return *reinterpret_cast<const bf16x8*>(smem + ((db >> 2) ? sVb : sVa) + fvl[G] + (db & 3) * 2048);
    }
    __device__ __forceinline__ bf16x8 frag_k(int T, int ks, int sKa, int sKb) const {
        return *reinterpret_cast<const bf16x8*>(smem + ((T >> 1) ? sKb : sKa) + fkl[ks] + (T & 1) * 2048);
    }
#define FVK_BS_PV4(FR, PAR, G, DB)                                                                                                        \
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %4, %5, %0\n\tv_mfma_f32_16x16x32_bf16 %1, %4, %6, %1\n\t"                                \
                 "v_mfma_f32_16x16x32_bf16 %2, %4, %7, %2\n\tv_mfma_f32_16x16x32_bf16 %3, %4, %8, %3"                                    \
                 : "+a"(o[0][DB]), "+a"(o[1][DB]), "+a"(o[2][DB]), "+a"(o[3][DB])                                                        \
                 : "v"(FR), "v"(pf[PAR][0][G]), "v"(pf[PAR][1][G]), "v"(pf[PAR][2][G]), "v"(pf[PAR][3][G]))
    // MFMA result -> compiler-generated reader and VALU / v_accvgpr_write -> MFMA operand, fenced by hand (asm statements get no hazard padding)
    template <int PAR>
    __device__ __forceinline__ void fence_s() {
        asm volatile("s_nop 15\n\ts_nop 3"
                     : "+v"(s[PAR][0][0]), "+v"(s[PAR][0][1]), "+v"(s[PAR][0][2]), "+v"(s[PAR][0][3]), "+v"(s[PAR][1][0]), "+v"(s[PAR][1][1]),
                       "+v"(s[PAR][1][2]), "+v"(s[PAR][1][3]), "+v"(s[PAR][2][0]), "+v"(s[PAR][2][1]), "+v"(s[PAR][2][2]), "+v"(s[PAR][2][3]));
        asm volatile("" : "+v"(s[PAR][3][0]), "+v"(s[PAR][3][1]), "+v"(s[PAR][3][2]), "+v"(s[PAR][3][3]));
    }
    template <int PAR>
    __device__ __forceinline__ void fence_p() {  // VALU-written P^T -> MFMA operand
        asm volatile("s_nop 7" : "+v"(pf[PAR][0][0]), "+v"(pf[PAR][0][1]), "+v"(pf[PAR][1][0]), "+v"(pf[PAR][1][1]), "+v"(pf[PAR][2][0]),
                                 "+v"(pf[PAR][2][1]), "+v"(pf[PAR][3][0]), "+v"(pf[PAR][3][1]));
    }
    __device__ __forceinline__ void fence_o() {
        asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");
#pragma unroll
        for (int qb = 0; qb < 4; ++qb)
            asm volatile("" : "+a"(o[qb][0]), "+a"(o[qb][1]), "+a"(o[qb][2]), "+a"(o[qb][3]), "+a"(o[qb][4]), "+a"(o[qb][5]), "+a"(o[qb][6]), "+a"(o[qb][7]), "+a"(o[qb][8]));
    }
    // P·V of one block from pf[PAR], plain form (tail, exact pass); block 8: the row sums (l = the sum of the bf16-rounded P)
    template <int PAR>
    __device__ __forceinline__ void pv_plain(int sVa, int sVb) {
#pragma unroll
        for (int db = 0; db < 9; ++db)
#pragma unroll
            for (int G = 0; G < 2; ++G) {
                const bf16x8 fr = db < 8 ? frag_v(G, db, sVa, sVb) : ones;
                FVK_BS_PV4(fr, PAR, G, db);
            }
    }
    // Q·K^T of one block -> s[PAR], plain form (prologue, exact pass)
    template <int PAR>
    __device__ __forceinline__ void qk_plain(int sKa, int sKb) {
#pragma unroll
        for (int T = 0; T < 4; ++T)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const bf16x8 fr = frag_k(T, ks, sKa, sKb);
                if (ks == 0)
                    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %4, %5, 0\n\tv_mfma_f32_16x16x32_bf16 %1, %4, %6, 0\n\t"
                                 "v_mfma_f32_16x16x32_bf16 %2, %4, %7, 0\n\tv_mfma_f32_16x16x32_bf16 %3, %4, %8, 0"
                                 : "=&v"(s[PAR][0][T]), "=&v"(s[PAR][1][T]), "=&v"(s[PAR][2][T]), "=&v"(s[PAR][3][T])
                                 : "v"(fr), "a"(qf[0][ks]), "a"(qf[1][ks]), "a"(qf[2][ks]), "a"(qf[3][ks]));
                else
                    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %4, %5, %0\n\tv_mfma_f32_16x16x32_bf16 %1, %4, %6, %1\n\t"
                                 "v_mfma_f32_16x16x32_bf16 %2, %4, %7, %2\n\tv_mfma_f32_16x16x32_bf16 %3, %4, %8, %3"
                                 : "+v"(s[PAR][0][T]), "+v"(s[PAR][1][T]), "+v"(s[PAR][2][T]), "+v"(s[PAR][3][T])
                                 : "v"(fr), "a"(qf[0][ks]), "a"(qf[1][ks]), "a"(qf[2][ks]), "a"(qf[3][ks]));
            }
        fence_s<PAR>();
    }
    // key (0..63, within the block) of score e of tile T in lane group g (fvk_v_transpose_bf16's key order, attn_w16.hip)
    static __device__ __forceinline__ int key_of(int T, int g, int e) { return 32 * (T >> 1) + 16 * (g >> 1) + 4 * (g & 1) + 8 * (T & 1) + e; }
    template <int PAR>
    __device__ __forceinline__ void mask_keys(int valid, int g) {  // scores of keys >= valid -> -inf (cold paths)
#pragma unroll
        for (int qb = 0; qb < 4; ++qb)
#pragma unroll
            for (int T = 0; T < 4; ++T)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (key_of(T, g, e) >= valid) s[PAR][qb][T][e] = -INFINITY;
    }
    // P of keys >= valid -> +0, on the packed registers: slot x of group G holds key 32 G + 8 (x >> 2) + (x & 3) + lp
    template <int PAR>
    __device__ __forceinline__ void mask_p(int valid) {
#pragma unroll
        for (int G = 0; G < 2; ++G)
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const int key0 = 32 * G + 8 * (w >> 1) + 2 * (w & 1) + lp;  // slots 2 w, 2 w + 1
                const unsigned mk = (key0 < valid ? 0xffffu : 0u) | (key0 + 1 < valid ? 0xffff0000u : 0u);
#pragma unroll
                for (int qb = 0; qb < 4; ++qb) {
                    u32x4_t u = __builtin_bit_cast(u32x4_t, pf[PAR][qb][G]);
                    u[w] &= mk;
                    pf[PAR][qb][G] = __builtin_bit_cast(bf16x8, u);
                }
            }
        fence_p<PAR>();
    }
    template <int PAR, int QB>
    __device__ __forceinline__ float row_max() const {
        float mx = s[PAR][QB][0][0];
#pragma unroll
        for (int T = 0; T < 4; ++T)
#pragma unroll
            for (int e = 0; e < 4; ++e) mx = fmaxf(mx, s[PAR][QB][T][e]);
        return row4_max(mx);
    }
    template <int PAR, int QB>
    __device__ __forceinline__ void exp_pack(float mc) {
#pragma unroll
        for (int G = 0; G < 2; ++G)
#pragma unroll
            for (int x = 0; x < 8; ++x)
                pf[PAR][QB][G][x] = (bf16_t)__builtin_amdgcn_exp2f(__builtin_fmaf(s[PAR][QB][2 * G + (x >> 2)][x & 3], c2, -mc));
    }
    template <int PAR>
    __device__ __forceinline__ void exp_pack_all() {
        exp_pack<PAR, 0>(m_run[0] * c2);
        exp_pack<PAR, 1>(m_run[1] * c2);
        exp_pack<PAR, 2>(m_run[2] * c2);
        exp_pack<PAR, 3>(m_run[3] * c2);
        fence_p<PAR>();
    }
    // exact online-softmax step (new running max first; O and l rescaled): the slow path
    template <int PAR>
    __device__ __forceinline__ void softmax_exact() {
        const float mx[4] = {row_max<PAR, 0>(), row_max<PAR, 1>(), row_max<PAR, 2>(), row_max<PAR, 3>()};
        fence_o();
#pragma unroll
        for (int qb = 0; qb < 4; ++qb) {
            const float m_new = fmaxf(m_run[qb], mx[qb]);
            const float alpha = __builtin_amdgcn_exp2f((m_run[qb] - m_new) * c2);
#pragma unroll
            for (int d = 0; d < 9; ++d)
#pragma unroll
                for (int r = 0; r < 4; ++r) o[qb][d][r] *= alpha;
            m_run[qb] = m_new;
        }
        exp_pack_all<PAR>();
        fence_o();
    }
    // The exact pass (rare): plain online softmax over the whole list, one block at a time through slots 0-3 (load, wait, compute).
    __device__ __forceinline__ void exact_pass(const int32_t* list, const int32_t* sizes, int n_real, int nkv, unsigned kblk, int g) {
#pragma unroll
        for (int qb = 0; qb < 4; ++qb) {
            m_run[qb] = -1e30f;
#pragma unroll
            for (int d = 0; d < 9; ++d)
#pragma unroll
                for (int r = 0; r < 4; ++r) o[qb][d][r] = 0.f;
        }
        fence_o();
        for (int st = 0; st < n_real; ++st) {
            int id = sload_sync(list, st);
            id = id < 0 ? 0 : (id < nkv ? id : nkv - 1);
            const int valid = sload_sync(sizes, id);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                issue_k(0, p, 0, (unsigned)id * kblk);
                issue_k(1, p, SLOT, (unsigned)id * kblk);
                issue_v(0, p, 2 * SLOT, (unsigned)id * 128u);
                issue_v(1, p, 3 * SLOT, (unsigned)id * 128u);
            }
            wait_vm<0>();
            __builtin_amdgcn_sched_barrier(0);
            if (valid > 0) {  // wave-uniform
                qk_plain<0>(0, SLOT);
                if (valid < 64) mask_keys<0>(valid, g);
                softmax_exact<0>();
                pv_plain<0>(2 * SLOT, 3 * SLOT);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this block's fragment reads are done before the next block's pieces land
            __builtin_amdgcn_sched_barrier(0);
        }
        fence_o();
    }

    // ---- iteration t: 136 chunks of { 1 MFMA | a fragment read, a DMA piece or half the softmax of one score }, pinned with sched_barrier.
    // MFMA stream: P·V of block t-1 from pf[EVEN] (d blocks outermost: 64 MFMAs, then the two row-sum slots), then Q·K^T of block t+1 -> s[EVEN]
    // (score tiles outermost: 64 MFMAs).  The 32 LDS fragments (read FD ahead) each feed four consecutive chunks (q blocks 0..3):
    //   L 0-7 = Va, 8-15 = Vb, 16-23 = Ka, 24-31 = Kb.   VALU stream: block t = s[1-EVEN] -> pf[1-EVEN] against the fixed reference, as attn_w16.
    // DMA pieces (one per four chunks, at chunks 1, 5, ..., 133): Va(t) pieces 1-7 -> spare slot | Vb(t) -> Va's slot (free from chunk 29) |
    // Ka(t+2) -> Vb's slot (from 61) | Kb(t+2) -> Ka's slot (from 101) | Va(t+1) piece 0 -> Kb's slot (chunk 133).  32 pieces per iteration.
    // Waits = the pieces issued after the awaited group's last one: Va(t-1) 25 (iteration top), Vb(t-1) 20 (before fragment 8, chunk 11),
    // Ka(t+1) 20 (before fragment 16, chunk 43), Kb(t+1) 22 (before fragment 24, chunk 83).
    template <int EVEN>
    __device__ __forceinline__ void iter(const IterArgs& c) {
        constexpr int CUR = 1 - EVEN, FD = 6;
        if (c.valid_prev < 64) mask_p<EVEN>(c.valid_prev);  // wave-uniform
        const float mc[4] = {m_run[0] * c2, m_run[1] * c2, m_run[2] * c2, m_run[3] * c2};
        auto load_frag = [&](int L) {
            return L < 16 ? frag_v(L & 1, L >> 1, c.sVa, c.sVb) : frag_k((L - 16) >> 2, (L - 16) & 3, c.sKa, c.sKb);
        };
        if (!(ABL & 4)) wait_vm<25>();
        __builtin_amdgcn_sched_barrier(0);
        bf16x8 fr[FD];
#pragma unroll
        for (int i = 0; i < FD; ++i) fr[i] = load_frag(i);
        float p_nx = __builtin_amdgcn_exp2f(__builtin_fmaf(s[CUR][0][0][0], c2, -mc[0]));
        float e_nx = __builtin_fmaf(s[CUR][0][0][1], c2, -mc[0]);
#pragma unroll
        for (int m = 0; m < 136; ++m) {
            const int F = m >> 2, qb = m & 3;
            const bool is_ones = F == 16 || F == 17;
            const int L = F < 16 ? F : F - 2;  // LDS fragment of this slot (ones slots: none)
            if (F < 16) {
                const int db = L >> 1, G = L & 1;
                asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(o[qb][db]) : "v"(fr[L % FD]), "v"(pf[EVEN][qb][G]));
            } else if (is_ones) {
                const int G = F - 16;
                asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(o[qb][8]) : "v"(ones), "v"(pf[EVEN][qb][G]));
            } else {
                const int T = (L - 16) >> 2, ks = (L - 16) & 3;
                if (ks == 0) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=&v"(s[EVEN][qb][T]) : "v"(fr[L % FD]), "a"(qf[qb][ks]));
                else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(s[EVEN][qb][T]) : "v"(fr[L % FD]), "a"(qf[qb][ks]));
            }
            __builtin_amdgcn_sched_barrier(0);  // the MFMA FIRST: everything below runs in its shadow
            if (qb == 3 && !is_ones && L + FD < 32) {
                if (!(ABL & 4) && L + FD == 8) wait_vm<20>();
                if (!(ABL & 4) && L + FD == 16) wait_vm<20>();
                if (!(ABL & 4) && L + FD == 24) wait_vm<22>();
                fr[L % FD] = load_frag(L + FD);
            }
            if ((m & 3) == 1 && !((ABL & 1) && ((m >> 2) & 1))) {
                const int k = m >> 2;
                if (k < 7) issue_v(0, k + 1, c.sSp, c.vb_t);
                else if (k < 15) issue_v(1, k - 7, c.sVa, c.vb_t);
                else if (k < 23) issue_k(0, k - 15, c.sVb, c.kb_t2);
                else if (k >= 25 && k < 33) issue_k(1, k - 25, c.sKa, c.kb_t2);
                else if (k == 33) issue_v(0, 0, c.sKb, c.vb_t1);
            }
            if (m < 128 && !(ABL & 2)) {
                const int cc = m >> 1;  // the score this chunk pair works on
                if ((m & 1) == 0) {
                    const int q_ = cc >> 4, G = (cc >> 3) & 1, x = cc & 7;
                    pf[CUR][q_][G][x] = (bf16_t)p_nx;
                    if (x & 1) asm volatile("" : "+v"(pf[CUR][q_][G]));  // the pair's v_cvt_pk stays in this chunk
                } else {
                    if (cc + 1 < 64) p_nx = __builtin_amdgcn_exp2f(e_nx);
                    if (cc + 2 < 64) {
                        const int y = cc + 2, q_ = y >> 4, G = (y >> 3) & 1, x = y & 7;
                        e_nx = __builtin_fmaf(s[CUR][q_][2 * G + (x >> 2)][x & 3], c2, -mc[q_]);
                    }
                    asm volatile("" : "+v"(p_nx), "+v"(e_nx));  // keep the chunk's work IN the chunk
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
};

// slot s of the ring as an LDS byte offset, for head h in 0..4 and k in 0..4
__device__ __forceinline__ int slot_of(int h, int k) {
    const int t = h + k;
    return (t >= NSLOT ? t - NSLOT : t) * SLOT;
}

// SPLIT (the launch's last, partial round of workgroups): logical workgroup wg_base + x / parts walks PART x % parts of each of its four lists
// (an even number of blocks per part, the last part the rest) and writes the part's result UN-merged — normalised O as fp32 rows + base-2 LSE,
// the split-KV form of attn_w16 — for bs16_merge_parts_kernel below.  Row r of part p lives at (p * tail_rows + r), r = 256 * (workgroup -
// wg_base) + 64 * wave + row.
struct Bs16Split {
    int wg_base, parts, tail_rows;
    float* o_part;    // [parts][tail_rows][128]
    float* lse_part;  // [parts][tail_rows]
};

template <bool PLAIN_IDS = false, int AUX = 0, int ABL = 0, bool SPLIT = false>
__global__ __launch_bounds__(256, 1) void attn_bs16_kernel(fvk_attn_args a, const int32_t* __restrict__ q2k_idx, const int32_t* __restrict__ q2k_num,
                                                           const int32_t* __restrict__ kv_block_sizes, int max_kv, Bs16Split sp) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_all[];
    FVK_CLAIM_WHOLE_REGISTER_FILE();
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g = lane >> 4;
    const int nqb = a.Sq >> 6;          // query blocks (= lists) per head
    const int nwg = (nqb + 3) >> 2;     // workgroups per head: four consecutive query blocks each
    // XCD-aware deal: hardware workgroup id x lands on XCD x % 8 (its own L2); XCD c gets the CONTIGUOUS logical ids [c*q + min(c, r), ...)
    const int nblk = gridDim.x, xq = nblk >> 3, xr = nblk & 7, xcd = blockIdx.x & 7;
    const int lid = PLAIN_IDS ? (int)blockIdx.x : xcd * xq + (xcd < xr ? xcd : xr) + (int)(blockIdx.x >> 3);
    const int bid = SPLIT ? sp.wg_base + lid / sp.parts : lid;
    const int part = SPLIT ? lid % sp.parts : 0;
    const int qb_id = (bid % nwg) * 4 + wave;
    if (qb_id >= nqb) return;  // wave-uniform; no barrier anywhere in this kernel
    const int h = (bid / nwg) % a.H;
    const int b = bid / (nwg * a.H);
    const long meta = ((long)b * a.H + h) * nqb + qb_id;
    const int32_t* list = q2k_idx + meta * max_kv;
    const int nkv = a.Skv >> 6;
    int n_real = __builtin_amdgcn_readfirstlane(q2k_num[meta]);
    n_real = n_real < 0 ? 0 : (n_real < max_kv ? n_real : max_kv);
    if (SPLIT) {  // this wave's part of the list: blocks [part * per, part * per + per) with per even (iterations come in pairs)
        const int per = ((n_real + sp.parts - 1) / sp.parts + 1) & ~1;
        const int rest = n_real - part * per;
        list += part * per;
        n_real = rest < 0 ? 0 : (rest < per ? rest : per);
    }

    const bf16_t* qp = (const bf16_t*)a.q + (long)b * a.q_bs + (long)h * a.q_hs;
    const bf16_t* kp = (const bf16_t*)a.k + (long)b * a.k_bs + (long)h * a.k_hs;
    const bf16_t* vtp = (const bf16_t*)a.vt + ((long)b * a.H + h) * 128L * a.Skv_pad;
    bf16_t* op = (bf16_t*)a.o + (long)b * a.o_bs + (long)h * a.o_hs;

    BS16<AUX, ABL> w;
    w.smem = smem_all + wave * WAVE_LDS;
    w.c2 = a.scale * 1.4426950408889634f;
    w.lp = 16 * (g >> 1) + 4 * (g & 1);
    // Q fragments of the wave's four 16-row blocks (B operand of S^T = K·Q^T): row q0 + 16*blk + l15, d = 32*ks + 8*g .. +8, loaded STRAIGHT
    // into the accumulator file (invisible to the compiler's vmcnt bookkeeping: the prologue's s_waitcnt vmcnt(0) covers them)
    const int q0 = qb_id * 64;
    int qrow[4];
#pragma unroll
    for (int qb = 0; qb < 4; ++qb) {
        qrow[qb] = q0 + 16 * qb + l15;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
            asm volatile("global_load_dwordx4 %0, %1, off" : "=a"(w.qf[qb][ks]) : "v"(qp + (long)qrow[qb] * a.q_ss + ks * 32 + g * 8) : "memory");
    }
    // LDS images (the hardware writes a piece lane-linearly, so each lane fetches the SOURCE chunk that belongs at its linear position):
    //   K group: key row r (0..31) at r * 256, 16-B chunk c at (c ^ ((r & 7) | ((r >> 1) & 8))) << 4   (attn_w16's image: the 16 rows {0-7, 16-23} /
    //            {8-15, 24-31} of a tile's fragment read land on 16 different chunk positions);  piece p = rows 4 p .. 4 p + 3, lane -> row 4 p + (lane >> 4)
    //   V^T group: d row r (0..63) at r * 128, chunk c (8 keys) at (c ^ ((r >> 1) & 7)) << 4;          piece p = rows 8 p .. 8 p + 7, lane -> row 8 p + (lane >> 3)
    w.k_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)kp, 0, (unsigned)((((long)a.Skv - 1) * a.k_ss + 128) * 2), 0x00020000);
    w.v_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)vtp, 0, (unsigned)(256L * a.Skv_pad), 0x00020000);
    {
        const int j = lane >> 4, c = lane & 15;
#pragma unroll
        for (int i = 0; i < 4; ++i) {  // i = (p & 1) + 2 * (p >> 2): row 4 p + j has (r & 7) = 4 (p & 1) + j and bit 4 = p >> 2
            const int swz = (4 * (i & 1) + j) | (8 * (i >> 1));
            w.kv[i] = (unsigned)((long)j * a.k_ss * 2) + (unsigned)((c ^ swz) << 4);
        }
        const int jr = lane >> 3, cv = lane & 7;
#pragma unroll
        for (int i = 0; i < 2; ++i)  // i = p & 1: row 8 p + jr has (r >> 1) & 7 = (4 (p & 1) + (jr >> 1)) & 7
            w.vv[i] = (unsigned)((long)jr * a.Skv_pad * 2) + (unsigned)((cv ^ ((4 * i + (jr >> 1)) & 7)) << 4);
    }
    w.k_pstride = (unsigned)(4 * a.k_ss * 2);
    w.v_pstride = (unsigned)(8 * a.Skv_pad * 2);
    const unsigned kblk = (unsigned)(64 * a.k_ss * 2);  // bytes between the K rows of consecutive blocks
    // fragment offsets.  K: lane row l15 of tile a is key row (l15 < 8 ? l15 : l15 + 8) of the 32-key group (+8 for tile b: + 2048 B); its swizzle
    // term is l15 itself.  V^T: row l15 of the d block (d blocks 2048 B apart), swizzle term (l15 >> 1) & 7.
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) w.fkl[ks] = (l15 < 8 ? l15 : l15 + 8) * 256 + (((4 * ks + g) ^ l15) << 4);
#pragma unroll
    for (int G = 0; G < 2; ++G) w.fvl[G] = l15 * 128 + (((4 * G + g) ^ ((l15 >> 1) & 7)) << 4);
#pragma unroll
    for (int qb = 0; qb < 4; ++qb) {
        w.m_run[qb] = -1e30f;
#pragma unroll
        for (int d = 0; d < 9; ++d)
#pragma unroll
            for (int r = 0; r < 4; ++r) w.o[qb][d][r] = 0.f;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) w.ones[e] = (bf16_t)1.0f;
    asm volatile("" : "+v"(w.ones));  // opaque: a known constant would be re-materialised right in front of the asm MFMA that reads it
    w.fence_o();

    if (n_real > 0) {  // wave-uniform (an empty list: zeros, as the round-1 kernel)
        const int n = (n_real + 1) & ~1;  // an odd list gets one more block with no valid key: iterations come in pairs
        auto entry = [&](int i) {         // block id of list position i (positions past the end re-read the last block: harmless prefetch)
            const int id = sload_sync(list, i < n_real ? i : n_real - 1);
            return id < 0 ? 0 : (id < nkv ? id : nkv - 1);
        };
        int id0 = entry(0), id1 = entry(1), idA = id1, idB = entry(2), idC = entry(3), idD = entry(4);
        int valA = sload_sync(kv_block_sizes, id0);
        int valB = 1 < n_real ? sload_sync(kv_block_sizes, id1) : 0;
        // ---- prologue: K(0) -> slots 0, 1; K(1) -> slots 2, 3; Q·K^T(0), the reference + softmax(0), Q·K^T(1) ---------------------------------
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            w.issue_k(0, p, 0, (unsigned)id0 * kblk);
            w.issue_k(1, p, SLOT, (unsigned)id0 * kblk);
            w.issue_k(0, p, 2 * SLOT, (unsigned)id1 * kblk);
            w.issue_k(1, p, 3 * SLOT, (unsigned)id1 * kblk);
        }
        wait_vm<0>();
        __builtin_amdgcn_sched_barrier(0);
        w.template qk_plain<0>(0, SLOT);
        if (valA < 64) w.template mask_keys<0>(valA, g);
        // the fixed reference: exact row max of the first block (it has at least one valid key)
        w.m_run[0] = w.template row_max<0, 0>();
        w.m_run[1] = w.template row_max<0, 1>();
        w.m_run[2] = w.template row_max<0, 2>();
        w.m_run[3] = w.template row_max<0, 3>();
        w.template exp_pack_all<0>();
        w.template qk_plain<1>(2 * SLOT, 3 * SLOT);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // every fragment read of slots 0-3 has returned: the slots may be overwritten
        __builtin_amdgcn_sched_barrier(0);
        // the ring as iteration 1 (head 0) expects it, in the steady state's ISSUE ORDER (the counted waits count on it):
        // Va(0) -> 0, Vb(0) -> 1, Ka(2) -> 2, Kb(2) -> 3, Va(1) piece 0 -> 4
#pragma unroll
        for (int p = 0; p < 8; ++p) w.issue_v(0, p, 0, (unsigned)id0 * 128u);
#pragma unroll
        for (int p = 0; p < 8; ++p) w.issue_v(1, p, SLOT, (unsigned)id0 * 128u);
#pragma unroll
        for (int p = 0; p < 8; ++p) w.issue_k(0, p, 2 * SLOT, (unsigned)idB * kblk);
#pragma unroll
        for (int p = 0; p < 8; ++p) w.issue_k(1, p, 3 * SLOT, (unsigned)idB * kblk);
        w.issue_v(0, 0, 4 * SLOT, (unsigned)idA * 128u);
        __builtin_amdgcn_sched_barrier(0);
        // ---- pairs j = 0 .. n/2 - 2: iterations t = 2j+1 (P·V(2j), softmax(2j+1), Q·K^T(2j+2)) and t = 2j+2 -------------------------------------
        // scalar state: idA..idD = block ids of list positions 2j+1 .. 2j+4, valA / valB = valid keys of positions 2j / 2j+1
        int hd = 0;  // ring head: slot of Va(t-1)
        for (int j = 0; 2 * j + 2 < n; ++j) {
            // next pair's list entries and valid-key counts: scalar loads issued here, retired at the end of the pair (straight-line code between)
            int idE, idF, valC, valD;
            {
                const int pE = 2 * j + 5 < n_real ? 2 * j + 5 : n_real - 1, pF = 2 * j + 6 < n_real ? 2 * j + 6 : n_real - 1;
                asm volatile("s_nop 4\n\ts_load_dword %0, %4, %5\n\ts_load_dword %1, %4, %6\n\ts_load_dword %2, %7, %8\n\ts_load_dword %3, %7, %9"
                             : "=&s"(idE), "=&s"(idF), "=&s"(valC), "=&s"(valD)
                             : "s"(list), "s"(pE * 4), "s"(pF * 4), "s"(kv_block_sizes), "s"(idB * 4), "s"(idC * 4)
                             : "memory");
            }
            IterArgs c;
            c.sVa = slot_of(hd, 0); c.sVb = slot_of(hd, 1); c.sKa = slot_of(hd, 2); c.sKb = slot_of(hd, 3); c.sSp = slot_of(hd, 4);
            c.vb_t = (unsigned)idA * 128u; c.kb_t2 = (unsigned)idC * kblk; c.vb_t1 = (unsigned)idB * 128u; c.valid_prev = valA;
            w.template iter<0>(c);
            hd = hd == 0 ? 4 : hd - 1;  // + 4 mod 5
            c.sVa = slot_of(hd, 0); c.sVb = slot_of(hd, 1); c.sKa = slot_of(hd, 2); c.sKb = slot_of(hd, 3); c.sSp = slot_of(hd, 4);
            c.vb_t = (unsigned)idB * 128u; c.kb_t2 = (unsigned)idD * kblk; c.vb_t1 = (unsigned)idC * 128u; c.valid_prev = valB;
            w.template iter<1>(c);
            hd = hd == 0 ? 4 : hd - 1;
            asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(idE), "+s"(idF), "+s"(valC), "+s"(valD)::"memory");
            idE = idE < 0 ? 0 : (idE < nkv ? idE : nkv - 1);
            idF = idF < 0 ? 0 : (idF < nkv ? idF : nkv - 1);
            valA = 2 * j + 2 < n_real ? valC : 0;
            valB = 2 * j + 3 < n_real ? valD : 0;
            idA = idC; idB = idD; idC = idE; idD = idF;
        }
        // ---- tail: P·V(n-2), softmax(n-1), P·V(n-1).  Ring (head hd): V(n-2) in slots hd, hd+1; Va(n-1) piece 0 in hd+4; hd+2, hd+3 hold the
        // harmless K prefetch past the end.  idA = block n-1, valA / valB = valid keys of blocks n-2 / n-1.
        wait_vm<0>();
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int p = 1; p < 8; ++p) w.issue_v(0, p, slot_of(hd, 4), (unsigned)idA * 128u);
#pragma unroll
        for (int p = 0; p < 8; ++p) w.issue_v(1, p, slot_of(hd, 2), (unsigned)idA * 128u);
        __builtin_amdgcn_sched_barrier(0);
        if (valA < 64) w.template mask_p<0>(valA);
        w.template pv_plain<0>(slot_of(hd, 0), slot_of(hd, 1));
        w.template fence_s<1>();
        w.template exp_pack_all<1>();
        if (valB < 64) w.template mask_p<1>(valB);
        wait_vm<0>();
        __builtin_amdgcn_sched_barrier(0);
        w.template pv_plain<1>(slot_of(hd, 4), slot_of(hd, 2));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    w.fence_o();  // the last MFMAs' results before the epilogue's v_accvgpr_read
    // ---- epilogue: normalise and store; a row whose fixed-reference sum left the safe range (NaN, infinite or >= 2^90) is redone by the exact
    // pass and stored again — per ROW.  Lane (l15, g) holds d = 16*db + 4*g + {0..3} of its query row.
    bool redo[4] = {false, false, false, false};
    const long prow0 = SPLIT ? (long)part * sp.tail_rows + ((long)(bid - sp.wg_base) * 4 + wave) * 64 : 0;  // this wave's first row of its part
#define FVK_BS_STORE_ROWS(ONLY_REDO)                                                                                 \
    _Pragma("unroll") for (int qb = 0; qb < 4; ++qb) {                                                               \
        const float l_tot = w.o[qb][8][0]; /* every row of block 8 holds the whole row sum */                        \
        const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;                                                          \
        if (!(ONLY_REDO)) redo[qb] = n_real > 0 && !(l_tot < L_LIMIT);                                               \
        if (!(ONLY_REDO) || redo[qb]) {                                                                              \
            if (SPLIT) {                                                                                             \
                float* prow = sp.o_part + (prow0 + 16 * qb + l15) * 128;                                             \
                _Pragma("unroll") for (int d = 0; d < 8; ++d) {                                                      \
                    f32x4 v4;                                                                                        \
                    _Pragma("unroll") for (int e = 0; e < 4; ++e) v4[e] = w.o[qb][d][e] * inv;                       \
                    *reinterpret_cast<f32x4*>(prow + d * 16 + g * 4) = v4;                                           \
                }                                                                                                    \
                /* an empty part (or one whose keys are all masked): weight 0 in the merge */                        \
                if (g == 0) sp.lse_part[prow0 + 16 * qb + l15] = l_tot > 0.f ? w.m_run[qb] * w.c2 + log2f(l_tot) : -INFINITY; \
            } else {                                                                                                 \
                bf16_t* orow = op + (long)qrow[qb] * a.o_ss;                                                         \
                _Pragma("unroll") for (int d = 0; d < 8; ++d) {                                                      \
                    bf16x4 v4;                                                                                       \
                    _Pragma("unroll") for (int e = 0; e < 4; ++e) v4[e] = (bf16_t)(w.o[qb][d][e] * inv);             \
                    *reinterpret_cast<bf16x4*>(orow + d * 16 + g * 4) = v4;                                          \
                }                                                                                                    \
                if (a.lse && g == 0) a.lse[((long)b * a.H + h) * a.Sq + qrow[qb]] = w.m_run[qb] * w.c2 + log2f(l_tot); \
            }                                                                                                        \
        }                                                                                                            \
    }
    FVK_BS_STORE_ROWS(false)
    if (ABL == 0 && __any(redo[0] || redo[1] || redo[2] || redo[3])) {  // this wave only: the waves of a workgroup share nothing
        w.exact_pass(list, kv_block_sizes, n_real, nkv, kblk, g);
        FVK_BS_STORE_ROWS(true)
    }
#undef FVK_BS_STORE_ROWS
#endif  // __HIP_DEVICE_COMPILE__
}

// out[b, row, h, :] = sum_p 2^(lse_p - max) o_part[p] / sum_p 2^(lse_p - max) over the parts of a SPLIT launch: one wave per row, two columns per
// lane (attn_w16's merge of its split-KV form); HBM-bound, parts * 516 B read and 256 B written per row.
__global__ __launch_bounds__(256) void bs16_merge_parts_kernel(fvk_attn_args a, Bs16Split sp) {
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= sp.tail_rows) return;
    const int lane = threadIdx.x & 63;
    const int nqb = a.Sq >> 6, nwg = (nqb + 3) >> 2;
    const int bid = sp.wg_base + (int)(r >> 8);
    const int qb_id = (bid % nwg) * 4 + (int)((r >> 6) & 3);
    if (qb_id >= nqb) return;  // (a head's last workgroup may hold fewer than four lists)
    const int h = (bid / nwg) % a.H, b = bid / (nwg * a.H);
    const int qrow = qb_id * 64 + (int)(r & 63);
    float mx = -INFINITY;
    for (int p = 0; p < sp.parts; ++p) mx = fmaxf(mx, sp.lse_part[(long)p * sp.tail_rows + r]);
    float acc0 = 0.f, acc1 = 0.f, wsum = 0.f;
    for (int p = 0; p < sp.parts; ++p) {
        const float wgt = exp2f(sp.lse_part[(long)p * sp.tail_rows + r] - mx);  // an empty part: 2^(-inf) = 0 (all empty: NaN, skipped)
        if (wgt > 0.f) {
            const float2 v = *reinterpret_cast<const float2*>(sp.o_part + ((long)p * sp.tail_rows + r) * 128 + lane * 2);
            acc0 += wgt * v.x;
            acc1 += wgt * v.y;
            wsum += wgt;
        }
    }
    const float inv = wsum > 0.f ? 1.0f / wsum : 0.f;
    bf16x2 o2;
    o2[0] = (bf16_t)(acc0 * inv);
    o2[1] = (bf16_t)(acc1 * inv);
    *reinterpret_cast<bf16x2*>((bf16_t*)a.o + (long)b * a.o_bs + (long)h * a.o_hs + (long)qrow * a.o_ss + lane * 2) = o2;
    if (a.lse && lane == 0) a.lse[((long)b * a.H + h) * a.Sq + qrow] = mx + log2f(wsum);
}

// The launch plan of the 64-row list kernel: `full` workgroups (whole rounds of one workgroup per CU) run whole lists; the `tail` workgroups of
// the last, partial round are cut into `parts` list parts each so that they fill the chip too (cfg2: 1 872 workgroups = 7.31 rounds of 256 -> 7
// rounds + 80 workgroups x 3 parts, i.e. 7.4 rounds instead of 8).  parts = 1: no split.
struct Bs16Plan { long full, tail; int parts; long ws_bytes; };
Bs16Plan bs16_plan(const fvk_attn_args* a, int max_kv) {
    static int cus = 0;
    if (cus == 0) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) cus = n;
        else cus = 256;
    }
    const long nblk = (((long)(a->Sq / 64) + 3) / 4) * a->H * a->B;
    Bs16Plan p{nblk, 0, 1, 0};
    const long tail = nblk % cus;
    int parts = tail > 0 ? (int)(cus / tail) : 1;
    if (parts > 4) parts = 4;
    if (parts > max_kv / 16) parts = max_kv / 16;  // a part shorter than ~16 blocks is mostly prologue
    if (parts >= 2) {
        p.full = nblk - tail; p.tail = tail; p.parts = parts;
        p.ws_bytes = (long)parts * tail * 256 * (128 + 1) * 4;
    }
    return p;
}

}  // namespace

// called by fvk_attn_block_sparse_ws_bf16 (attn_fwd.hip) after its argument checks; variant 1 (measurement build) = hardware workgroup order.
// ws / ws_bytes: optional device workspace for the split last round (fvk_attn_bs16_workspace_bytes); null or too small = every list whole.
long fvk_attn_bs16_workspace_bytes(const fvk_attn_args* a, int max_kv) { return bs16_plan(a, max_kv).ws_bytes; }

int fvk_attn_bs16_launch(const fvk_attn_args* a, const int32_t* q2k_idx, const int32_t* q2k_num, const int32_t* kv_block_sizes, int max_kv,
                         int variant, void* ws, long ws_bytes, hipStream_t s) {
    const long nwg = ((long)(a->Sq / 64) + 3) / 4;
    const long nblk = nwg * a->H * a->B;
    FVK_CHECK(nblk < 0x7fffffffL / 4, FVK_ERR_ARG, "fvk_attn_block_sparse_bf16: grid too large");
#if FVK_VARIANTS
#define FVK_BS16_VARIANT(N, ...)                                                                                                          \
    if (variant == N) {                                                                                                                   \
        static FvkLdsConfigured configured_v;                                                                                             \
        if (int rc = fvk_config_lds(configured_v, (const void*)attn_bs16_kernel<__VA_ARGS__>, LDS_BYTES, "fvk_attn_block_sparse_bf16 (bs16)")) return rc; \
        hipLaunchKernelGGL((attn_bs16_kernel<__VA_ARGS__>), dim3((unsigned)nblk), dim3(256), LDS_BYTES, s, *a, q2k_idx, q2k_num, kv_block_sizes, max_kv, Bs16Split{}); \
        FVK_LAUNCH_CHECK();                                                                                                               \
        return FVK_OK;                                                                                                                    \
    }
    FVK_BS16_VARIANT(1, true, 0)    // hardware workgroup order
    FVK_BS16_VARIANT(2, false, 2)   // LDS-DMA pieces with the nt policy (aux = 2)
    FVK_BS16_VARIANT(3, false, 1)   // ... with sc0 (aux = 1)
    FVK_BS16_VARIANT(4, false, 0)   // every list whole (no split last round)
    FVK_BS16_VARIANT(11, false, 0, 1)  // timing ablations (wrong results): half the LDS-DMA pieces
    FVK_BS16_VARIANT(12, false, 0, 2)  // no softmax VALU
    FVK_BS16_VARIANT(14, false, 0, 4)  // no counted waits
    FVK_BS16_VARIANT(13, false, 0, 3)  // half the pieces, no softmax
    FVK_BS16_VARIANT(17, false, 0, 7)  // MFMAs + fragment reads + half the pieces
#undef FVK_BS16_VARIANT
#endif
    (void)variant;
    Bs16Plan plan = bs16_plan(a, max_kv);
    if (plan.parts < 2 || !ws || ws_bytes < plan.ws_bytes) plan = Bs16Plan{nblk, 0, 1, 0};
    static FvkLdsConfigured configured, configured_split;
    if (plan.full > 0) {
        if (int rc = fvk_config_lds(configured, (const void*)attn_bs16_kernel<false>, LDS_BYTES, "fvk_attn_block_sparse_bf16 (bs16)")) return rc;
        hipLaunchKernelGGL((attn_bs16_kernel<false>), dim3((unsigned)plan.full), dim3(256), LDS_BYTES, s, *a, q2k_idx, q2k_num, kv_block_sizes, max_kv, Bs16Split{});
        FVK_LAUNCH_CHECK();
    }
    if (plan.parts >= 2) {
        Bs16Split sp;
        sp.wg_base = (int)plan.full; sp.parts = plan.parts; sp.tail_rows = (int)(plan.tail * 256);
        sp.o_part = (float*)ws;
        sp.lse_part = sp.o_part + (long)plan.parts * sp.tail_rows * 128;
        if (int rc = fvk_config_lds(configured_split, (const void*)attn_bs16_kernel<false, 0, 0, true>, LDS_BYTES, "fvk_attn_block_sparse_bf16 (bs16, split)")) return rc;
        hipLaunchKernelGGL((attn_bs16_kernel<false, 0, 0, true>), dim3((unsigned)(plan.tail * plan.parts)), dim3(256), LDS_BYTES, s, *a, q2k_idx, q2k_num,
                           kv_block_sizes, max_kv, sp);
        FVK_LAUNCH_CHECK();
        hipLaunchKernelGGL(bs16_merge_parts_kernel, dim3((unsigned)((sp.tail_rows + 3) / 4)), dim3(256), 0, s, *a, sp);
        FVK_LAUNCH_CHECK();
    }
    return FVK_OK;
}
