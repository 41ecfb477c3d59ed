// Dense flash-style attention forward, head_dim 128, gfx950 — the design of scripts/probes/attn_w64.hip (its 32x32x16 predecessor, kept in the measurement build) (4 waves x 64 query rows, ONE wave per SIMD, the
// softmax software-pipelined inside the wave, fixed softmax reference, 128-key LDS-DMA stages) on v_mfma_f32_16x16x32_bf16.
//
// Why the other MFMA shape (profiles/r03_mfma_shapes.md): MI355X is power-bound, and a registers-only loop of 16x16x32 MFMAs (K = 32 per
// instruction, 4 accumulator registers) sustains 2.40 PFLOP/s against 2.13 for 32x32x16 (16 accumulator registers, K = 16): same FLOP per
// cycle, a fifth less register-file traffic per FLOP.  Everything that follows is the re-tiling that shape asks for.
//
//   wave = 64 query rows = 4 q blocks of 16;  sub-tile = 64 keys = 2 groups of 32 keys = 4 S^T tiles of 16 keys;  d = 128 = 4 k-steps of 32.
//   S^T tile = K rows (A, 16 keys x 32 d, from LDS) x Q^T (B, 32 d x 16 q, resident in the accumulator file)
//              -> lane (q = lane & 15, g = lane >> 4) holds 4 scores of its query row per tile: rows 4g + e.
//   O^T tile = V^T rows (A, 16 d x 32 key slots, from LDS) x P^T (B, 32 key slots x 16 q: lane g supplies slots 8g .. 8g+7).
//   P^T must be the bf16-packed S^T registers of the SAME lane (no exchange): lane g's eight slots of group G are its four scores of tile
//   (G, a) followed by its four scores of tile (G, b).  V^T arrives in fvk_v_transpose_bf16's layout (shared with every other kernel): chunk
//   4G + g of a V^T row holds keys 32G + 16(g >> 1) + 4(g & 1) + {0..3, 8..11}.  So tile (G, a) is fed the K ROWS in the order that puts
//   key 32G + 16(g >> 1) + 4(g & 1) + e into row 4g + e — rows 0..7, 16..23 of the group — and tile (G, b) the remaining rows (8..15,
//   24..31): a per-lane row offset in the K fragment address, no data movement.  The K tile's XOR swizzle uses exactly the row bits that
//   make those 16 rows hit 16 different chunk positions (bits 0-2 and 4), so every fragment read stays conflict-free.
//
//   iteration t:   MFMA stream:  P·V of sub-tile t-1 (64 MFMAs)  then  Q·K^T of sub-tile t+1 (64)      — 2048 matrix cycles, as attn_w64
//                  VALU stream:  softmax of sub-tile t: 64 scores per lane, two VALU instructions behind each MFMA
//   Every LDS fragment (32 per iteration, read 6 ahead) feeds FOUR consecutive MFMAs (the four q blocks).
//
// Register files (every MFMA is inline asm, files pinned by constraints — see scripts/probes/attn_w64.hip): accumulator file: O 128 + Q 64; arch VGPRs:
// S 2 x 64 + P 2 x 32 + 6 fragments + softmax temporaries.  Fixed softmax reference, exact per-row recompute, split-KV form, staging ring,
// barrier protocol and epilogue are attn_w64's.  Results agree with attn_w64 / attn_pp2 to rounding (another summation order inside the MFMA).
#include "fvk_common.h"

namespace {

constexpr int KT = 128;               // keys per staged tile (two 64-key sub-tiles)
constexpr int K_TILE = KT * 256;      // 128 keys x 128 d bf16
constexpr int V_TILE = 128 * KT * 2;  // 128 d x 128 keys bf16
constexpr int RING = 2;
constexpr int V_BASE = RING * K_TILE;
constexpr int LDS_BYTES = RING * (K_TILE + V_TILE);  // 131 072
constexpr float L_LIMIT = 1.2379400392853803e27f;  // 2^90: a row sum at or above it (or NaN) triggers the exact recompute

typedef __attribute__((address_space(3))) void lds_void;

// reductions over the four lanes {l, l^16, l^32, l^48} that share a query row (cold: first sub-tile, epilogue, exact pass)
__device__ __forceinline__ float row4_max(float v) {
    v = fmaxf(v, __shfl_xor(v, 16));
    return fmaxf(v, __shfl_xor(v, 32));
}
__device__ __forceinline__ float row4_sum(float v) {
    v += __shfl_xor(v, 16);
    return v + __shfl_xor(v, 32);
}

// LV (round 6 A/B): the row sums as fp32 VALU adds of the UNROUNDED probabilities (what the reference kernels sum:
// block_sparse_attn_triton.py:150, st_attn_triton.py:83) instead of a ninth d block of ones on the matrix pipe (1/17 of the MFMA work)
template <bool LV = false>
struct W16 {
    // registers of one wave (everything is indexed with compile-time constants after unrolling)
    bf16x8 qf[4][4];     // Q fragments [q block][k-step of 32 d]
    f32x4 o[4][9];       // O^T accumulators [q block][16-row d block]; block 8 = the row sums (its "V^T rows" are all ones, see below)
    f32x4 s[2][4][4];    // S^T [sub-tile parity][q block][tile: 2 * group + (a = 0 / b = 1)]
    bf16x8 pf[2][4][2];  // P^T, packed [sub-tile parity][q block][32-key group]
    float m_run[4];
    float l_acc[4];      // LV: this lane's share of the row sums (16 of a row's 64 scores per sub-tile)
    bf16x8 ones;         // A operand of the row-sum MFMA
    // fragment offsets INCLUDING the ring slot of the stage the pipelined loop reads next (V^T: stage j, K: stage j + 1); flipped
    // (^ 32 KiB) once per pair instead of adding a run-time slot offset in front of every read
    int fk[4], fv[4];    // K [k-step], V^T [32-key step of the stage]
    float c2;
    // LDS-DMA addressing: wave w moves pieces {w, w+4, ..., w+28} of K (4 key rows x 256 B) and of V^T (4 d rows x 256 B)
    __amdgpu_buffer_rsrc_t k_rsrc, v_rsrc;
    unsigned kv0, kv1, vv0, k_pstride, v_pstride, k_tile_bytes;  // kv0 / kv1: this lane's K source offset in even / odd pieces (the swizzle uses row bit 4)
    int pdst, n;
    unsigned char* smem;

    // piece I (0..7 = K(j+2), 8..15 = V^T(j+1)) of the set issued behind the barrier of pair j; stages past the end re-read stage 0 (harmless)
    __device__ __forceinline__ void issue_piece(int I, int j) const {  // I is a compile-time constant after unrolling
        if (I < 8) {
            const int t_ = j + 2 < n ? j + 2 : 0;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(k_rsrc, (lds_void*)(smem + (j & 1) * K_TILE + pdst + I * 4096), 16, (I & 1) ? kv1 : kv0,
                                                     __builtin_amdgcn_readfirstlane(I * k_pstride + (unsigned)t_ * k_tile_bytes), 0, 0);
        } else {
            const int t_ = j + 1 < n ? j + 1 : 0;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(v_rsrc, (lds_void*)(smem + V_BASE + ((j + 1) & 1) * V_TILE + pdst + (I - 8) * 4096), 16, vv0,
                                                     __builtin_amdgcn_readfirstlane((I - 8) * v_pstride + t_ * (KT * 2)), 0, 0);
        }
    }
    // V^T fragment of stage st: 32-key step kk (0..3), d block db (0..7);  K fragment of stage st: k-step ks, half hf, tile T = 2 * group + a/b
    // flip: read the OTHER ring slot than the one fv / fk currently point at (cold paths; a compile-time false in the loop)
    __device__ __forceinline__ bf16x8 frag_v(int kk, int db, bool flip = false) const {
        return *reinterpret_cast<const bf16x8*>(smem + (fv[kk] ^ (flip ? V_TILE : 0)) + db * 4096);
    }
    __device__ __forceinline__ bf16x8 frag_k(int ks, int hf, int T, bool flip = false) const {
        return *reinterpret_cast<const bf16x8*>(smem + (fk[ks] ^ (flip ? K_TILE : 0)) + (64 * hf + 32 * (T >> 1) + 8 * (T & 1)) * 256);
    }
    __device__ __forceinline__ void flip_slots() {
#pragma unroll
        for (int c = 0; c < 4; ++c) { fk[c] ^= K_TILE; fv[c] ^= V_TILE; }
    }
#define FVK_PV4(QB0, FR, PAR, G, DB)                                                                                                      \
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
    __device__ __forceinline__ void fence_o() {
        asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");
#pragma unroll
        for (int qb = 0; qb < 4; ++qb)
            asm volatile("" : "+a"(o[qb][0]), "+a"(o[qb][1]), "+a"(o[qb][2]), "+a"(o[qb][3]), "+a"(o[qb][4]), "+a"(o[qb][5]), "+a"(o[qb][6]), "+a"(o[qb][7]), "+a"(o[qb][8]));
    }
    // P·V of the sub-tile (stage st, half hf) from pf[PAR]: plain form for the tail
    // (block 8: the row sums, l = sum of the bf16-rounded P — the same values the numerator uses — from the matrix pipe instead of 64 v_add_f32
    // per sub-tile: the wave's instruction issue, not the matrix pipe, is what bounds this kernel)
    template <int PAR>
    __device__ __forceinline__ void pv_plain(int hf, bool flip) {
#pragma unroll
        for (int G = 0; G < 2; ++G)
#pragma unroll
            for (int db = 0; db < (LV ? 8 : 9); ++db) {
                const bf16x8 fr = db < 8 ? frag_v(2 * hf + G, db, flip) : ones;
                FVK_PV4(0, fr, PAR, G, db);
            }
    }
    // key (0..63, within the sub-tile) of score e of tile T in lane group g
    static __device__ __forceinline__ int key_of(int T, int g, int e) { return 32 * (T >> 1) + 16 * (g >> 1) + 4 * (g & 1) + 8 * (T & 1) + e; }
    // keys of the sub-tile at or beyond `valid` (0..64) get -inf (last stage only)
    template <int PAR>
    __device__ __forceinline__ void mask_keys(int valid, int g) {
#pragma unroll
        for (int qb = 0; qb < 4; ++qb)
#pragma unroll
            for (int T = 0; T < 4; ++T)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (key_of(T, g, e) >= valid) s[PAR][qb][T][e] = -INFINITY;
    }
    // row max of the sub-tile (this lane holds 16 of its row's 64 scores)
    template <int PAR, int QB>
    __device__ __forceinline__ float row_max() const {
        float mx = s[PAR][QB][0][0];
#pragma unroll
        for (int T = 0; T < 4; ++T)
#pragma unroll
            for (int e = 0; e < 4; ++e) mx = fmaxf(mx, s[PAR][QB][T][e]);
        return row4_max(mx);
    }
    // p = exp2(s*c2 - mc) for the lane's 16 scores of q block QB, packed to bf16 (the row sums come out of the P·V MFMAs, block 8)
    template <int PAR, int QB>
    __device__ __forceinline__ void exp_pack(float mc) {
#pragma unroll
        for (int G = 0; G < 2; ++G)
#pragma unroll
            for (int x = 0; x < 8; ++x) {
                const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s[PAR][QB][2 * G + (x >> 2)][x & 3], c2, -mc));
                if (LV) l_acc[QB] += p;
                pf[PAR][QB][G][x] = (bf16_t)p;
            }
    }
    template <int PAR>
    __device__ __forceinline__ void exp_pack_all() {
        exp_pack<PAR, 0>(m_run[0] * c2);
        exp_pack<PAR, 1>(m_run[1] * c2);
        exp_pack<PAR, 2>(m_run[2] * c2);
        exp_pack<PAR, 3>(m_run[3] * c2);
    }
    // exact online-softmax step (new running max first; O and l rescaled): the slow path
    template <int PAR>
    __device__ __forceinline__ void softmax_exact() {
        const float mx[4] = {row_max<PAR, 0>(), row_max<PAR, 1>(), row_max<PAR, 2>(), row_max<PAR, 3>()};
        fence_o();  // the P·V MFMAs just issued have written O before the compiler's v_accvgpr_read
#pragma unroll
        for (int qb = 0; qb < 4; ++qb) {
            const float m_new = fmaxf(m_run[qb], mx[qb]);
            const float alpha = __builtin_amdgcn_exp2f((m_run[qb] - m_new) * c2);
#pragma unroll
            for (int d = 0; d < 9; ++d)  // block 8 = the running row sum: rescaled with O
#pragma unroll
                for (int r = 0; r < 4; ++r) o[qb][d][r] *= alpha;
            if (LV) l_acc[qb] *= alpha;
            m_run[qb] = m_new;
        }
        exp_pack_all<PAR>();
        fence_o();  // v_accvgpr_write -> MFMA source C
    }
    // Q·K^T of the sub-tile (ring slot st, half hf) -> s: plain form (prologue, exact pass)
    template <int PAR>
    __device__ __forceinline__ void qk_plain(int hf, bool flip) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int T = 0; T < 4; ++T) {
                const bf16x8 fr = frag_k(ks, hf, T, flip);
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
    // The exact pass (rare: only when a row's fixed-reference sum left the safe range): plain online softmax over all keys with a running
    // max and O rescaled by the VALU, one stage at a time through ring slot 0 (load, wait, barrier, compute — no overlap).  Every wave of
    // the workgroup takes part.  Leaves o (with its row-sum block) and m_run as the pipelined pass would have.
    __device__ __forceinline__ void exact_pass(int v_last, int g) {
        // after the pipelined pass fv points at the slot of stage n-1 and fk at the other one: slot 0 for both is a flip by parity
        const bool flip_v = ((n - 1) & 1) != 0, flip_k = (n & 1) != 0;
#pragma unroll
        for (int qb = 0; qb < 4; ++qb) {
            m_run[qb] = -1e30f;
            l_acc[qb] = 0.f;
#pragma unroll
            for (int d = 0; d < 9; ++d)
#pragma unroll
                for (int r = 0; r < 4; ++r) o[qb][d][r] = 0.f;
        }
        fence_o();
        for (int st = 0; st < n; ++st) {
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();  // every wave has finished reading slot 0
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(k_rsrc, (lds_void*)(smem + pdst + i * 4096), 16, (i & 1) ? kv1 : kv0,
                                                         __builtin_amdgcn_readfirstlane(i * k_pstride + (unsigned)st * k_tile_bytes), 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(v_rsrc, (lds_void*)(smem + V_BASE + pdst + i * 4096), 16, vv0,
                                                         __builtin_amdgcn_readfirstlane(i * v_pstride + st * (KT * 2)), 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            const int valid = st == n - 1 ? v_last : KT;
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {  // (fully unrolled: a runtime index into fk[] / fv[] would pin the whole register struct to scratch)
                const int vh = valid - 64 * hf < 64 ? valid - 64 * hf : 64;  // valid keys of this sub-tile
                if (vh > 0) {
                    qk_plain<0>(hf, flip_k);
                    if (vh < 64) mask_keys<0>(vh, g);
                    softmax_exact<0>();
                    pv_plain<0>(hf, flip_v);
                }
            }
        }
        fence_o();
    }

    // ---- iteration t = 2j + 1 + EVEN (pair j): 128 chunks of { 1 MFMA | half the softmax of one score }, pinned with sched_barrier.
    // MFMA stream: P·V of sub-tile t-1 (stage j, half EVEN, pf[EVEN]) then Q·K^T of sub-tile t+1 (stage j+1, half EVEN -> s[EVEN]); each of
    // the 32 fragments (read FD ahead) feeds four consecutive chunks (q blocks 0..3).  VALU stream: sub-tile t = s[1-EVEN] -> pf[1-EVEN]
    // against the fixed reference, software-pipelined over the scores: the even chunk of score x finishes it (row sum, bf16 pack), the odd
    // chunk takes the exponential of score x+1 and forms the exponent of score x+2.  Score x of the lane's 64: q block x >> 4, group
    // (x >> 3) & 1, slot x & 7 (tile a / b = bit 2, register = bits 0-1).  EVEN = 0 also carries this wave's 16 DMA pieces of set j
    // ({K(j+2), V^T(j+1)}: their slots were freed by the pair's barrier), one per 8 chunks.   MASK: keys >= valid of sub-tile t are masked.
    // ABL (measurement build; results are wrong, timing is what is measured): bit 0 no DMA pieces in the loop, bit 2 no softmax VALU in the loop
    template <int EVEN, bool MASK, int ABL = 0>
    __device__ __forceinline__ void iter(int j, int valid, int g) {
        constexpr int CUR = 1 - EVEN, FD = 6;
        if (MASK) {
            fence_s<CUR>();  // the previous iteration's last Q·K^T MFMAs wrote these registers a few instructions ago
            mask_keys<CUR>(valid, g);
        }
        const float mc[4] = {m_run[0] * c2, m_run[1] * c2, m_run[2] * c2, m_run[3] * c2};
        // the 32 LDS fragments of the iteration in MFMA order: L < 16: V^T (group L >> 3, d block L & 7), else K (k-step (L-16) >> 2, tile (L-16) & 3)
        auto load_frag = [&](int L) { return L < 16 ? frag_v(2 * EVEN + (L >> 3), L & 7) : frag_k((L - 16) >> 2, EVEN, (L - 16) & 3); };
        bf16x8 fr[FD];
#pragma unroll
        for (int i = 0; i < FD; ++i) fr[i] = load_frag(i);
        // pipeline prologue of the softmax: exponential of score 0, exponent of score 1
        float p_nx = __builtin_amdgcn_exp2f(__builtin_fmaf(s[CUR][0][0][0], c2, -mc[0]));
        float e_nx = __builtin_fmaf(s[CUR][0][0][1], c2, -mc[0]);
        // 34 operand slots x 4 q blocks = 136 MFMAs: slots 0-7 V^T(G = 0), 8 the row sums of G = 0 (A = ones), 9-16 V^T(G = 1), 17 the row
        // sums of G = 1, 18-33 K
#pragma unroll
        for (int m = 0; m < 136; ++m) {
            const int F = m >> 2, qb = m & 3;
            const bool is_ones = F == 8 || F == 17;
            const int L = F < 8 ? F : F < 17 ? F - 1 : F - 2;  // LDS fragment of this slot (ones slots: none)
            if (F < 18) {
                const int G = F >= 9, db = is_ones ? 8 : (F - 9 * G);
                if (is_ones) { if (!LV) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(o[qb][8]) : "v"(ones), "v"(pf[EVEN][qb][G])); }
                else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(o[qb][db]) : "v"(fr[L % FD]), "v"(pf[EVEN][qb][G]));
            } else {
                const int ks = (L - 16) >> 2, T = (L - 16) & 3;
                if (ks == 0) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=&v"(s[EVEN][qb][T]) : "v"(fr[L % FD]), "a"(qf[qb][ks]));
                else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(s[EVEN][qb][T]) : "v"(fr[L % FD]), "a"(qf[qb][ks]));
            }
            __builtin_amdgcn_sched_barrier(0);  // the MFMA FIRST: everything below runs in its shadow
            if (qb == 3 && !is_ones && L + FD < 32) fr[L % FD] = load_frag(L + FD);
            if (!(ABL & 1)) {
                if (ABL & 8) {  // A/B: the set spread over BOTH iterations — K(j+2) here in the odd one (one piece per 16 MFMAs), V^T(j+1) in the first half of the even one
                    if (EVEN == 0 && (m & 15) == 1 && m < 128) issue_piece(m >> 4, j);
                    if (EVEN == 1 && (m & 7) == 1 && m < 64) issue_piece(8 + (m >> 3), j);
                } else if (EVEN == 0 && (m & 7) == 1 && m < 128) issue_piece(m >> 3, j);
            }
            if (!(ABL & 4) && m < 128) {
                const int c = m >> 1;  // the score this chunk pair works on
                if ((m & 1) == 0) {
                    const int q_ = c >> 4, G = (c >> 3) & 1, x = c & 7;
                    if (LV) l_acc[q_] += p_nx;
                    pf[CUR][q_][G][x] = (bf16_t)p_nx;
                    if (x & 1) asm volatile("" : "+v"(pf[CUR][q_][G]));  // the pair's v_cvt_pk stays in this chunk
                } else {
                    if (c + 1 < 64) p_nx = __builtin_amdgcn_exp2f(e_nx);
                    if (c + 2 < 64) {
                        const int y = c + 2, q_ = y >> 4, G = (y >> 3) & 1, x = y & 7;
                        e_nx = __builtin_fmaf(s[CUR][q_][2 * G + (x >> 2)][x & 3], c2, -mc[q_]);
                    }
                    asm volatile("" : "+v"(p_nx), "+v"(e_nx));  // keep the chunk's work IN the chunk (a use here, before the scheduling barrier)
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
};

// SPLIT (fvk_attn_dense_split_bf16): the key axis is cut into `n_split` runs of whole 128-key stages and every (query block, head, batch,
// run) is its own workgroup; a run's result is written UN-merged (normalised O as fp32 rows + base-2 LSE) and attn_merge_splits_kernel
// combines the runs (a run with no keys writes LSE = -inf and is ignored).
template <int ABL = 0, bool PLAIN_IDS = false, bool SPLIT = false, bool LV = false>
__global__ __launch_bounds__(256, 1) void attn_w16_kernel(fvk_attn_args a, int n_split, float* o_part, float* lse_part) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int BMQ = 256;
    FVK_CLAIM_WHOLE_REGISTER_FILE();
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g = lane >> 4;
    const int nqb = (a.Sq + BMQ - 1) / BMQ;
    // XCD-aware deal: hardware workgroup id x lands on XCD x % 8 (its own L2); XCD c gets the CONTIGUOUS logical ids [c*q + min(c, r), ...), i.e.
    // consecutive query blocks of the same head, which stream the same K / V^T
    const int nblk = gridDim.x, xq = nblk >> 3, xr = nblk & 7, xcd = blockIdx.x & 7;
    const int bid = PLAIN_IDS ? (int)blockIdx.x : xcd * xq + (xcd < xr ? xcd : xr) + (int)(blockIdx.x >> 3);
    const int q_first = (bid % nqb) * BMQ;
    const int h = (bid / nqb) % a.H;
    const int b = (bid / (nqb * a.H)) % a.B;
    const int run = SPLIT ? bid / (nqb * a.H * a.B) : 0;
    // this workgroup's keys: stages [st0, st1) of the (b, h) slice; Skv_w = its key count (the last stage of the last run may be ragged)
    const int n_all = (a.Skv + KT - 1) / KT;
    const int st0 = SPLIT ? (int)((long)n_all * run / n_split) : 0, st1 = SPLIT ? (int)((long)n_all * (run + 1) / n_split) : n_all;
    const int n = st1 - st0;  // stages of this workgroup (SPLIT: may be 0)
    const int Skv_w = (st1 * KT < a.Skv ? st1 * KT : a.Skv) - st0 * KT;
    const int v_last = Skv_w - (n - 1) * KT;  // valid keys of the last stage, 1..128

    const bf16_t* qp = (const bf16_t*)a.q + (long)b * a.q_bs + (long)h * a.q_hs;
    const bf16_t* kp = (const bf16_t*)a.k + (long)b * a.k_bs + (long)h * a.k_hs + (long)st0 * KT * a.k_ss;
    const bf16_t* vtp = (const bf16_t*)a.vt + ((long)b * a.H + h) * 128L * a.Skv_pad + (long)st0 * KT;
    bf16_t* op = (bf16_t*)a.o + (long)b * a.o_bs + (long)h * a.o_hs;
    if (SPLIT && n <= 0) {  // an empty run: LSE = -inf (weight 0 in the merge); workgroup-uniform
        const int r = q_first + tid;
        if (r < a.Sq) lse_part[(((long)run * a.B + b) * a.H + h) * a.Sq + r] = -INFINITY;
        return;
    }

    W16<LV> w;
    w.smem = smem;
    w.n = n;
    w.c2 = a.scale * 1.4426950408889634f;
    // Q fragments of the wave's four 16-row blocks (B operand of S^T = K·Q^T): row q0 + 16*blk + l15, d = 32*ks + 8*g .. +8
    const int q0 = q_first + wave * 64;
    int qrow[4];
    bool q_ok[4];
#pragma unroll
    for (int qb = 0; qb < 4; ++qb) {
        const int r = q0 + 16 * qb + l15;
        q_ok[qb] = r < a.Sq;
        qrow[qb] = q_ok[qb] ? r : a.Sq - 1;
        // loaded STRAIGHT into the accumulator file (see scripts/probes/attn_w64.hip); invisible to the compiler's vmcnt bookkeeping: the prologue's
        // s_waitcnt vmcnt(0) covers them
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
            asm volatile("global_load_dwordx4 %0, %1, off" : "=a"(w.qf[qb][ks]) : "v"(qp + (long)qrow[qb] * a.q_ss + ks * 32 + g * 8) : "memory");
    }
    // LDS images: V^T row r, 16-B chunk c at r*256 + ((c ^ (r & 15)) << 4); K row r, chunk c at r*256 + ((c ^ ((r & 7) | ((r >> 1) & 8))) << 4)
    // (row bits 0-2 and 4: the 16 rows {0-7, 16-23} / {8-15, 24-31} of a tile's fragment read land on 16 different chunk positions).  The
    // hardware writes lane-linearly, so each lane fetches the SOURCE chunk that belongs at its linear position.  Key rows >= Skv are
    // outside the descriptor's range -> zeros.
    w.k_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)kp, 0, (unsigned)((((long)Skv_w - 1) * a.k_ss + 128) * 2), 0x00020000);
    w.v_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)vtp, 0, (unsigned)(256L * a.Skv_pad - 2L * st0 * KT), 0x00020000);
    const int r0 = 4 * wave + (lane >> 4);  // this lane's row in a piece set; piece I holds rows r0 + 16 I
    w.kv0 = (unsigned)(((long)r0 * a.k_ss) * 2) + (unsigned)(((lane & 15) ^ (r0 & 7)) << 4);
    w.kv1 = (unsigned)(((long)r0 * a.k_ss) * 2) + (unsigned)(((lane & 15) ^ ((r0 & 7) | 8)) << 4);
    w.vv0 = (unsigned)(r0 * a.Skv_pad * 2) + (unsigned)(((lane & 15) ^ (r0 & 15)) << 4);
    w.k_pstride = (unsigned)(16 * a.k_ss * 2);
    w.v_pstride = (unsigned)(16 * a.Skv_pad * 2);
    w.k_tile_bytes = (unsigned)(a.k_ss * 2 * KT);
    w.pdst = wave * 1024;
    // fragment offsets.  K: lane row l15 of a tile is key row (l15 < 8 ? l15 : l15 + 8) of the tile's 32-key group (+8 for the b tile, in
    // frag_k); its swizzle term (row bits 0-2 | bit 4 -> 3) is l15 itself.  V^T: row l15 of the d block, swizzle term l15.
#pragma unroll
    for (int c4 = 0; c4 < 4; ++c4) {
        w.fk[c4] = K_TILE + (l15 < 8 ? l15 : l15 + 8) * 256 + (((4 * c4 + g) ^ l15) << 4);  // pair 0 reads K of stage 1 (slot 1) ...
        w.fv[c4] = V_BASE + l15 * 256 + (((4 * c4 + g) ^ l15) << 4);                         // ... and V^T of stage 0 (slot 0)
    }
#pragma unroll
    for (int qb = 0; qb < 4; ++qb) {
        w.m_run[qb] = -1e30f;
        w.l_acc[qb] = 0.f;
#pragma unroll
        for (int d = 0; d < 9; ++d)
#pragma unroll
            for (int r = 0; r < 4; ++r) w.o[qb][d][r] = 0.f;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) w.ones[e] = (bf16_t)1.0f;
    // opaque from here on: a known constant would be re-materialised (v_mov) right in front of the asm MFMA that reads it — a VALU-write ->
    // MFMA-read hazard the compiler does not pad for asm statements (and a write into a register an in-flight MFMA may still be reading)
    asm volatile("" : "+v"(w.ones));

    w.fence_o();  // zero-initialised accumulators (v_accvgpr_write) -> first MFMA
#define WAIT_ALL() asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#define BAR()                                   \
    {                                           \
        __builtin_amdgcn_sched_barrier(0);      \
        __builtin_amdgcn_s_barrier();           \
        __builtin_amdgcn_sched_barrier(0);      \
    }
    // ---- prologue: K(0), V^T(0) -> slot 0, K(1) -> slot 1; Q·K^T(0), the reference + softmax(0), Q·K^T(1) --------------------------------
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(w.k_rsrc, (lds_void*)(smem + w.pdst + i * 4096), 16, (i & 1) ? w.kv1 : w.kv0,
                                                 __builtin_amdgcn_readfirstlane(i * w.k_pstride), 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(w.k_rsrc, (lds_void*)(smem + K_TILE + w.pdst + i * 4096), 16, (i & 1) ? w.kv1 : w.kv0,
                                                 __builtin_amdgcn_readfirstlane(i * w.k_pstride + (n > 1 ? w.k_tile_bytes : 0u)), 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(w.v_rsrc, (lds_void*)(smem + V_BASE + w.pdst + i * 4096), 16, w.vv0,
                                                 __builtin_amdgcn_readfirstlane(i * w.v_pstride), 0, 0);
    }
    WAIT_ALL()
    BAR()
    const int v0 = v_last < 64 ? v_last : 64;  // valid keys of the last stage's two sub-tiles
    const int v1 = v_last > 64 ? v_last - 64 : 0;
    w.template qk_plain<0>(0, true);  // stage 0 = the slot fk does NOT point at
    if (n == 1) w.template mask_keys<0>(v0, g);
    // the fixed reference: exact row max of the first sub-tile (it has at least one valid key)
    w.m_run[0] = w.template row_max<0, 0>();
    w.m_run[1] = w.template row_max<0, 1>();
    w.m_run[2] = w.template row_max<0, 2>();
    w.m_run[3] = w.template row_max<0, 3>();
    w.template exp_pack_all<0>();
    w.template qk_plain<1>(1, true);
    // ---- pairs j = 0 .. n-2: iterations t = 2j+1 and 2j+2 behind ONE barrier; the last pair masks sub-tile 2n-2 ------------------------------
    for (int j = 0; j + 2 < n; ++j) {  // straight-line body: a conditional inside would make the register assignment of the two paths meet with copies
        if (!(ABL & 2)) {
            WAIT_ALL()  // this wave's pieces of set j-1 (issued a whole pair ago) have landed
            BAR()       // every wave is past iteration 2j: the slots of K(j) and V^T(j-1) are free, set j-1 is visible
        }
        w.template iter<0, false, ABL>(j, 64, g);
        w.template iter<1, false, ABL>(j, 64, g);
        w.flip_slots();
    }
    if (n >= 2) {
        WAIT_ALL()
        BAR()
        w.template iter<0, false, ABL & 8>(n - 2, 64, g);
        w.template iter<1, true, ABL & 8>(n - 2, v0, g);
        w.flip_slots();
    }
    // ---- tail: P·V(2n-2), softmax of sub-tile 2n-1 (second half of the last stage, masked), P·V(2n-1) --------------------------------------
    WAIT_ALL()  // V^T(n-1) (and the harmless re-reads of stage 0) landed
    BAR()
    w.template pv_plain<0>(0, false);  // fv points at the slot of stage n-1
    if (v1 > 0) {  // workgroup-uniform
        w.template fence_s<1>();
        w.template mask_keys<1>(v1, g);
        w.template exp_pack_all<1>();
        w.template pv_plain<1>(1, false);
    }
    w.fence_o();  // the last MFMAs' results before the epilogue's v_accvgpr_read
    WAIT_ALL()  // the harmless re-reads of stage 0 have landed (the exact pass below re-uses the ring; afterwards the LDS can be re-assigned)
    // ---- epilogue: normalise and store; a row whose fixed-reference sum left the safe range (NaN, infinite or >= 2^90) is redone by the
    // exact pass and stored again — per ROW, so that a row's result never depends on which other rows share its wave or workgroup (the
    // sequence-parallel paths rely on that).  The whole workgroup takes part in the exact pass; only the flagged rows are overwritten.
    // Lane (l15, g) holds d = 16*db + 4*g + {0..3} of its query row.
    bool redo[4] = {false, false, false, false};
#define FVK_STORE_ROWS(ONLY_REDO)                                                                                    \
    _Pragma("unroll") for (int qb = 0; qb < 4; ++qb) {                                                               \
        const float l_tot = LV ? row4_sum(w.l_acc[qb]) : w.o[qb][8][0]; /* every row of block 8 holds the whole row sum */ \
        const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;                                                          \
        if (!(ONLY_REDO)) redo[qb] = !(l_tot < L_LIMIT);                                                             \
        if (q_ok[qb] && (!(ONLY_REDO) || redo[qb])) {                                                                \
            if (SPLIT) {                                                                                             \
                const long prow = (((long)run * a.B + b) * a.H + h) * a.Sq + qrow[qb];                               \
                float* orow = o_part + prow * 128;                                                                   \
                _Pragma("unroll") for (int d = 0; d < 8; ++d) {                                                      \
                    f32x4 v4;                                                                                        \
                    _Pragma("unroll") for (int e = 0; e < 4; ++e) v4[e] = w.o[qb][d][e] * inv;                       \
                    *reinterpret_cast<f32x4*>(orow + d * 16 + g * 4) = v4;                                           \
                }                                                                                                    \
                if (g == 0) lse_part[prow] = w.m_run[qb] * w.c2 + log2f(l_tot);                                      \
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
    FVK_STORE_ROWS(false)
    if (ABL == 0 && __syncthreads_or(redo[0] || redo[1] || redo[2] || redo[3])) {  // (timing ablations produce garbage sums: no redo)
        w.exact_pass(v_last, g);
        FVK_STORE_ROWS(true)
    }
#undef FVK_STORE_ROWS
#undef WAIT_ALL
#undef BAR
#endif  // __HIP_DEVICE_COMPILE__
}

template <int ABL = 0, bool PLAIN_IDS = false, bool LV = false>
int launch_w16(const fvk_attn_args* a, hipStream_t s) {
    static FvkLdsConfigured configured;
    if (int rc = fvk_config_lds(configured, (const void*)attn_w16_kernel<ABL, PLAIN_IDS, false, LV>, LDS_BYTES, "fvk_attn_dense_bf16 (w16)")) return rc;
    const long nblk = (long)((a->Sq + 255) / 256) * a->H * a->B;
    hipLaunchKernelGGL((attn_w16_kernel<ABL, PLAIN_IDS, false, LV>), dim3((unsigned)nblk), dim3(256), LDS_BYTES, s, *a, 1, (float*)nullptr, (float*)nullptr);
    FVK_LAUNCH_CHECK();
    return FVK_OK;
}

// out[b, row, h, :] = sum_r 2^(lse_r - max) * o_part[r] / sum_r 2^(lse_r - max): one wave per (b, h, row), 2 columns per lane; HBM-bound
// (reads n_split * 512 B, writes 256 B per row).
__global__ __launch_bounds__(256) void attn_merge_splits16_kernel(fvk_attn_args a, int n_split, const float* o_part, const float* lse_part) {
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

}  // namespace

// variant (measurement build): 2 = hardware workgroup order, 11.. = timing ablations (bits: 1 no DMA in the loop, 2 no barrier / wait in the
// loop, 4 no softmax VALU in the loop)
int fvk_attn_w16_launch(const fvk_attn_args* a, int variant, hipStream_t s) {
#if FVK_VARIANTS
    switch (variant) {
        case 2: return launch_w16<0, true>(a, s);
        case 11: return launch_w16<1>(a, s);
        case 12: return launch_w16<2>(a, s);
        case 14: return launch_w16<4>(a, s);
        case 17: return launch_w16<7>(a, s);
        case 18: return launch_w16<8>(a, s);
        case 19: return launch_w16<16>(a, s);  // no exact recompute (timing probe: are rows being redone?)
        case 20: return launch_w16<0, false, true>(a, s);  // row sums as fp32 VALU adds instead of the ninth d block (round 6 A/B)
        default: break;
    }
#endif
    (void)variant;
    return launch_w16<0>(a, s);
}

// split-KV form (called by fvk_attn_dense_split_bf16, attn_fwd.hip, after its argument checks)
int fvk_attn_w16_split_launch(const fvk_attn_args* a, int n_split, float* o_part, float* lse_part, hipStream_t s) {
    static FvkLdsConfigured configured;
    if (int rc = fvk_config_lds(configured, (const void*)attn_w16_kernel<0, false, true>, LDS_BYTES, "fvk_attn_dense_split_bf16")) return rc;
    const long nblk = (long)((a->Sq + 255) / 256) * a->H * a->B * n_split;
    hipLaunchKernelGGL((attn_w16_kernel<0, false, true>), dim3((unsigned)nblk), dim3(256), LDS_BYTES, s, *a, n_split, o_part, lse_part);
    FVK_LAUNCH_CHECK();
    const long rows = (long)a->B * a->H * a->Sq;
    hipLaunchKernelGGL(attn_merge_splits16_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, *a, n_split, (const float*)o_part, (const float*)lse_part);
    FVK_LAUNCH_CHECK();
    return FVK_OK;
}
