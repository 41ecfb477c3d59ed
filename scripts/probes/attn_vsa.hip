// Block-sparse attention for 64-row query blocks (the VSA sparse branch), head_dim 128, gfx950 — the KEY-SPLIT kernel ("attn_impl" 54;
// NOT the shipped kernel: measured 2.93 ms against 2.55-2.60 ms of attn_fwd.hip's two-list kernel at cfg2, kept as the experiment that
// rules out two suspected limits).  Same mathematics, operand orientation, LDS image and V^T layout as attn_fwd.hip's MODE_BLOCKS
// (S^T = K·Q^T, O^T = V^T·P^T with P^T straight from the packed S^T accumulators, fp32 online softmax in the exp2 domain, P rounded to
// bf16 before P·V; ref: fastvideo-kernel/python/fastvideo_kernel/triton_kernels/block_sparse_attn_triton.py:32-160,
// csrc/attention/block_sparse_h100.cu).
//
// The hypothesis.  LDS holds two lists x two 35.8-KiB stages per CU and no more, so a CU runs FOUR 32-row compute streams.  In
// attn_fwd.hip each is one wave per SIMD that walks Q·K^T (16 MFMAs) -> softmax -> P·V (16 MFMAs) serially, next to a loader wave, and
// the one-stage-ahead LDS-DMA makes the tile time look like the loaded arrival latency (~2 900 cycles).  This kernel removes both:
//   * all 8 waves compute: a list's 64 rows x 64 keys tile is split FOUR ways — wave = (row half, KEY half) — so every SIMD holds two
//     compute waves (one of each list), each with a chain half as long (8 + 8 MFMAs, a 16-score softmax), and one wave's softmax runs
//     under the other's MFMAs.  The two key halves of a row keep independent running (max, sum, O) and are merged once, at the end,
//     through LDS.  Fragment reads per CU are unchanged (the keys are split, not replicated).
//   * no loader waves: each wave fetches a quarter of its list's stage global -> VGPR (buffer_load_dwordx4, 9 per tile) TWO tiles ahead
//     and copies it into the stage with ds_write_b128 right after the barrier that frees it.
// One raw s_barrier per tile; loads are retired with counted vmcnt (tiles past the end of a list re-read its last tile).
// The result: correct (tests/test_gpu_kernels.py), and no faster — every variant moves ~23 B / clock / CU of K / V^T, with K blocks
// contiguous or strided alike (scripts/vsa_rstg_ab.py).  The bound is the per-CU ingest rate of 16-B-per-lane loads, i.e. bytes per FLOP
// at 64 query rows per fetched tile.  Two things learnt on the way are recorded at load_b128 / VS_LOAD below.
#include "gemm_common.h"

namespace {

constexpr int K_ROW_BYTES = 272, V_ROW_BYTES = 144;  // rows padded by one 16-B chunk: conflict-free, immediate-addressed fragment reads
constexpr int KC = 17;                                // 16-B chunks per padded K row = wave-instructions per K tile
constexpr int K_TILE_BYTES = 64 * K_ROW_BYTES;        // 17 408
constexpr int V_TILE_BYTES = 128 * V_ROW_BYTES;       // 18 432
constexpr int STAGE_BYTES = K_TILE_BYTES + V_TILE_BYTES;
constexpr int N_INSTR = KC + 18;                      // 35 wave-instructions (1 KiB each) per stage
constexpr int NLD = (N_INSTR + 3) / 4;                // per wave (4 waves per list): 9
constexpr int LIST_CAP = 2048;
constexpr int LDS_BYTES = 2 * 2 * STAGE_BYTES + 2 * LIST_CAP * 4;
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
typedef int i32x4_t __attribute__((ext_vector_type(4)));

// Raw buffer descriptor in SGPRs (base, stride 0, num_records = bytes, DATA_FORMAT 32 / raw) for the inline-asm loads below.
__device__ __forceinline__ i32x4_t make_rsrc(const void* p, unsigned bytes) {
    const unsigned long long addr = reinterpret_cast<unsigned long long>(p);
    i32x4_t r;
    r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)addr);
    r[1] = __builtin_amdgcn_readfirstlane((int)((unsigned)(addr >> 32) & 0xffffu));
    r[2] = __builtin_amdgcn_readfirstlane((int)bytes);
    r[3] = 0x00020000;
    return r;
}
// The staged loads are issued through INLINE ASM on purpose: hipcc's waitcnt pass tracks builtin / C++ loads per destination register and,
// across the loop back-edge, retires them with vmcnt(7) ... vmcnt(0) in front of the next tile's MFMAs — i.e. it waits for the loads it
// has just issued, every tile (seen in the ISA of the first version of this kernel and of attn_fwd.hip's RSTG loaders).  Loads the compiler
// does not know about are retired only by the counted waits written below.
__device__ __forceinline__ u32x4_t load_b128(int voff, i32x4_t rsrc, int soff) {
    u32x4_t v;
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=&v"(v) : "v"(voff), "s"(rsrc), "s"(soff) : "memory");
    return v;
}

template <int N>
__device__ __forceinline__ void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

__global__ __launch_bounds__(512, 2) void attn_vsa_kernel(fvk_attn_args a, const int32_t* q2k_idx, const int32_t* q2k_num,
                                                          const int32_t* kv_block_sizes, int max_kv) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const int pr = wave >> 2;        // which of the workgroup's two lists
    const int iw = wave & 3;         // index among the list's four waves (= the SIMD)
    const int rh = wave & 1;         // row half: query rows 32*rh .. +32 of the block
    const int kh = (wave >> 1) & 1;  // key half: keys 32*kh .. +32 of every 64-key tile
    const int nqb = a.Sq >> 6;       // query blocks (= lists) per head
    const int nwg = (nqb + 1) >> 1;
    const int qb = (blockIdx.x % nwg) * 2 + pr;
    const bool pair_ok = qb < nqb;
    const int h = (blockIdx.x / nwg) % a.H;
    const int b = blockIdx.x / (nwg * a.H);
    unsigned char* const smem_p = smem + pr * (2 * STAGE_BYTES);
    int32_t* const lds_lists = reinterpret_cast<int32_t*>(smem + 4 * STAGE_BYTES);

    const bf16_t* qp = (const bf16_t*)a.q + (long)b * a.q_bs + (long)h * a.q_hs;
    const bf16_t* kp = (const bf16_t*)a.k + (long)b * a.k_bs + (long)h * a.k_hs;
    const bf16_t* vtp = (const bf16_t*)a.vt + ((long)b * a.H + h) * 128L * a.Skv_pad;
    bf16_t* op = (bf16_t*)a.o + (long)b * a.o_bs + (long)h * a.o_hs;

    // ---- the two KV lists of the workgroup -> LDS: entry = block id | (valid keys << 24) ------------------------------------------------
    int n_tiles = 0, n_loop = 0;
    for (int p = 0; p < 2; ++p) {
        const int qb_p = (blockIdx.x % nwg) * 2 + p;
        int n_p = 0;
        if (qb_p < nqb) {
            const long meta_p = ((long)b * a.H + h) * nqb + qb_p;
            n_p = q2k_num[meta_p];
            n_p = n_p < LIST_CAP ? n_p : LIST_CAP;  // (longer lists are refused by the launcher)
            const int32_t* src = q2k_idx + meta_p * max_kv;
            for (int i = tid; i < n_p; i += 512) {
                const int id = src[i];
                lds_lists[p * LIST_CAP + i] = id | (kv_block_sizes[id] << 24);
            }
        }
        n_loop = n_p > n_loop ? n_p : n_loop;
        if (p == pr) n_tiles = n_p;
    }
    __syncthreads();
    auto get_tile = [&](int j, int& kv0, int& valid) {
        const int e = lds_lists[pr * LIST_CAP + j];
        kv0 = (e & 0xffffff) << 6;
        valid = e >> 24;
    };

    // ---- Q fragments (B operand of S^T = K·Q^T): row q0 + l31, d = 16*ks + 8*hi .. +8 -------------------------------------------------
    int qrow = qb * 64 + rh * 32 + l31;
    const bool q_ok = pair_ok && qrow < a.Sq;
    qrow = q_ok ? qrow : a.Sq - 1;
    bf16x8 qf[8];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) qf[ks] = ld_bf16x8(qp + (long)qrow * a.q_ss + ks * 16 + hi * 8);
    // vmcnt(0) as a REAL s_waitcnt instruction (hipcc's waitcnt pass understands it, unlike inline asm): the Q loads are retired here, once.
    // Left to the pass, they are re-waited in front of every tile's MFMAs with vmcnt(7) .. vmcnt(0) — which, with the staged loads below in
    // flight behind them, means draining the whole prefetch every tile.
    __builtin_amdgcn_s_waitcnt(0x0F70);

    // ---- staging: a stage is one linear array of 2240 16-B chunks (K rows of 17 chunks x 64, then V^T rows of 9 chunks x 128); wave-
    // instruction t moves chunks 64 t .. 64 t + 63.  This wave owns t = 4 i + iw; per-lane SOURCE offsets precomputed (pad chunks re-read
    // chunk 0 of their row); rows >= Skv are out of the descriptor's range and read as zeros. -------------------------------------------
    const i32x4_t k_rsrc = make_rsrc(kp, (unsigned)((((long)a.Skv - 1) * a.k_ss + 128) * 2));
    const i32x4_t v_rsrc = make_rsrc(vtp, (unsigned)(256L * a.Skv_pad));
    int ld_voff[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int t = i * 4 + iw;
        if (t < KC) {
            const int g = t * 64 + lane;
            const int kr = g / KC, kc = g % KC;
            ld_voff[i] = (int)(((long)kr * a.k_ss + (kc < KC - 1 ? kc : 0) * 8) * 2);
        } else if (t < N_INSTR) {
            const int g = (t - KC) * 64 + lane;
            const int vr = g / 9, vc = g % 9;
            ld_voff[i] = (vr * a.Skv_pad + (vc < 8 ? vc : 0) * 8) * 2;
        } else {
            ld_voff[i] = 0x7fffff00;  // the 36th slot: out of range -> zeros, never stored
        }
    }
    const int k_tile_stride = (int)(a.k_ss * 2);  // bytes per key row
    u32x4_t rb[2][NLD];
    int tk0, tvalid;
#define VS_LOAD(SET, T)                                                                                              \
    {                                                                                                                \
        get_tile((T) < n_tiles ? (T) : n_tiles - 1, tk0, tvalid);                                                    \
        int ks_ = __builtin_amdgcn_readfirstlane(tk0 * k_tile_stride);                                               \
        int vs_ = __builtin_amdgcn_readfirstlane(tk0 * 2);                                                           \
        /* v_readfirstlane -> SGPR -> VMEM soffset needs 5 wait states on gfx9; hipcc pads that for its own instructions, not for inline asm \
           (without this the first loads of a tile ran with the PREVIOUS soffset: every block but block 0 was garbled) */               \
        asm volatile("s_nop 4" : "+s"(ks_), "+s"(vs_)::"memory"); /* tied to both: the pad cannot be scheduled above their definition */ \
        _Pragma("unroll") for (int i = 0; i < NLD; ++i) {                                                            \
            if (i * 4 + 3 < KC) rb[SET][i] = load_b128(ld_voff[i], k_rsrc, ks_);                                     \
            else if (i * 4 >= KC) rb[SET][i] = load_b128(ld_voff[i], v_rsrc, vs_);                                   \
            else if (i * 4 + iw < KC) rb[SET][i] = load_b128(ld_voff[i], k_rsrc, ks_);                               \
            else rb[SET][i] = load_b128(ld_voff[i], v_rsrc, vs_);                                                    \
        }                                                                                                            \
    }
#define VS_STORE(SET, ST)                                                                                            \
    {                                                                                                                \
        _Pragma("unroll") for (int i = 0; i < NLD; ++i) {                                                            \
            const int t_ = i * 4 + iw;                                                                               \
            if (t_ < N_INSTR) *reinterpret_cast<u32x4_t*>((ST) + t_ * 1024 + lane * 16) = rb[SET][i];                \
        }                                                                                                            \
    }
#define VS_BARRIER()                                             \
    {                                                            \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       \
        __builtin_amdgcn_s_barrier();                            \
    }

    // fragment read bases (per lane); everything else is an immediate
    const int k_rbase = (kh * 32 + l31) * K_ROW_BYTES + hi * 16;
    const int v_rbase = K_TILE_BYTES + l31 * V_ROW_BYTES + kh * 64 + hi * 16;  // keys 32 kh .. : 64 B into the V^T row

    f32x16 o[4];
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
    float m_run = -1e30f, l_run = 0.f;
    const float c2 = a.scale * 1.4426950408889634f;
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    // ---- prologue: tiles 0 and 1 on their way, tile 0 in stage 0, tile 2 requested ------------------------------------------------------
    if (n_tiles > 0) {
        VS_LOAD(0, 0)
        VS_LOAD(1, 1)
        wait_vm<NLD>();
        VS_STORE(0, smem_p)
        VS_LOAD(0, 2)
    }
    VS_BARRIER()

    // One tile step; SET = the register set of tile j+1 (= (j+1) & 1)
#define VS_STEP(J, SET)                                                                                              \
    {                                                                                                                \
        const int j = (J);                                                                                           \
        if (j + 1 < n_tiles) { /* stage (j+1)&1 was read in step j-1, behind the last barrier: refill it, then request tile j+3 */ \
            wait_vm<NLD>();                                                                                          \
            VS_STORE(SET, smem_p + ((j + 1) & 1) * STAGE_BYTES)                                                      \
            VS_LOAD(SET, j + 3)                                                                                      \
        }                                                                                                            \
        if (j < n_tiles) {                                                                                           \
            const unsigned char* cur = smem_p + (j & 1) * STAGE_BYTES;                                               \
            int kv0_, valid_;                                                                                        \
            get_tile(j, kv0_, valid_);                                                                               \
            (void)kv0_;                                                                                              \
            /* S^T = K·Q^T for this wave's 32 keys: 8 k-steps, fragments read 4 ahead */                               \
            f32x16 s;                                                                                                \
            {                                                                                                        \
                bf16x8 fr[4];                                                                                        \
                _Pragma("unroll") for (int i = 0; i < 4; ++i) fr[i] = *reinterpret_cast<const bf16x8*>(cur + k_rbase + i * 32); \
                __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);                                                   \
                __builtin_amdgcn_s_setprio(1);                                                                       \
                _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                                      \
                    s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[i % 4], qf[i], i == 0 ? zero16 : s, 0, 0, 0);     \
                    if (i + 4 < 8) fr[i % 4] = *reinterpret_cast<const bf16x8*>(cur + k_rbase + (i + 4) * 32);       \
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                               \
                    if (i + 4 < 8) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                \
                }                                                                                                    \
                __builtin_amdgcn_s_setprio(0);                                                                       \
            }                                                                                                        \
            /* first V^T fragments: issued now, they land under the softmax */                                       \
            bf16x8 vr[4];                                                                                            \
            _Pragma("unroll") for (int i = 0; i < 4; ++i) vr[i] = *reinterpret_cast<const bf16x8*>(cur + v_rbase + i * 32 * V_ROW_BYTES); \
            /* online softmax over this wave's 32 keys (row q = lane & 31: 16 scores here, 16 in lane ^ 32) */        \
            if (valid_ < 32 * kh + 32) {                                                                             \
                _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                                     \
                    const int key = kh * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;                                       \
                    if (key >= valid_) s[r] = -INFINITY;                                                             \
                }                                                                                                    \
            }                                                                                                        \
            float mx = fmaxf(s[0], s[1]);                                                                            \
            _Pragma("unroll") for (int r = 2; r < 16; r += 2) mx = fmaxf(fmaxf(mx, s[r]), s[r + 1]);                 \
            {                                                                                                        \
                const auto sw_ = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false); \
                mx = fmaxf(__uint_as_float(sw_[0]), __uint_as_float(sw_[1]));                                        \
            }                                                                                                        \
            const float m_new = fmaxf(m_run, mx);                                                                    \
            if (!__all(m_new == m_run)) {                                                                            \
                float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c2);                                          \
                asm volatile("s_nop 1" : "+v"(alpha));                                                               \
                l_run *= alpha;                                                                                      \
                _Pragma("unroll") for (int d = 0; d < 4; ++d) _Pragma("unroll") for (int r = 0; r < 16; ++r)         \
                    asm volatile("v_mul_f32 %0, %1, %0" : "+v"(o[d][r]) : "v"(alpha));                               \
                m_run = m_new;                                                                                       \
            }                                                                                                        \
            const float mc = m_run * c2;                                                                             \
            float ps4[4] = {0.f, 0.f, 0.f, 0.f};                                                                     \
            _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                                         \
                const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s[r], c2, -mc));                               \
                s[r] = p;                                                                                            \
                ps4[r & 3] += p;                                                                                     \
            }                                                                                                        \
            l_run += (ps4[0] + ps4[1]) + (ps4[2] + ps4[3]);                                                          \
            bf16x8 pf[2];                                                                                            \
            _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) _Pragma("unroll") for (int jj = 0; jj < 8; ++jj)       \
                pf[kk][jj] = (bf16_t)s[kk * 8 + jj];                                                                 \
            /* O^T += V^T · P^T over this wave's 32 keys: 2 k-steps of 16 keys x 4 d-blocks of 32 */                   \
            __builtin_amdgcn_s_setprio(1);                                                                           \
            _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                                          \
                o[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vr[i % 4], pf[i >> 2], o[i & 3], 0, 0, 0);        \
                if (i + 4 < 8) vr[i % 4] = *reinterpret_cast<const bf16x8*>(cur + v_rbase + ((i + 4) & 3) * 32 * V_ROW_BYTES + ((i + 4) >> 2) * 32); \
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 1);                                                   \
                if (i + 4 < 8) __builtin_amdgcn_sched_group_barrier(0x100, 1, 1);                                    \
            }                                                                                                        \
            __builtin_amdgcn_s_setprio(0);                                                                           \
        }                                                                                                            \
        VS_BARRIER()                                                                                                 \
    }
    for (int j0 = 0; j0 < n_loop; j0 += 2) {
        VS_STEP(j0, 1)
        if (j0 + 1 < n_loop) VS_STEP(j0 + 1, 0)
    }
#undef VS_STEP
#undef VS_LOAD
#undef VS_STORE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the tail re-reads
    VS_BARRIER()                                       // every wave is done with the stages: they become the merge area
#undef VS_BARRIER

    // ---- merge the two key halves of every row: wave kh = 1 hands (m, l, O) to its kh = 0 partner through LDS ---------------------------
    // merge slot of (list pr, row half rh): 64 lanes x (64 O values + m + l) fp32, lane-major with a 4-B skew per lane (bank spread)
    float* const mg = reinterpret_cast<float*>(smem) + (pr * 2 + rh) * (64 * 67);
    if (kh == 1) {
#pragma unroll
        for (int d = 0; d < 4; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) mg[lane * 67 + d * 16 + r] = o[d][r];
        mg[lane * 67 + 64] = m_run;
        mg[lane * 67 + 65] = l_run;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (kh == 1) return;
    {
        const float m1 = mg[lane * 67 + 64], l1 = mg[lane * 67 + 65];
        const float m = fmaxf(m_run, m1);
        const float a0 = __builtin_amdgcn_exp2f((m_run - m) * c2), a1 = __builtin_amdgcn_exp2f((m1 - m) * c2);
        l_run = l_run * a0 + l1 * a1;
        m_run = m;
#pragma unroll
        for (int d = 0; d < 4; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[d][r] = o[d][r] * a0 + mg[lane * 67 + d * 16 + r] * a1;
    }
    float l_tot;
    {
        const auto sw_ = __builtin_amdgcn_permlane32_swap(__float_as_uint(l_run), __float_as_uint(l_run), false, false);
        l_tot = __uint_as_float(sw_[0]) + __uint_as_float(sw_[1]);
    }
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
        if (a.lse && hi == 0) a.lse[((long)b * a.H + h) * a.Sq + qrow] = m_run * c2 + log2f(l_tot);
    }
#endif  // __HIP_DEVICE_COMPILE__
}

}  // namespace

// called by fvk_attn_block_sparse_bf16 (attn_fwd.hip) for q_block = 64 after its argument checks
int fvk_attn_vsa_launch(const fvk_attn_args* a, const int32_t* q2k_idx, const int32_t* q2k_num, const int32_t* kv_block_sizes, int max_kv,
                        hipStream_t s) {
    static_assert(LDS_BYTES <= 163840 && 4 * 64 * 67 * 4 <= 4 * STAGE_BYTES, "LDS budget");
    static FvkLdsConfigured configured;
    if (int rc = fvk_config_lds(configured, (const void*)attn_vsa_kernel, LDS_BYTES, "fvk_attn_block_sparse_bf16 (key-split kernel)")) return rc;
    const long nlists = a->Sq / 64;
    const long nblk = ((nlists + 1) / 2) * a->H * a->B;
    hipLaunchKernelGGL(attn_vsa_kernel, dim3((unsigned)nblk), dim3(512), LDS_BYTES, s, *a, q2k_idx, q2k_num, kv_block_sizes, max_kv);
    FVK_LAUNCH_CHECK();
    return FVK_OK;
}
