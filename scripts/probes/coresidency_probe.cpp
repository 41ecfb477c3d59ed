// Round 5: a STAND-ALONE two-kernel reproduction attempt of round 4's co-residency bug (a small kernel's wave
// sharing a SIMD with a gemm_w1 wave — 408 of 512 registers, inline-asm 16x16x32 MFMAs on AGPR accumulators, LDS-DMA staging — got wrong LOW
// halves out of v_pk_mul_f32 / v_pk_add_f32 with op_sel / neg modifiers).  No library code: a synthetic VICTIM whose waves execute exactly those
// packed forms next to their scalar twins and count disagreements in-kernel, and a synthetic AGGRESSOR (one wave per SIMD on every CU) whose
// properties are varied ONE at a time: accumulators in AGPRs or arch VGPRs, the arch-VGPR count (= accum_offset), the MFMA shape, LDS-DMA
// traffic on / off.  Victim knobs: s_nop padding after each packed instruction, extra live registers (register-file alignment of its pairs).
// Both kernels run on two streams of ONE process; the victim launches are checked (events) to fall inside the aggressor's run.
//   hipcc --offload-arch=gfx950 -O3 coresidency_probe.cpp -o coresidency_probe ;  ./coresidency_probe [victim launches per cell = 40]
// Output: one JSON line per (aggressor, victim) cell: wrong = disagreeing (lane, iteration, form) events, of = events checked.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

// ---------------------------------------------------------------- victim
// NOP: s_nop count after every packed instruction (0 = none).  PADREGS: extra live VGPRs allocated below the pairs (moves their register numbers).
template <int NOP, int PADREGS>
__global__ __launch_bounds__(64) void victim(const float* __restrict__ in, unsigned long long* __restrict__ counters, int iters) {
    const int gid = blockIdx.x * 64 + threadIdx.x;
    float pad[PADREGS > 0 ? PADREGS : 1];
#pragma unroll
    for (int i = 0; i < (PADREGS > 0 ? PADREGS : 1); ++i) { pad[i] = in[(gid + i) & 4095]; asm volatile("" : "+v"(pad[i])); }
    f32x2 x = {in[(2 * gid) & 4095], in[(2 * gid + 1) & 4095]};
    f32x2 cs = {in[(gid + 7) & 4095] * 0.5f + 0.25f, in[(gid + 7) & 4095] * 0.5f + 0.25f};     // cos, cos
    f32x2 sn = {in[(gid + 13) & 4095] * 0.5f - 0.1f, in[(gid + 13) & 4095] * 0.5f - 0.1f};    // sin, sin
    unsigned bad = 0, bad_lo = 0, bad_hi = 0;
    for (int it = 0; it < iters; ++it) {
        f32x2 t, u, r;
        // the RoPE rotation as hipcc emits it: t = x * cos ; u = swap(x) * sin ; r = t + (-lo) u     [out_e = x_e c - x_o s, out_o = x_o c + x_e s]
        asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(t) : "v"(x), "v"(cs));
        if (NOP) asm volatile("s_nop %0" :: "n"(NOP > 0 ? NOP - 1 : 0));
        asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1]" : "=v"(u) : "v"(x), "v"(sn));
        if (NOP) asm volatile("s_nop %0" :: "n"(NOP > 0 ? NOP - 1 : 0));
        asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,0]" : "=v"(r) : "v"(t), "v"(u));
        if (NOP) asm volatile("s_nop %0" :: "n"(NOP > 0 ? NOP - 1 : 0));
        // the same arithmetic with scalar (non-packed) instructions
        float t0, t1, u0, u1, r0, r1;
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(t0) : "v"(x[0]), "v"(cs[0]));
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(t1) : "v"(x[1]), "v"(cs[1]));
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(u0) : "v"(x[1]), "v"(sn[0]));
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(u1) : "v"(x[0]), "v"(sn[1]));
        asm volatile("v_sub_f32 %0, %1, %2" : "=v"(r0) : "v"(t0), "v"(u0));
        asm volatile("v_add_f32 %0, %1, %2" : "=v"(r1) : "v"(t1), "v"(u1));
        const bool wl = __float_as_uint(r[0]) != __float_as_uint(r0), wh = __float_as_uint(r[1]) != __float_as_uint(r1);
        bad += (wl || wh); bad_lo += wl; bad_hi += wh;
        // keep the values moving (bounded): x <- 0.5 * scalar result + pad
        x[0] = r0 * 0.5f + pad[0] * 0.25f;
        x[1] = r1 * 0.5f - pad[0] * 0.25f;
    }
    float keep = 0.f;
#pragma unroll
    for (int i = 0; i < (PADREGS > 0 ? PADREGS : 1); ++i) keep += pad[i];
    if (keep == 123456.f) counters[3] = 1;  // (keeps the pad registers live)
    if (bad) { atomicAdd(&counters[0], (unsigned long long)bad); atomicAdd(&counters[1], (unsigned long long)bad_lo); atomicAdd(&counters[2], (unsigned long long)bad_hi); }
}

// ---------------------------------------------------------------- aggressor
// ACC: 0 = 16x16x32 on AGPR accumulators (64 x 4 = 256 AGPRs), 1 = 16x16x32 on arch-VGPR accumulators (32 x 4 = 128 VGPRs), 2 = 32x32x16 on AGPRs
// ARCH: arch VGPRs the kernel is made to claim (asm clobber of v[ARCH-1]) = its accum_offset.  DMA: LDS-DMA (buffer_load ... lds) traffic in the loop.
template <int ACC, int ARCH, bool DMA>
__global__ __launch_bounds__(256, 1) void aggressor(const float* __restrict__ src, float* __restrict__ out, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef __attribute__((address_space(3))) void lds_void;
    if (ARCH == 152) asm volatile("" ::: "v151");
    if (ARCH == 104) asm volatile("" ::: "v103");
    if (ARCH == 248) asm volatile("" ::: "v247");
    bf16x8 a[8], b[8];
    for (int i = 0; i < 8; ++i)
        for (int e = 0; e < 8; ++e) { a[i][e] = (__bf16)(src[(threadIdx.x * 8 + i + e) & 4095]); b[i][e] = (__bf16)(src[(threadIdx.x * 5 + 3 * i + e) & 4095]); }
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 4096 * 4, 0x00020000);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float total = 0.f;
    if (ACC == 0 || ACC == 1) {
        constexpr int NA = ACC == 0 ? 64 : 32;
        f32x4 acc[NA];
        for (int i = 0; i < NA; ++i) for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                if (ACC == 0) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a[(i >> 2) & 7]), "v"(b[i & 7]));
                else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a[(i >> 2) & 7]), "v"(b[i & 7]));
                if (DMA && (i & 15) == 0)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(smem + wave * 4096 + (i >> 4) * 1024), 16, (threadIdx.x & 63) * 16, 0, 0, 0);
            }
            if (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        total += acc[0][0] + acc[NA - 1][3];   // (a register-light use: a full reduction makes the compiler claim the whole arch file)
    } else {
        f32x16 acc[16];
        for (int i = 0; i < 16; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a[(i >> 1) & 7]), "v"(b[i & 7]));
                if (DMA && (i & 3) == 0)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(smem + wave * 4096 + (i >> 2) * 1024), 16, (threadIdx.x & 63) * 16, 0, 0, 0);
            }
            if (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        total += acc[0][0] + acc[15][15];
    }
    if (DMA) total += (float)smem[threadIdx.x];
    out[blockIdx.x * 256 + threadIdx.x] = total;
}

struct Agg { const char* name; void (*fn)(const float*, float*, int); int lds; };
struct Vic { const char* name; void (*fn)(const float*, unsigned long long*, int); };

int main(int argc, char** argv) {
    const int launches = argc > 1 ? atoi(argv[1]) : 40;
    float* src; float* out; unsigned long long* cnt;
    CK(hipMalloc(&src, 4096 * 4)); CK(hipMalloc(&out, 256 * 256 * 4)); CK(hipMalloc(&cnt, 4 * 8));
    std::vector<float> h(4096);
    unsigned s = 12345u;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (float)(s >> 8) * (2.0f / 16777216.0f) - 1.0f; }
    CK(hipMemcpy(src, h.data(), 4096 * 4, hipMemcpyHostToDevice));
    hipStream_t sa, sb; CK(hipStreamCreate(&sa)); CK(hipStreamCreate(&sb));
    const Agg aggs[] = {
        {"none", nullptr, 0},
        {"16x16x32 MFMAs, 256 AGPR accumulators, 152 arch VGPRs (gemm_w1's 408 registers), no LDS-DMA", aggressor<0, 152, false>, 0},
        {"16x16x32 MFMAs, 256 AGPR accumulators, 152 arch VGPRs, LDS-DMA in the loop", aggressor<0, 152, true>, 16384},
        {"16x16x32 MFMAs, 256 AGPR accumulators, 104 arch VGPRs (accum_offset 104)", aggressor<0, 104, false>, 0},
        {"16x16x32 MFMAs, 256 AGPR accumulators, arch VGPRs as the compiler allocates", aggressor<0, 0, false>, 0},
        {"16x16x32 MFMAs, accumulators in 128 ARCH VGPRs (no AGPR use), 248 arch VGPRs claimed", aggressor<1, 248, false>, 0},
        {"16x16x32 MFMAs, accumulators in 128 ARCH VGPRs, LDS-DMA in the loop", aggressor<1, 248, true>, 16384},
        {"32x32x16 MFMAs, 256 AGPR accumulators, 152 arch VGPRs", aggressor<2, 152, false>, 0},
        {"32x32x16 MFMAs, 256 AGPR accumulators, 152 arch VGPRs, LDS-DMA in the loop", aggressor<2, 152, true>, 16384},
    };
    const Vic vics[] = {
        {"packed RoPE forms, no padding", victim<0, 0>},
        {"packed RoPE forms, s_nop 1 after each packed instruction", victim<2, 0>},
        {"packed RoPE forms, s_nop 7 after each packed instruction", victim<8, 0>},
        {"packed RoPE forms, +3 live registers below the pairs (odd alignment shift)", victim<0, 3>},
        {"packed RoPE forms, +30 live registers", victim<0, 30>},
    };
    // calibrate: aggressor iterations for ~40 ms
    const int vic_iters = 2000, vic_blocks = 4096;
    for (const Agg& ag : aggs) {
        int ag_iters = 20000;
        if (ag.fn) {
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            CK(hipEventRecord(e0, sa));
            hipLaunchKernelGGL(ag.fn, dim3(256), dim3(256), ag.lds, sa, src, out, 2000);
            CK(hipEventRecord(e1, sa)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            ag_iters = (int)(2000 * 60.0f / (ms > 0.01f ? ms : 0.01f));
        }
        for (const Vic& vc : vics) {
            CK(hipMemset(cnt, 0, 32));
            CK(hipDeviceSynchronize());
            hipEvent_t a0, a1, v0, v1; CK(hipEventCreate(&a0)); CK(hipEventCreate(&a1)); CK(hipEventCreate(&v0)); CK(hipEventCreate(&v1));
            if (ag.fn) { CK(hipEventRecord(a0, sa)); hipLaunchKernelGGL(ag.fn, dim3(256), dim3(256), ag.lds, sa, src, out, ag_iters); CK(hipEventRecord(a1, sa)); }
            CK(hipEventRecord(v0, sb));
            for (int l = 0; l < launches; ++l) hipLaunchKernelGGL(vc.fn, dim3(vic_blocks), dim3(64), 0, sb, src, cnt, vic_iters);
            CK(hipEventRecord(v1, sb));
            CK(hipDeviceSynchronize());
            float ag_ms = 0.f, vic_ms = 0.f, lead = 0.f;
            CK(hipEventElapsedTime(&vic_ms, v0, v1));
            if (ag.fn) { CK(hipEventElapsedTime(&ag_ms, a0, a1)); CK(hipEventElapsedTime(&lead, a0, v1)); }
            unsigned long long c[4]; CK(hipMemcpy(c, cnt, 32, hipMemcpyDeviceToHost));
            printf("{\"aggressor\": \"%s\", \"victim\": \"%s\", \"wrong\": %llu, \"wrong_low_half\": %llu, \"wrong_high_half\": %llu, \"of\": %llu, "
                   "\"aggressor_ms\": %.2f, \"victim_ms\": %.2f, \"victims_done_ms_after_aggressor_start\": %.2f}\n",
                   ag.name, vc.name, c[0], c[1], c[2], (unsigned long long)launches * vic_blocks * 64ull * vic_iters, ag_ms, vic_ms, lead);
            fflush(stdout);
        }
    }
    return 0;
}
