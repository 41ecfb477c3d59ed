// Round 5: a STAND-ALONE two-kernel reproduction of the gfx950 co-residency fault — no library code.
//   aggressor wave (one per SIMD, leaves register-file room): a bf16 MFMA stream WHILE its own LDS-DMA pieces (buffer_load ... lds, 16 B per lane)
//                   are in flight — gemm_w1's cadence: one 1-KiB piece per 8 MFMAs, counted waits that leave 16 pieces flying, a barrier per 32 MFMAs
//   victim wave   : packed-fp32 VALU on register values next to its scalar twin; disagreements are counted in-kernel
// What scripts/coresidency/coresidency_strips.py established with the real gemm_w1 (MFMAs + in-flight DMA necessary and sufficient, fp8 MFMAs harmless,
// barriers an amplifier) is re-run here with synthetic aggressors, one ingredient at a time, and with victims of one packed / 64-bit opcode each.
//   hipcc --offload-arch=gfx950 -O3 coresidency_repro.cpp -o coresidency_repro ;  ./coresidency_repro [victim launches per cell = 30]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

// ---------------------------------------------------------------- victims: FORM selects the instruction under test
// 0 the RoPE rotation (pk_mul, pk_mul op_sel, pk_add neg)   1 v_pk_mul_f32 plain   2 v_pk_add_f32 plain   3 v_pk_fma_f32   4 v_pk_mul_f32 op_sel only
// 5 v_add_f64 (a 64-bit result that is not VOP3P)            6 v_lshl_add_u64 (64-bit integer result)       7 scalar v_mul_f32 / v_add_f32 (control)
template <int FORM>
__global__ __launch_bounds__(64) void victim(const float* __restrict__ in, unsigned long long* __restrict__ counters, int iters) {
    const int gid = blockIdx.x * 64 + threadIdx.x;
    f32x2 x = {in[(2 * gid) & 4095], in[(2 * gid + 1) & 4095]};
    f32x2 c = {in[(gid + 7) & 4095] * 0.5f + 0.25f, in[(gid + 11) & 4095] * 0.5f + 0.3f};
    f32x2 s = {in[(gid + 13) & 4095] * 0.5f - 0.1f, in[(gid + 17) & 4095] * 0.5f - 0.2f};
    unsigned bad = 0, bad_lo = 0, bad_hi = 0;
    for (int it = 0; it < iters; ++it) {
        float e0, e1;      // expected, by scalar instructions
        f32x2 r;           // result of the instruction under test
        if (FORM == 0) {
            f32x2 t, u;
            asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(t) : "v"(x), "v"(c));
            asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1]" : "=v"(u) : "v"(x), "v"(s));
            asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,0]" : "=v"(r) : "v"(t), "v"(u));
            float t0, t1, u0, u1;
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(t0) : "v"(x[0]), "v"(c[0]));
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(t1) : "v"(x[1]), "v"(c[1]));
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(u0) : "v"(x[1]), "v"(s[0]));
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(u1) : "v"(x[0]), "v"(s[1]));
            asm volatile("v_sub_f32 %0, %1, %2" : "=v"(e0) : "v"(t0), "v"(u0));
            asm volatile("v_add_f32 %0, %1, %2" : "=v"(e1) : "v"(t1), "v"(u1));
        } else if (FORM == 1) {
            asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(c));
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(e0) : "v"(x[0]), "v"(c[0]));
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(e1) : "v"(x[1]), "v"(c[1]));
        } else if (FORM == 2) {
            asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(c));
            asm volatile("v_add_f32 %0, %1, %2" : "=v"(e0) : "v"(x[0]), "v"(c[0]));
            asm volatile("v_add_f32 %0, %1, %2" : "=v"(e1) : "v"(x[1]), "v"(c[1]));
        } else if (FORM == 3) {
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(c), "v"(s));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(e0) : "v"(x[0]), "v"(c[0]), "v"(s[0]));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(e1) : "v"(x[1]), "v"(c[1]), "v"(s[1]));
        } else if (FORM == 4) {
            asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1]" : "=v"(r) : "v"(x), "v"(s));
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(e0) : "v"(x[1]), "v"(s[0]));
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(e1) : "v"(x[0]), "v"(s[1]));
        } else if (FORM == 5) {
            double a = (double)x[0] + 1.5, b = (double)c[0] * 3.25, d1, d2;
            asm volatile("v_add_f64 %0, %1, %2" : "=v"(d1) : "v"(a), "v"(b));
            asm volatile("v_add_f64 %0, %1, %2" : "=v"(d2) : "v"(a), "v"(b));   // the same instruction twice: both must agree with each other
            const unsigned long long u1 = __double_as_longlong(d1), u2 = __double_as_longlong(d2);
            const double ex = a + b;
            const unsigned long long ue = __double_as_longlong(ex);
            r[0] = __uint_as_float((unsigned)u1); r[1] = __uint_as_float((unsigned)(u1 >> 32));
            e0 = __uint_as_float((unsigned)ue); e1 = __uint_as_float((unsigned)(ue >> 32));
            if (u1 != u2) { ++bad; }
        } else if (FORM == 6) {
            unsigned long long a = ((unsigned long long)__float_as_uint(x[0]) << 20) ^ __float_as_uint(x[1]), b = __float_as_uint(c[0]), d1;
            asm volatile("v_lshl_add_u64 %0, %1, 3, %2" : "=v"(d1) : "v"(a), "v"(b));
            const unsigned long long ex = (a << 3) + b;
            r[0] = __uint_as_float((unsigned)d1); r[1] = __uint_as_float((unsigned)(d1 >> 32));
            e0 = __uint_as_float((unsigned)ex); e1 = __uint_as_float((unsigned)(ex >> 32));
        } else {
            float a0, a1;
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(a0) : "v"(x[0]), "v"(c[0]));
            asm volatile("v_add_f32 %0, %1, %2" : "=v"(a1) : "v"(x[1]), "v"(c[1]));
            r[0] = a0; r[1] = a1;
            e0 = x[0] * c[0]; e1 = x[1] + c[1];
        }
        const bool wl = __float_as_uint(r[0]) != __float_as_uint(e0), wh = __float_as_uint(r[1]) != __float_as_uint(e1);
        bad += (wl || wh); bad_lo += wl; bad_hi += wh;
        // keep the operands moving, bounded, from SCALAR results only
        x[0] = x[0] * 0.75f + 0.01f * (float)((it * 37 + gid) & 15) - 0.05f;
        x[1] = x[1] * -0.5f + 0.02f * (float)((it * 11 + gid) & 7);
    }
    if (bad) { atomicAdd(&counters[0], (unsigned long long)bad); atomicAdd(&counters[1], (unsigned long long)bad_lo); atomicAdd(&counters[2], (unsigned long long)bad_hi); }
}

// ---------------------------------------------------------------- aggressor
// MF: 0 no MFMAs, 1 v_mfma_f32_16x16x32_bf16 on AGPRs, 2 v_mfma_f32_32x32x16_bf16, 3 v_mfma_scale_f32_16x16x128_f8f6f4 (fp8)
// DEPTH: LDS-DMA pieces left in flight by the counted waits (0 = no DMA at all).  BAR: a workgroup barrier per 32 MFMAs.
template <int MF, int DEPTH, bool BAR>
__global__ __launch_bounds__(256, 1) void aggressor(const float* __restrict__ src, long src_bytes, float* __restrict__ out, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef __attribute__((address_space(3))) void lds_void;
    asm volatile("" ::: "v151");      // 152 arch VGPRs + 256 AGPRs = gemm_w1's 408 registers: 104 left for a neighbour
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    bf16x8 a[8], b[8];
    for (int i = 0; i < 8; ++i)
        for (int e = 0; e < 8; ++e) { a[i][e] = (__bf16)(src[(threadIdx.x * 8 + i + e) & 4095]); b[i][e] = (__bf16)(src[(threadIdx.x * 5 + 3 * i + e) & 4095]); }
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, (int)(src_bytes > 0x7fffffffL ? 0x7fffffffL : src_bytes), 0x00020000);
    f32x4 acc[64];
    for (int i = 0; i < 64; ++i) for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
    unsigned off = (blockIdx.x * 4u + wave) * 65536u;   // every wave walks its own 1-KiB pieces through a large buffer (real L2 / HBM latency)
    const unsigned span = (unsigned)(src_bytes - 2048);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 64; ++i) {
            if (MF == 1) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a[(i >> 2) & 7]), "v"(b[i & 7]));
            else if (MF == 2) {
                if ((i & 3) == 0) {   // a 32x32x16 MFMA = 16 accumulator registers: four f32x4 at a time
                    typedef float f32x16 __attribute__((ext_vector_type(16)));
                    f32x16 t;
#pragma unroll
                    for (int q = 0; q < 4; ++q) for (int r = 0; r < 4; ++r) t[4 * q + r] = acc[i + q][r];
                    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(t) : "v"(a[(i >> 3) & 7]), "v"(b[(i >> 2) & 7]));
#pragma unroll
                    for (int q = 0; q < 4; ++q) for (int r = 0; r < 4; ++r) acc[i + q][r] = t[4 * q + r];
                }
            } else if (MF == 3) {
                if ((i & 1) == 0) {
                    const i32x8 wa = {__builtin_bit_cast(int, (float)a[i & 7][0]), 2, 3, 4, 5, 6, 7, 8 + i}, xa = {9, 10, 11, __builtin_bit_cast(int, (float)b[i & 7][1]), 13, 14, 15, 16};
                    acc[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wa, xa, acc[i], 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
                }
            }
            if (DEPTH > 0 && (i & 7) == 0) {   // one 1-KiB piece per 8 MFMAs into a 16-slot ring per wave
                const int slot = ((it * 8 + (i >> 3)) & 15);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(smem + wave * 16384 + slot * 1024), 16, lane * 16, (int)(off % span) & ~1023, 0, 0);
                off += 1024u * 257u;
            }
            if ((i & 31) == 31) {
                if (DEPTH >= 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
                else if (DEPTH >= 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                else if (DEPTH > 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (BAR) __builtin_amdgcn_s_barrier();
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float total = acc[0][0] + acc[63][3];
    if (DEPTH > 0) total += (float)smem[threadIdx.x];
    out[blockIdx.x * 256 + threadIdx.x] = total;
}

struct Agg { const char* name; void (*fn)(const float*, long, float*, int); int lds; };
struct Vic { const char* name; void (*fn)(const float*, unsigned long long*, int); };

int main(int argc, char** argv) {
    const int launches = argc > 1 ? atoi(argv[1]) : 30;
    const long SRC = 256L << 20;
    float* src; float* out; unsigned long long* cnt;
    CK(hipMalloc(&src, SRC)); CK(hipMalloc(&out, 256 * 256 * 4)); CK(hipMalloc(&cnt, 4 * 8));
    std::vector<float> h(SRC / 4);
    unsigned s = 12345u;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (float)(s >> 8) * (2.0f / 16777216.0f) - 1.0f; }
    CK(hipMemcpy(src, h.data(), SRC, hipMemcpyHostToDevice));
    hipStream_t sa, sb; CK(hipStreamCreate(&sa)); CK(hipStreamCreate(&sb));
    const Agg aggs[] = {
        {"none", nullptr, 0},
        {"bf16 MFMAs 16x16x32 + LDS-DMA, 16 pieces in flight, barriers  (gemm_w1's cadence)", aggressor<1, 16, true>, 65536},
        {"bf16 MFMAs 16x16x32 + LDS-DMA, 16 pieces in flight, NO barriers", aggressor<1, 16, false>, 65536},
        {"bf16 MFMAs 16x16x32 + LDS-DMA, 4 pieces in flight, barriers", aggressor<1, 4, true>, 65536},
        {"bf16 MFMAs 16x16x32 + LDS-DMA waited to completion every 32 MFMAs, barriers", aggressor<1, 1, true>, 65536},
        {"bf16 MFMAs 16x16x32, NO DMA, barriers", aggressor<1, 0, true>, 65536},
        {"NO MFMAs, LDS-DMA 16 pieces in flight, barriers", aggressor<0, 16, true>, 65536},
        {"bf16 MFMAs 32x32x16 + LDS-DMA, 16 pieces in flight, barriers", aggressor<2, 16, true>, 65536},
        {"fp8 MFMAs 16x16x128 + LDS-DMA, 16 pieces in flight, barriers", aggressor<3, 16, true>, 65536},
    };
    const Vic vics[] = {
        {"RoPE rotation (pk_mul, pk_mul op_sel, pk_add neg)", victim<0>}, {"v_pk_mul_f32", victim<1>}, {"v_pk_add_f32", victim<2>}, {"v_pk_fma_f32", victim<3>},
        {"v_pk_mul_f32 op_sel", victim<4>}, {"v_add_f64", victim<5>}, {"v_lshl_add_u64", victim<6>}, {"scalar v_mul_f32 / v_add_f32 (control)", victim<7>},
    };
    const int vic_iters = 2000, vic_blocks = 4096;
    for (const Agg& ag : aggs) {
        if (ag.fn) CK(hipFuncSetAttribute((const void*)ag.fn, hipFuncAttributeMaxDynamicSharedMemorySize, ag.lds));
        int ag_iters = 2000;
        if (ag.fn) {
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            CK(hipEventRecord(e0, sa));
            hipLaunchKernelGGL(ag.fn, dim3(256), dim3(256), ag.lds, sa, src, SRC, out, 500);
            CK(hipEventRecord(e1, sa)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            ag_iters = (int)(500 * 40.0f / (ms > 0.01f ? ms : 0.01f));
            if (ag_iters < 50) ag_iters = 50;
        }
        for (const Vic& vc : vics) {
            CK(hipMemset(cnt, 0, 32));
            CK(hipDeviceSynchronize());
            hipEvent_t a0, a1, v0, v1; CK(hipEventCreate(&a0)); CK(hipEventCreate(&a1)); CK(hipEventCreate(&v0)); CK(hipEventCreate(&v1));
            if (ag.fn) { CK(hipEventRecord(a0, sa)); hipLaunchKernelGGL(ag.fn, dim3(256), dim3(256), ag.lds, sa, src, SRC, out, ag_iters); CK(hipEventRecord(a1, sa)); }
            CK(hipEventRecord(v0, sb));
            for (int l = 0; l < launches; ++l) hipLaunchKernelGGL(vc.fn, dim3(vic_blocks), dim3(64), 0, sb, src, cnt, vic_iters);
            CK(hipEventRecord(v1, sb));
            CK(hipDeviceSynchronize());
            float ag_ms = 0.f, vic_ms = 0.f;
            CK(hipEventElapsedTime(&vic_ms, v0, v1));
            if (ag.fn) CK(hipEventElapsedTime(&ag_ms, a0, a1));
            unsigned long long c[4]; CK(hipMemcpy(c, cnt, 32, hipMemcpyDeviceToHost));
            printf("{\"aggressor\": \"%s\", \"victim\": \"%s\", \"wrong\": %llu, \"wrong_low_half\": %llu, \"wrong_high_half\": %llu, \"of\": %llu, \"aggressor_ms\": %.1f, \"victim_ms\": %.1f}\n",
                   ag.name, vc.name, c[0], c[1], c[2], (unsigned long long)launches * vic_blocks * 64ull * vic_iters, ag_ms, vic_ms);
            fflush(stdout);
        }
    }
    return 0;
}
