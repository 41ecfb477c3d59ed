// Round 5, co-residency bug: synthetic VICTIM kernels that can be launched on a torch stream beside the REAL aggressor (gemm_w1 of
// scripts/probes/libfvk_bug.so) — scripts/coresidency/coresidency_victims.py.  The first synthetic pair (coresidency_probe.cpp: packed forms on register
// values) stayed clean beside a synthetic MFMA stream; the real victim's disassembly shows what it did not cover: the packed multiplies run IN
// PLACE on register pairs that a global_load_dwordx4 has JUST written (s_waitcnt vmcnt(N) directly in front), with op_sel:[0,1]
// op_sel_hi:[1,0] on the second source, followed by v_pk_add_f32 pairs with neg_lo / neg_hi.  Variants (one thing changes at a time):
//   0  packed ops on register-resident values (the first probe's victim)                                     [control]
//   1  values loaded from global memory each iteration, packed ops write OTHER registers (not in place)
//   2  loaded, packed ops IN PLACE on the loaded pairs — the real kernel's instruction sequence
//   3  as 2 with s_nop 7 between every s_waitcnt and the packed op that follows
//   4  as 2 with scalar v_mul_f32 / v_sub_f32 / v_add_f32 instead of the packed forms                        [control: round 4's second fence]
//   5  as 2, loads only CHECKED (no arithmetic): is the loaded DATA itself wrong?
// Every loaded value is re-derivable from its address (tab[i] = f(i)), so a wrong load is told apart from wrong arithmetic.
// counters: [0] lanes x iterations with a wrong packed result, [1] wrong low half, [2] wrong high half, [3] wrong loaded value
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC pk_victims.hip -o libpk_victims.so
#include <hip/hip_runtime.h>
#include <cstdint>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__host__ __device__ inline float tab_value(unsigned i) {  // exactly representable, varied mantissas
    const unsigned h = (i * 2654435761u) >> 9;               // 23 bits
    return (float)h * (1.0f / 8388608.0f) - 0.5f;
}

extern "C" __global__ void pkv_fill(float* tab, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) tab[i] = tab_value((unsigned)i);
}

template <int VARIANT>
__global__ __launch_bounds__(64) void pkv_victim(const float* __restrict__ tab, unsigned long long* __restrict__ counters, int iters, int n4) {
    const int gid = blockIdx.x * 64 + threadIdx.x;
    f32x2 w = {tab_value((unsigned)gid * 7u + 1u) + 1.0f, tab_value((unsigned)gid * 7u + 2u) - 1.0f};   // the "x" pair the tables multiply (register-resident, as the normalised q / k values are)
    unsigned bad = 0, bad_lo = 0, bad_hi = 0, bad_ld = 0;
    for (int it = 0; it < iters; ++it) {
        const unsigned i4 = ((unsigned)gid * 5u + (unsigned)it * 64u * 3u) % (unsigned)n4;   // 16-B chunk index of the "cos" table; "sin" = the next chunk
        const float* pc = tab + 4ull * i4;
        const float* ps = tab + 4ull * ((i4 + 1u) % (unsigned)n4);
        const float c0 = tab_value(4u * i4), c1 = tab_value(4u * i4 + 1u);
        const float s0 = tab_value(4u * ((i4 + 1u) % (unsigned)n4)), s1 = tab_value(4u * ((i4 + 1u) % (unsigned)n4) + 1u);
        // the load CHECK looks at elements 2, 3 of each 16-B chunk: the registers the in-place packed ops do not overwrite
        const float k0 = tab_value(4u * i4 + 2u), k1 = tab_value(4u * i4 + 3u);
        const float k2 = tab_value(4u * ((i4 + 1u) % (unsigned)n4) + 2u), k3 = tab_value(4u * ((i4 + 1u) % (unsigned)n4) + 3u);
        // expected (scalar, from the re-derived values): t = c * w ; u = {s0 * w1, s1 * w0} ; r = {t0 - u0, t1 - u1} ; q = {t0 + u0, t1 + u1}
        const float et0 = c0 * w[0], et1 = c1 * w[1], eu0 = s0 * w[1], eu1 = s1 * w[0];
        float er0, er1;
        asm volatile("v_sub_f32 %0, %1, %2" : "=v"(er0) : "v"(et0), "v"(eu0));
        asm volatile("v_sub_f32 %0, %1, %2" : "=v"(er1) : "v"(et1), "v"(eu1));
        f32x2 r;
        float l0, l1, l2, l3;   // the loaded values as the packed ops saw them (variant 5: only these are checked)
        if (VARIANT == 0) {
            f32x2 c = {c0, c1}, s = {s0, s1}, t, u;
            asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(t) : "v"(c), "v"(w));
            asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(u) : "v"(s), "v"(w));
            asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(t), "v"(u));
            l0 = k0; l1 = k1; l2 = k2; l3 = k3;
        } else if (VARIANT == 1) {
            f32x2 t, u;
            asm volatile(
                "global_load_dwordx4 v[40:43], %5, off\n\t"
                "global_load_dwordx4 v[44:47], %6, off\n\t"
                "s_waitcnt vmcnt(1)\n\t"
                "v_pk_mul_f32 v[48:49], v[40:41], %7\n\t"
                "s_waitcnt vmcnt(0)\n\t"
                "v_pk_mul_f32 v[50:51], v[44:45], %7 op_sel:[0,1] op_sel_hi:[1,0]\n\t"
                "v_pk_add_f32 %0, v[48:49], v[50:51] neg_lo:[0,1] neg_hi:[0,1]\n\t"
                "v_mov_b32 %1, v42\n\tv_mov_b32 %2, v43\n\tv_mov_b32 %3, v46\n\tv_mov_b32 %4, v47\n\t"
                : "=&v"(r), "=&v"(l0), "=&v"(l1), "=&v"(l2), "=&v"(l3)
                : "v"(pc), "v"(ps), "v"(w)
                : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "memory");
        } else if (VARIANT == 2 || VARIANT == 3) {
#define PKV_INPLACE(NOPS)                                                                                    \
            asm volatile(                                                                                    \
                "global_load_dwordx4 v[40:43], %5, off\n\t"                                                  \
                "global_load_dwordx4 v[44:47], %6, off\n\t"                                                  \
                "s_waitcnt vmcnt(1)\n\t" NOPS                                                                \
                "v_pk_mul_f32 v[40:41], v[40:41], %7\n\t"                                                    \
                "s_waitcnt vmcnt(0)\n\t" NOPS                                                                \
                "v_pk_mul_f32 v[44:45], v[44:45], %7 op_sel:[0,1] op_sel_hi:[1,0]\n\t"                       \
                "v_pk_add_f32 %0, v[40:41], v[44:45] neg_lo:[0,1] neg_hi:[0,1]\n\t"                          \
                "v_mov_b32 %1, v42\n\tv_mov_b32 %2, v43\n\tv_mov_b32 %3, v46\n\tv_mov_b32 %4, v47\n\t"      \
                : "=&v"(r), "=&v"(l0), "=&v"(l1), "=&v"(l2), "=&v"(l3)                                       \
                : "v"(pc), "v"(ps), "v"(w)                                                                   \
                : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "memory")
            if (VARIANT == 2) { PKV_INPLACE(""); } else { PKV_INPLACE("s_nop 7\n\t"); }
#undef PKV_INPLACE
        } else if (VARIANT == 4) {
            float r0s, r1s;
            asm volatile(
                "global_load_dwordx4 v[40:43], %6, off\n\t"
                "global_load_dwordx4 v[44:47], %7, off\n\t"
                "s_waitcnt vmcnt(1)\n\t"
                "v_mul_f32 v40, v40, %8\n\t"
                "v_mul_f32 v41, v41, %9\n\t"
                "s_waitcnt vmcnt(0)\n\t"
                "v_mul_f32 v44, v44, %9\n\t"
                "v_mul_f32 v45, v45, %8\n\t"
                "v_sub_f32 %0, v40, v44\n\t"
                "v_sub_f32 %5, v41, v45\n\t"
                "v_mov_b32 %1, v42\n\tv_mov_b32 %2, v43\n\tv_mov_b32 %3, v46\n\tv_mov_b32 %4, v47\n\t"
                : "=&v"(r0s), "=&v"(l0), "=&v"(l1), "=&v"(l2), "=&v"(l3), "=&v"(r1s)
                : "v"(pc), "v"(ps), "v"(w[0]), "v"(w[1])
                : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "memory");
            r[0] = r0s; r[1] = r1s;
        } else {
            asm volatile(
                "global_load_dwordx4 v[40:43], %5, off\n\t"
                "global_load_dwordx4 v[44:47], %6, off\n\t"
                "s_waitcnt vmcnt(0)\n\t"
                "v_mov_b32 %1, v42\n\tv_mov_b32 %2, v43\n\tv_mov_b32 %3, v46\n\tv_mov_b32 %4, v47\n\t"
                : "=&v"(r), "=&v"(l0), "=&v"(l1), "=&v"(l2), "=&v"(l3)
                : "v"(pc), "v"(ps), "v"(w)
                : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "memory");
            r[0] = er0; r[1] = er1;
        }
        const bool wl = __float_as_uint(r[0]) != __float_as_uint(er0), wh = __float_as_uint(r[1]) != __float_as_uint(er1);
        const bool wld = __float_as_uint(l0) != __float_as_uint(k0) || __float_as_uint(l1) != __float_as_uint(k1) ||
                         __float_as_uint(l2) != __float_as_uint(k2) || __float_as_uint(l3) != __float_as_uint(k3);
        bad += (wl || wh); bad_lo += wl; bad_hi += wh; bad_ld += wld;
        w[0] = w[0] * 0.999f + 0.001f; w[1] = w[1] * 0.998f - 0.001f;
    }
    if (bad | bad_ld) {
        atomicAdd(&counters[0], (unsigned long long)bad); atomicAdd(&counters[1], (unsigned long long)bad_lo);
        atomicAdd(&counters[2], (unsigned long long)bad_hi); atomicAdd(&counters[3], (unsigned long long)bad_ld);
    }
}

extern "C" int pkv_launch(int variant, const float* tab, unsigned long long* counters, int blocks, int iters, int n4, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    switch (variant) {
        case 0: hipLaunchKernelGGL(pkv_victim<0>, dim3(blocks), dim3(64), 0, s, tab, counters, iters, n4); break;
        case 1: hipLaunchKernelGGL(pkv_victim<1>, dim3(blocks), dim3(64), 0, s, tab, counters, iters, n4); break;
        case 2: hipLaunchKernelGGL(pkv_victim<2>, dim3(blocks), dim3(64), 0, s, tab, counters, iters, n4); break;
        case 3: hipLaunchKernelGGL(pkv_victim<3>, dim3(blocks), dim3(64), 0, s, tab, counters, iters, n4); break;
        case 4: hipLaunchKernelGGL(pkv_victim<4>, dim3(blocks), dim3(64), 0, s, tab, counters, iters, n4); break;
        case 5: hipLaunchKernelGGL(pkv_victim<5>, dim3(blocks), dim3(64), 0, s, tab, counters, iters, n4); break;
        default: return -1;
    }
    return (int)hipGetLastError();
}

extern "C" int pkv_fill_launch(float* tab, int n, void* stream) {
    hipLaunchKernelGGL(pkv_fill, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, tab, n);
    return (int)hipGetLastError();
}
