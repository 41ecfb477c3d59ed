#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ i32x4 make_rsrc(const void* p, unsigned bytes) {
    const uint64_t a = reinterpret_cast<uint64_t>(p);
    i32x4 r;
    r[0] = __builtin_amdgcn_readfirstlane((int)(uint32_t)a);
    r[1] = __builtin_amdgcn_readfirstlane((int)((uint32_t)(a >> 32) & 0xffffu));
    r[2] = __builtin_amdgcn_readfirstlane((int)bytes);
    r[3] = 0x00020000;
    return r;
}
__global__ void k(const unsigned* p, u32x4* o, int nbytes, int so) {
#if defined(__HIP_DEVICE_COMPILE__)
    const i32x4 rs = make_rsrc(p, nbytes);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, nbytes, 0x00020000);
    u32x4 v;
    int voff = threadIdx.x * 16;
    int s = __builtin_amdgcn_readfirstlane(so);
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=&v"(v) : "v"(voff), "s"(rs), "s"(s) : "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    u32x4 w = __builtin_amdgcn_raw_buffer_load_b128(rb, voff, s, 0);
    o[threadIdx.x] = v;
    o[64 + threadIdx.x] = w;
#endif
}
int main() {
    unsigned* d; u32x4* o; const int n = 1 << 16;
    hipMalloc(&d, n * 4); hipMalloc(&o, 128 * 16);
    unsigned* h = new unsigned[n]; for (int i = 0; i < n; ++i) h[i] = i;
    hipMemcpy(d, h, n * 4, hipMemcpyHostToDevice);
    for (int so : {0, 64, 16384}) {
        hipLaunchKernelGGL(k, 1, 64, 0, 0, d, o, n * 4, so);
        unsigned r[512]; hipMemcpy(r, o, 128 * 16, hipMemcpyDeviceToHost);
        printf("soffset %d: asm lane0 %u lane1 %u | builtin lane0 %u lane1 %u (expect %u, %u)\n", so, r[0], r[4], r[256], r[260], so / 4, so / 4 + 4);
    }
    return 0;
}
