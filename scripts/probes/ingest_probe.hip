// How fast can a CU ingest 16-KiB blocks through the 16-B-per-lane load path?  (The roofline of the block-sparse attention kernel.)
// One 512-thread workgroup per CU; every wave streams pseudo-random 16-KiB blocks of a buffer, DEPTH wave-instructions (1 KiB each) in
// flight, either global -> LDS (LDS-DMA, mode 0) or global -> VGPR (buffer_load_dwordx4, mode 1), no compute.  Prints TB/s for a buffer
// that fits the L2s (8 MiB), one that fits the 256-MiB infinity cache (192 MiB) and one that does not (1 GiB).
// build: hipcc --offload-arch=gfx950 -O3 scripts/probes/ingest_probe.hip -o scripts/probes/ingest_probe.bin
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int DEPTH>
__global__ __launch_bounds__(512) void ingest(const unsigned char* buf, unsigned n_blocks, int iters, unsigned* sink) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef __attribute__((address_space(3))) void lds_void;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    unsigned state = blockIdx.x * 9781u + wave * 6271u + 12345u;
    u32x4 acc = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
        state = state * 1664525u + 1013904223u;
        const unsigned blk = __builtin_amdgcn_readfirstlane((state >> 8) % n_blocks);
        const unsigned char* p = buf + (size_t)blk * 16384;
        const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, 16384, 0x00020000);
        // 16 wave-instructions of 1 KiB = one 16-KiB block; DEPTH of them outstanding at most
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (MODE == 0) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void*)(smem + wave * 16384 + i * 1024), 16, lane * 16 + i * 1024, 0, 0, 0);
            } else {
                const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, lane * 16 + i * 1024, 0, 0);
                acc ^= v;
            }
            if (MODE == 0 && (i % DEPTH) == DEPTH - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH / 2) : "memory");
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (MODE == 1 && (acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) sink[0] = 1;
    if (MODE == 0 && smem[threadIdx.x] == 0x7f && buf == nullptr) sink[1] = 1;
#endif
}

template <int MODE, int DEPTH>
double run(const unsigned char* d, size_t bytes, unsigned* sink, int wgs) {
    const unsigned n_blocks = (unsigned)(bytes / 16384);
    const int iters = 2000;
    hipFuncSetAttribute((const void*)ingest<MODE, DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 16384);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((ingest<MODE, DEPTH>), dim3(wgs), dim3(512), 8 * 16384, 0, d, n_blocks, 200, sink);
    hipEventRecord(e0);
    hipLaunchKernelGGL((ingest<MODE, DEPTH>), dim3(wgs), dim3(512), 8 * 16384, 0, d, n_blocks, iters, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return (double)wgs * 8 * iters * 16384.0 / (ms * 1e-3) / 1e12;
}

int main() {
    const size_t cap = (size_t)1 << 30;
    unsigned char* d; unsigned* sink;
    if (hipMalloc(&d, cap) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(d, 1, cap);
    const size_t sizes[3] = {(size_t)8 << 20, (size_t)192 << 20, cap};
    const char* names[3] = {"8 MiB (L2-resident)", "192 MiB (infinity-cache-resident)", "1 GiB (HBM)"};
    for (int s = 0; s < 3; ++s) {
        const double dma8 = run<0, 8>(d, sizes[s], sink, 256), dma16 = run<0, 16>(d, sizes[s], sink, 256), vg = run<1, 16>(d, sizes[s], sink, 256);
        printf("%-36s LDS-DMA depth 8: %6.2f TB/s   depth 16: %6.2f TB/s   global->VGPR: %6.2f TB/s   (= %.1f / %.1f / %.1f B/clk/CU at 2.0 GHz)\n", names[s],
               dma8, dma16, vg, dma8 * 1e12 / 256 / 2e9, dma16 * 1e12 / 256 / 2e9, vg * 1e12 / 256 / 2e9);
    }
    return 0;
}
