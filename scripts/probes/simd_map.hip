// Which SIMD does wave w of a 512-thread (8-wave) workgroup land on?  Prints HW_ID fields per wave for a few workgroups.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ __launch_bounds__(512, 2) void probe(unsigned* out) {
    extern __shared__ unsigned char smem[];
    unsigned hw = __builtin_amdgcn_s_getreg((4 /*HW_REG_HW_ID*/) | (0 << 6) | (31 << 11));
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = hw;
    smem[threadIdx.x] = 1;
}
int main() {
    const int nb = 512;
    unsigned* d; hipMalloc(&d, nb * 8 * 4);
    hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
    hipLaunchKernelGGL(probe, dim3(nb), dim3(512), 98304, 0, d);
    unsigned h[nb * 8]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int same = 0, tot = 0;
    for (int b = 0; b < nb; ++b) {
        if (b < 6) { printf("wg %d:", b); for (int w = 0; w < 8; ++w) printf(" w%d[simd %u wave %u cu %u]", w, (h[b*8+w] >> 4) & 3, h[b*8+w] & 15, (h[b*8+w] >> 8) & 15); printf("\n"); }
        for (int w = 0; w < 4; ++w) { tot++; same += (((h[b*8+w] >> 4) & 3) == ((h[b*8+w+4] >> 4) & 3)); }
    }
    printf("waves w and w+4 on the same SIMD: %d / %d\n", same, tot);
    int hist[8][4] = {};
    for (int b = 0; b < nb; ++b) for (int w = 0; w < 8; ++w) hist[w][(h[b*8+w] >> 4) & 3]++;
    for (int w = 0; w < 8; ++w) printf("wave %d simd histogram: %d %d %d %d\n", w, hist[w][0], hist[w][1], hist[w][2], hist[w][3]);
    return 0;
}
