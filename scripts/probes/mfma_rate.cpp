// Sustained MFMA rate of the two bf16 shapes with nothing else running (registers only): is v_mfma_f32_16x16x32_bf16 cheaper in energy
// per FLOP than v_mfma_f32_32x32x16_bf16 on a power-bound MI355X?   hipcc --offload-arch=gfx950 -O3 mfma_rate.cpp -o mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// MODE 0: 32x32x16, 16 accumulators (256 regs), each A operand used twice in a row;  MODE 1: 16x16x32, 64 accumulators, A used 4x in a row;
// MODE 2: 16x16x32, A operand changes every MFMA
template <int MODE>
__global__ __launch_bounds__(256, 1) void mfma_loop(float* out, int iters) {
    bf16x8 a[8], b[8];
    for (int i = 0; i < 8; ++i)
        for (int e = 0; e < 8; ++e) { a[i][e] = (__bf16)(0.001f * (threadIdx.x % 7 + i)); b[i][e] = (__bf16)(0.002f * (threadIdx.x % 5 + e)); }
    if (MODE == 0) {
        f32x16 acc[16];
        for (int i = 0; i < 16; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a[(i >> 1) & 7]), "v"(b[i & 7]));
        }
        float s = 0.f;
        for (int i = 0; i < 16; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
        out[blockIdx.x * 256 + threadIdx.x] = s;
    } else {
        f32x4 acc[64];
        for (int i = 0; i < 64; ++i) for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 64; ++i) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a[MODE == 1 ? (i >> 2) & 7 : i & 7]), "v"(b[MODE == 1 ? i & 7 : (i >> 3) & 7]));
        }
        float s = 0.f;
        for (int i = 0; i < 64; ++i) for (int r = 0; r < 4; ++r) s += acc[i][r];
        out[blockIdx.x * 256 + threadIdx.x] = s;
    }
}

int main() {
    float* out; hipMalloc(&out, 1024 * 256 * sizeof(float));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int rep = 0; rep < 3; ++rep)
        for (int mode = 0; mode < 3; ++mode) {
            for (int wgs : {256, 512}) {
                auto run = [&]() {
                    if (mode == 0) hipLaunchKernelGGL(mfma_loop<0>, dim3(wgs), dim3(256), 0, 0, out, iters);
                    else if (mode == 1) hipLaunchKernelGGL(mfma_loop<1>, dim3(wgs), dim3(256), 0, 0, out, iters);
                    else hipLaunchKernelGGL(mfma_loop<2>, dim3(wgs), dim3(256), 0, 0, out, iters);
                };
                run(); hipDeviceSynchronize();
                hipEventRecord(e0); run(); run(); run(); hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
                const double flop = (double)wgs * 4 * iters * (mode == 0 ? 16 * 32768.0 : 64 * 16384.0);
                printf("mode %d (%s) wgs %d: %.3f ms  %.1f TFLOP/s\n", mode, mode == 0 ? "32x32x16" : mode == 1 ? "16x16x32 A x4" : "16x16x32 A x1", wgs, ms, flop / ms / 1e9);
            }
        }
    return 0;
}
