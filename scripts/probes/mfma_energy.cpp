// Round 5: what the 1400-W cap leaves of the bf16 matrix pipe as a function of OPERAND DATA, registers only.
// scripts/probes/mfma_rate.cpp (round 3) fed near-constant operands (0.001 * small integers: few toggling bits); a kernel on real activations
// feeds random mantissas.  Same loops here with the operand registers filled from a per-lane xorshift stream:
//   data 0: zeros   data 1: round 3's near-constant values   data 2: uniform random bf16 in [-1, 1)   data 3: N(0,1)-like (sum of 4 uniforms)
// usage: mfma_energy <shape 0|1> <data 0..3> <seconds>      shape 0 = 32x32x16 (A reused 2x), shape 1 = 16x16x32 (A reused 4x)
// prints one line: shape data launches ms_per_launch TFLOP/s.   hipcc --offload-arch=gfx950 -O3 mfma_energy.cpp -o mfma_energy
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ inline unsigned xs(unsigned& s) { s ^= s << 13; s ^= s >> 17; s ^= s << 5; return s; }
__device__ inline float uni(unsigned& s) { return (float)(xs(s) >> 8) * (2.0f / 16777216.0f) - 1.0f; }

template <int SHAPE>
__global__ __launch_bounds__(256, 1) void mfma_loop(float* out, int iters, int data) {
    bf16x8 a[8], b[8];
    unsigned s = 0x9E3779B9u * (blockIdx.x * 256 + threadIdx.x + 1);
    for (int i = 0; i < 8; ++i)
        for (int e = 0; e < 8; ++e) {
            float va, vb;
            if (data == 0) { va = 0.f; vb = 0.f; }
            else if (data == 1) { va = 0.001f * (threadIdx.x % 7 + i); vb = 0.002f * (threadIdx.x % 5 + e); }
            else if (data == 2) { va = uni(s); vb = uni(s); }
            else { va = 0.866f * (uni(s) + uni(s) + uni(s) + uni(s)); vb = 0.866f * (uni(s) + uni(s) + uni(s) + uni(s)); }
            a[i][e] = (__bf16)va; b[i][e] = (__bf16)vb;
        }
    if (SHAPE == 0) {
        f32x16 acc[16];
        for (int i = 0; i < 16; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a[(i >> 1) & 7]), "v"(b[i & 7]));
        }
        float t = 0.f;
        for (int i = 0; i < 16; ++i) for (int r = 0; r < 16; ++r) t += acc[i][r];
        out[blockIdx.x * 256 + threadIdx.x] = t;
    } else {
        f32x4 acc[64];
        for (int i = 0; i < 64; ++i) for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 64; ++i) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a[(i >> 2) & 7]), "v"(b[i & 7]));
        }
        float t = 0.f;
        for (int i = 0; i < 64; ++i) for (int r = 0; r < 4; ++r) t += acc[i][r];
        out[blockIdx.x * 256 + threadIdx.x] = t;
    }
}

int main(int argc, char** argv) {
    const int shape = argc > 1 ? atoi(argv[1]) : 1, data = argc > 2 ? atoi(argv[2]) : 2;
    const double seconds = argc > 3 ? atof(argv[3]) : 2.0;
    float* out; hipMalloc(&out, 1024 * 256 * sizeof(float));
    const int iters = 20000, wgs = 256;
    auto run = [&]() {
        if (shape == 0) hipLaunchKernelGGL(mfma_loop<0>, dim3(wgs), dim3(256), 0, 0, out, iters, data);
        else hipLaunchKernelGGL(mfma_loop<1>, dim3(wgs), dim3(256), 0, 0, out, iters, data);
    };
    run(); hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const auto t0 = std::chrono::steady_clock::now();
    int n = 0;
    hipEventRecord(e0);
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) { run(); run(); hipDeviceSynchronize(); n += 2; }
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= n;
    const double flop = (double)wgs * 4 * iters * (shape == 0 ? 16 * 32768.0 : 64 * 16384.0);
    printf("{\"shape\": \"%s\", \"data\": %d, \"launches\": %d, \"ms_per_launch\": %.4f, \"tflops\": %.1f}\n", shape == 0 ? "32x32x16" : "16x16x32", data, n, ms, flop / ms / 1e9);
    return 0;
}
