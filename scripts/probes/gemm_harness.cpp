// Stand-alone A/B harness for the GEMM kernels behind fvk_gemm_bf16 (no Python / torch: starts in a second on a fresh GPU box).
//   build:  hipcc --offload-arch=gfx950 -O2 scripts/probes/gemm_harness.cpp -o scripts/probes/gemm_harness.bin \
//               -Lfastvideo_amd -lfvk_amd -Wl,-rpath,'$ORIGIN/../../fastvideo_amd'
//   run:    scripts/probes/gemm_harness.bin [impl ...] [--only=<shape substring>]   (gemm_impl values; the first is the bit-exactness reference)
// For every shape: each impl's output is compared byte-for-byte with the first impl's, then the impls are timed interleaved
// (3 rounds x 10 launches, HIP events) and the median TFLOP/s is printed as one JSON line.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "../../include/fvk_amd.h"

#define CK(x)                                                                                   \
    do {                                                                                        \
        hipError_t e_ = (x);                                                                    \
        if (e_ != hipSuccess) {                                                                 \
            fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
            exit(2);                                                                            \
        }                                                                                       \
    } while (0)

__device__ inline uint32_t mix(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
// bf16 values ~ uniform(-1, 1) * scale (sum of two uniforms keeps a bell shape)
__global__ void fill_bf16(uint16_t* p, long n, uint32_t seed, float scale) {
    long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    const long stride = (long)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        const uint32_t h = mix((uint32_t)i * 2654435761u + seed), g = mix(h + 0x9e3779b9u);
        const float u = ((h >> 8) * (1.0f / 16777216.0f) + (g >> 8) * (1.0f / 16777216.0f) - 1.0f) * scale;
        uint32_t b = __float_as_uint(u);
        b += 0x7fffu + ((b >> 16) & 1u);
        p[i] = (uint16_t)(b >> 16);
    }
}
__global__ void fill_f32(float* p, long n, uint32_t seed) {
    long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i < n) p[i] = (mix((uint32_t)i + seed) >> 8) * (2.0f / 16777216.0f) - 1.0f;
}

struct Shape { const char* name; int M, N, K, epi; bool bias; };

int main(int argc, char** argv) {
    std::vector<int> impls;
    const char* only = nullptr;  // --only=<substring>: run the matching shapes only (PMC passes)
    for (int i = 1; i < argc; ++i) {
        if (strncmp(argv[i], "--only=", 7) == 0) only = argv[i] + 7;
        else impls.push_back(atoi(argv[i]));
    }
    if (impls.empty()) impls = {0, 3, 4};
    const int S = 32760, d = 1536, F = 8960;
    const Shape shapes[] = {
        {"edge[777,1000,192]+bias", 777, 1000, 192, FVK_EPI_NONE, true},
        {"edge[300,264,128]+resgate", 300, 264, 128, FVK_EPI_RESIDUAL_GATE, true},
        {"qkv[S,4608,1536]+bias", S, 3 * d, d, FVK_EPI_NONE, true},
        {"out[S,1536,1536]+resgate", S, d, d, FVK_EPI_RESIDUAL_GATE, true},
        {"ffn_in[S,8960,1536]+gelu", S, F, d, FVK_EPI_GELU_TANH, true},
        {"ffn_out[S,1536,8960]+resgate", S, d, F, FVK_EPI_RESIDUAL_GATE, true},
        {"sp8_qkv[4095,4608,1536]+bias", 4095, 3 * d, d, FVK_EPI_NONE, true},
        {"sp8_out[4095,1536,1536]+resgate", 4095, d, d, FVK_EPI_RESIDUAL_GATE, true},
        {"sp8_ffn_in[4095,8960,1536]+gelu", 4095, F, d, FVK_EPI_GELU_TANH, true},
        {"sp8_ffn_out[4095,1536,8960]+resgate", 4095, d, F, FVK_EPI_RESIDUAL_GATE, true},
        {"sp4_out[8190,1536,1536]+resgate", 8190, d, d, FVK_EPI_RESIDUAL_GATE, true},
        {"sp4_ffn_out[8190,1536,8960]+resgate", 8190, d, F, FVK_EPI_RESIDUAL_GATE, true},
        {"4k^3", 4096, 4096, 4096, FVK_EPI_NONE, false},
        {"8k^3", 8192, 8192, 8192, FVK_EPI_NONE, false},
    };
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    int rc = 0;
    for (const Shape& sh : shapes) {
        if (only && !strstr(sh.name, only)) continue;
        const long nx = (long)sh.M * sh.K, nw = (long)sh.N * sh.K, no = (long)sh.M * sh.N;
        uint16_t *x, *w, *bias, *res, *out, *ref;
        float* gate;
        CK(hipMalloc(&x, nx * 2)); CK(hipMalloc(&w, nw * 2)); CK(hipMalloc(&bias, sh.N * 2)); CK(hipMalloc(&res, no * 2));
        CK(hipMalloc(&out, no * 2)); CK(hipMalloc(&ref, no * 2)); CK(hipMalloc(&gate, sh.N * 4));
        fill_bf16<<<2048, 256, 0, st>>>(x, nx, 1u, 1.0f);
        fill_bf16<<<2048, 256, 0, st>>>(w, nw, 2u, 1.7f / sqrtf((float)sh.K));
        fill_bf16<<<64, 256, 0, st>>>(bias, sh.N, 3u, 0.5f);
        fill_bf16<<<2048, 256, 0, st>>>(res, no, 4u, 1.0f);
        fill_f32<<<(sh.N + 255) / 256, 256, 0, st>>>(gate, sh.N, 5u);
        CK(hipStreamSynchronize(st));
        const bool rg = sh.epi == FVK_EPI_RESIDUAL_GATE;
        auto run = [&](uint16_t* o) {
            int r = fvk_gemm_bf16(x, w, sh.bias ? bias : nullptr, o, sh.M, sh.N, sh.K, sh.K, sh.N, sh.epi, rg ? res : nullptr,
                                  rg ? gate : nullptr, sh.M, st);
            if (r != 0) { fprintf(stderr, "fvk_gemm_bf16 failed: %s\n", fvk_last_error()); exit(3); }
        };
        std::vector<uint16_t> href(no), hout(no);
        printf("{\"shape\": \"%s\"", sh.name);
        for (size_t k = 0; k < impls.size(); ++k) {
            fvk_set_tunable("gemm_impl", impls[k]);
            uint16_t* o = k == 0 ? ref : out;
            CK(hipMemsetAsync(o, 0xff, no * 2, st));
            run(o);
            CK(hipStreamSynchronize(st));
            if (k == 0) { CK(hipMemcpy(href.data(), ref, no * 2, hipMemcpyDeviceToHost)); continue; }
            CK(hipMemcpy(hout.data(), out, no * 2, hipMemcpyDeviceToHost));
            long bad = 0, first = -1;
            for (long i = 0; i < no; ++i)
                if (hout[i] != href[i]) { if (first < 0) first = i; ++bad; }
            printf(", \"impl%d_mismatch\": %ld", impls[k], bad);
            if (bad) { printf(", \"impl%d_first\": [%ld, %ld]", impls[k], first / sh.N, first % sh.N); rc = 1; }
        }
        const int rounds = 3, reps = sh.M >= 4000 ? 10 : 3;
        std::vector<std::vector<float>> ms(impls.size());
        for (int r = 0; r < rounds; ++r)
            for (size_t k = 0; k < impls.size(); ++k) {
                fvk_set_tunable("gemm_impl", impls[k]);
                run(out);
                CK(hipEventRecord(e0, st));
                for (int i = 0; i < reps; ++i) run(out);
                CK(hipEventRecord(e1, st));
                CK(hipEventSynchronize(e1));
                float t;
                CK(hipEventElapsedTime(&t, e0, e1));
                ms[k].push_back(t / reps);
            }
        for (size_t k = 0; k < impls.size(); ++k) {
            std::sort(ms[k].begin(), ms[k].end());
            const double tf = 2.0 * sh.M * sh.N * sh.K / (ms[k][rounds / 2] * 1e-3) / 1e12;
            printf(", \"impl%d_ms\": %.4f, \"impl%d_tflops\": %.1f", impls[k], ms[k][rounds / 2], impls[k], tf);
        }
        printf("}\n");
        fflush(stdout);
        fvk_set_tunable("gemm_impl", 0);
        hipFree(x); hipFree(w); hipFree(bias); hipFree(res); hipFree(out); hipFree(ref); hipFree(gate);
    }
    return rc;
}
