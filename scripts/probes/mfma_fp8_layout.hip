// Determines the operand layout of v_mfma_scale_f32_32x32x64_f8f6f4 (fp8 e4m3 x fp8 e4m3, unit scales) empirically:
// computes D = A·B^T for integer-valued A[32][64], B[32][64] under candidate lane->k mappings and reports which one matches.
#include <hip/hip_runtime.h>
#include <hip/hip_fp8.h>
#include <stdio.h>
#include <stdint.h>
#include <math.h>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

__global__ void k(const uint8_t* A, const uint8_t* B, float* D, int mode) {
    const int lane = threadIdx.x, r = lane & 31, hi = lane >> 5;
    uint8_t ab[32], bb[32];
    for (int j = 0; j < 32; ++j) {
        int kk;
        if (mode == 0) kk = hi * 32 + j;                       // 32 contiguous k per lane
        else kk = (j >> 4) * 32 + hi * 16 + (j & 15);          // two 16-byte blocks: k = hi*16.. and 32+hi*16..
        ab[j] = A[r * 64 + kk];
        bb[j] = B[r * 64 + kk];
    }
    v8i a, b;
    for (int j = 0; j < 8; ++j) {
        a[j] = ab[4*j] | (ab[4*j+1] << 8) | (ab[4*j+2] << 16) | (ab[4*j+3] << 24);
        b[j] = bb[4*j] | (bb[4*j+1] << 8) | (bb[4*j+2] << 16) | (bb[4*j+3] << 24);
    }
    v16f c = {0};
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
    // C/D layout: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
    for (int g = 0; g < 16; ++g) D[((g & 3) + 8 * (g >> 2) + 4 * hi) * 32 + r] = c[g];
}

static uint8_t f2fp8(float f) { __hip_fp8_e4m3 v(f); return *reinterpret_cast<uint8_t*>(&v); }
int main() {
    uint8_t hA[32 * 64], hB[32 * 64]; float fA[32 * 64], fB[32 * 64];
    for (int i = 0; i < 32 * 64; ++i) {
        fA[i] = (float)((i * 7 + (i >> 6) * 3) % 9 - 4) * 0.5f;  fB[i] = (float)((i * 5 + (i >> 6)) % 7 - 3);
        hA[i] = f2fp8(fA[i]); hB[i] = f2fp8(fB[i]);
    }
    float ref[32 * 32];
    for (int m = 0; m < 32; ++m) for (int n = 0; n < 32; ++n) { float s = 0; for (int kk = 0; kk < 64; ++kk) s += fA[m * 64 + kk] * fB[n * 64 + kk]; ref[m * 32 + n] = s; }
    uint8_t *dA, *dB; float* dD; hipMalloc(&dA, sizeof(hA)); hipMalloc(&dB, sizeof(hB)); hipMalloc(&dD, sizeof(ref));
    hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice);
    for (int mode = 0; mode < 2; ++mode) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD, mode);
        float h[32 * 32]; hipMemcpy(h, dD, sizeof(h), hipMemcpyDeviceToHost);
        double e1 = 0, e2 = 0;
        for (int m = 0; m < 32; ++m) for (int n = 0; n < 32; ++n) { e1 += fabs(h[m * 32 + n] - ref[m * 32 + n]); e2 += fabs(h[n * 32 + m] - ref[m * 32 + n]); }
        printf("mode %d: sum|D - A.B^T| = %g   sum|D^T - A.B^T| = %g   (D[0][1]=%g ref=%g)\n", mode, e1, e2, h[1], ref[1]);
    }
    return 0;
}
