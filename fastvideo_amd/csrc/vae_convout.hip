// conv_out of the Wan VAE decoder (96 -> 3 channels, 3x3x3 causal, fp32 planar output with the final clamp), "rolling three-frame" form — round 6.
// ref: WanDecoder3d.forward's head (fastvideo/models/vaes/wanvae.py:857-993: norm_out + SiLU (fused into the producer) -> conv_out), WanCausalConv3d
// (:160-207), the decode's `.float().clamp(-1, 1)` (:1210-1211).
//
// Why its own kernel.  The general 3x3 kernels (vae_conv3.hip / vae_conv3w.hip) put the output channels on the MFMA's N axis: with 3 of them a
// tile pads 3 -> 32 columns, walks K = 27 taps x 96 channels per OUTPUT frame and fetches every input frame three times (once per time tap):
// 0.88 ms per 16-frame launch at 480 x 832, 5.3 ms of a decode, at 95 useful TFLOP/s.  Here the three time taps ride the N axis too:
//   * one MFMA 16x16x32 multiplies 16 pixels x 32 input channels of ONE input frame j by a 16-row weight fragment whose rows are
//     (dt, co) = 3 time taps x 3 output channels — i.e. the frame's contribution to the THREE output frames j, j - 1, j - 2 at once; K per
//     input frame is 9 spatial taps x 96 channels, and every input frame is fetched ONCE per launch (T + 2 frames for T outputs);
//   * D's row m = 4 g + co lives in lane group g = lane >> 4, register co: group g holds the running sum of the output frame o with
//     o mod 3 == g, so the accumulators never move — the WEIGHT rows rotate instead (group g multiplies input frame j by time tap
//     dt = (j - g) mod 3, a per-lane LDS offset), a group stores + clears when its dt is 2 (the frame's last contribution);
//   * a 16 x 32-pixel workgroup tile (4 waves x 4 rows), the 18 x 34-pixel halo slab of one 32-channel chunk in LDS (conv3w's conflict-free
//     64-B rows), every slab fragment read once and used for the (up to) three dh taps that touch it; next chunk prefetched through
//     registers while the current one is multiplied; two workgroups per CU.
// Accumulation order of an output frame: input frames o, o + 1, o + 2 (dt = 0, 1, 2), within a frame channel chunks, slab rows, columns, dw,
// dh — independent of T, so any split of a decode into passes gives the same bits.  HBM-bound: (T + 2) x H x W x 96 x 2 B x 1.2 (halo) per launch.
#include "fvk_common.h"
#include "vae_conv3_args.h"

namespace {

using fvkc3::Conv3Args;

constexpr int CO_TH = 16, CO_TW = 32;                    // workgroup tile (pixels)
constexpr int CO_HH = CO_TH + 2, CO_WW = 40, CO_WWV = CO_TW + 2;  // slab: 18 rows pitched 40 pixels, 34 used
constexpr int CO_SLAB = CO_HH * CO_WW * 64;              // 46 080 B: one 32-channel chunk
constexpr int CO_WROW = 64, CO_WCC = 4 * CO_WROW, CO_WTAP = 3 * CO_WCC, CO_WDT = 9 * CO_WTAP + 64;  // weight image strides (dt stride padded: banks)
constexpr int CO_WBYTES = 3 * CO_WDT;
constexpr int CO_LDS = CO_SLAB + CO_WBYTES;
constexpr int CO_NLD = (CO_HH * CO_WWV * 4 + 255) / 256;  // 16-B loads per thread per chunk: 10

__global__ __launch_bounds__(256, 2) void vae_convout_kernel(Conv3Args a) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const slab = smem;
    unsigned char* const wl = smem + CO_SLAB;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    // XCD-contiguous tile ids (neighbouring tiles share halo columns / rows in one XCD's L2)
    const int nblk = gridDim.x, xq = nblk >> 3, xr = nblk & 7, xcd = blockIdx.x & 7;
    const int bid = xcd * xq + (xcd < xr ? xcd : xr) + (int)(blockIdx.x >> 3);
    const int tw_i = bid % a.tiles_w, th_i = bid / a.tiles_w;
    const int h0 = th_i * CO_TH, w0 = tw_i * CO_TW;
    const int Cin = a.Cin, ncc = Cin >> 5;
    const long frameE = (long)a.H * a.W * Cin;  // elements per input frame
    const int Ktot = 27 * Cin;

    // ---- weight image: wl[dt][tap][cc][row co (row 3 = zeros)][32 channels] -------------------------------------------------------------
    for (int i = tid; i < 3 * 9 * ncc * 4 * 4; i += 256) {
        const int k = i & 3, co = (i >> 2) & 3;
        int r = i >> 4;
        const int cc = r % ncc; r /= ncc;
        const int tap = r % 9, dt = r / 9;
        bf16x8 v;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (bf16_t)0.f;
        if (co < a.Cout) v = *reinterpret_cast<const bf16x8*>(a.w + (long)co * Ktot + (long)(dt * 9 + tap) * Cin + cc * 32 + k * 8);
        *reinterpret_cast<bf16x8*>(wl + dt * CO_WDT + tap * CO_WTAP + cc * CO_WCC + co * CO_WROW + k * 16) = v;
    }
    // this lane's weight-fragment row: m = l15 = 4 g' + co; lanes of g' = 3 (no output frame) and of co = 3 read the zero row
    const int gq = l15 >> 2, co_l = (gq < 3) ? (l15 & 3) : 3;
    const int a_lane = co_l * CO_WROW + g * 16;

    // ---- slab staging through registers: load i of this thread = 16-B chunk (idx & 3) of used slab pixel idx >> 2 ------------------------
    int ld_goff[CO_NLD];   // element offset inside a frame's chunk 0 (or -1: outside the image / past the slab)
    int ld_soff[CO_NLD];   // LDS byte offset
#pragma unroll
    for (int i = 0; i < CO_NLD; ++i) {
        const int idx = tid + 256 * i, c = idx & 3, q = idx >> 2;
        const int hh = q / CO_WWV, ww = q - hh * CO_WWV;
        const int h = h0 - 1 + hh, w = w0 - 1 + ww;
        const bool in_slab = hh < CO_HH;
        const bool ok = in_slab && (unsigned)h < (unsigned)a.H && (unsigned)w < (unsigned)a.W;
        const int p = hh * CO_WW + ww;
        ld_goff[i] = ok ? (h * a.W + w) * Cin + c * 8 : -1;
        ld_soff[i] = in_slab ? p * 64 + ((c ^ (((p >> 2) & 1) << 1)) << 4) : -1;
    }
    bf16x8 R[CO_NLD];
    auto issue_loads = [&](int j, int cc) {
        int sl = a.ring_start + j;
        sl = sl >= a.ring ? sl - a.ring : sl;
        const bf16_t* base = a.in + (long)sl * frameE + cc * 32;
#pragma unroll
        for (int i = 0; i < CO_NLD; ++i) {
            bf16x8 v;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (bf16_t)0.f;
            if (ld_goff[i] >= 0) v = *reinterpret_cast<const bf16x8*>(base + ld_goff[i]);
            R[i] = v;
        }
    };
    auto write_slab = [&]() {
#pragma unroll
        for (int i = 0; i < CO_NLD; ++i)
            if (ld_soff[i] >= 0) *reinterpret_cast<bf16x8*>(slab + ld_soff[i]) = R[i];
    };
    // pixel-fragment offsets: slab row s (0..5 of this wave), column group c (0..1), dw: pixel p = (4 wave + s) * 40 + 16 c + dw + l15, chunk g
    auto frag_off = [&](int s, int c, int dw) {
        const int p = (4 * wave + s) * CO_WW + 16 * c + dw + l15;
        return p * 64 + ((g ^ (((p >> 2) & 1) << 1)) << 4);
    };

    f32x4 acc[4][2];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 2; ++c) acc[r][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bias_e[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) bias_e[e] = (a.bias && e < a.Cout) ? (float)a.bias[e] : 0.f;
    const long HW = (long)a.H * a.W;

    const int nj = a.T + 2;
    issue_loads(0, 0);
    int jm3 = 0;  // j mod 3
    for (int j = 0; j < nj; ++j) {
        // time tap of this lane's weight rows for input frame j: dt = (j - g') mod 3
        int dt_l = jm3 - gq;
        dt_l = dt_l < 0 ? dt_l + 3 : dt_l;
        const int a_base = (gq < 3 ? dt_l * CO_WDT : 0) + a_lane;
        for (int cc = 0; cc < ncc; ++cc) {
            __syncthreads();  // the previous chunk's fragment reads are done (first pass: the weight image is written)
            write_slab();
            __syncthreads();
            {   // prefetch the next chunk while this one is multiplied
                int cn = cc + 1, jn = j;
                if (cn == ncc) { cn = 0; jn = j + 1; }
                if (jn < nj) issue_loads(jn, cn);
            }
            bf16x8 af[9];
#pragma unroll
            for (int t = 0; t < 9; ++t) af[t] = *reinterpret_cast<const bf16x8*>(wl + a_base + t * CO_WTAP + cc * CO_WCC);
#pragma unroll
            for (int s = 0; s < 6; ++s)
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int dw = 0; dw < 3; ++dw) {
                        const bf16x8 xf = *reinterpret_cast<const bf16x8*>(slab + frag_off(s, c, dw));
#pragma unroll
                        for (int dh = 0; dh < 3; ++dh) {
                            const int r = s - dh;  // output row of the wave that reads slab row s through tap dh
                            if (r >= 0 && r < 4) acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[dh * 3 + dw], xf, acc[r][c], 0, 0, 0);
                        }
                    }
        }
        // the lane group whose time tap was 2 holds a finished output frame o = j - 2: store (clamped, planar fp32) and clear
        const int o = j - 2;
        if (g == (jm3 == 2 ? 0 : jm3 + 1)) {   // g == (j - 2) mod 3 = (j + 1) mod 3 (never 3)
            if (o >= 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        const int h = h0 + 4 * wave + r, w = w0 + 16 * c + l15;
                        if (h < a.H && w < a.W) {
#pragma unroll
                            for (int e = 0; e < 3; ++e)
                                if (e < a.Cout) {
                                    const float v = acc[r][c][e] + bias_e[e];
                                    a.out_f32[(long)e * a.plane_stride + (long)o * HW + (long)h * a.W + w] = fminf(fmaxf(v, -1.0f), 1.0f);
                                }
                        }
                    }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 2; ++c) acc[r][c] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        jm3 = jm3 == 2 ? 0 : jm3 + 1;
    }
#endif  // __HIP_DEVICE_COMPILE__
}

}  // namespace

// called by fvk_vae_conv3_launch (vae_conv3.hip) for EPI_FINAL, KT = 3, Cout <= 3, no upsampling; false = not eligible
bool fvk_vae_convout_launch(fvkc3::Conv3Args a, hipStream_t s, int* rc) {
    if (a.KT != 3 || a.Cout > 3 || a.Cin % 32 != 0 || a.Cin > 96 || a.Hin != a.H || a.Win != a.W || !a.out_f32) return false;
    if ((long)a.H * a.W * a.Cin >= 0x7fffffffL) return false;  // per-frame element offsets are 32-bit
    a.tiles_h = (a.H + CO_TH - 1) / CO_TH;
    a.tiles_w = (a.W + CO_TW - 1) / CO_TW;
    static FvkLdsConfigured configured;
    *rc = fvk_config_lds(configured, (const void*)vae_convout_kernel, CO_LDS, "fvk_vae_conv_bf16 (conv_out)");
    if (*rc) return true;
    hipLaunchKernelGGL(vae_convout_kernel, dim3((unsigned)(a.tiles_h * a.tiles_w)), dim3(256), CO_LDS, s, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        fvk_set_error("%s:%d launch failed: %s", __FILE__, __LINE__, hipGetErrorString(e));
        *rc = FVK_ERR_LAUNCH;
    }
    return true;
}
