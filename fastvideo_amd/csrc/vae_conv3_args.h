// Launch arguments shared by the two 3x3-tap causal-conv kernel families of the Wan VAE decoder (vae_conv3.hip: 8 waves x (2 rows x 32 px x 96 ch) on
// 32x32x16 MFMAs; vae_conv3w.hip: 4 waves x (4 rows x 32 px x 96 ch), one wave per SIMD, on 16x16x32 MFMAs).  Host-side contract: fvk_vae_conv_bf16 /
// fvk_vae_conv_norm_bf16 (vae_conv.hip, include/fvk_amd.h).
#pragma once
#include "fvk_common.h"

namespace fvkc3 {

struct Conv3Args {
    const bf16_t* in; const bf16_t* w; const bf16_t* bias; bf16_t* out; const bf16_t* residual; float* out_f32;
    long out_fs, res_fs, plane_stride;
    int T, H, W, Hin, Win, Cin, Cout, KT;
    int ring, ring_start;
    int tiles_h, tiles_w, ntn;
    // fused WanRMS_norm (+SiLU) of this conv's output into the CONSUMER conv's input ring: norm_out != NULL.  Cout == 96: one wave holds all
    // channels of its pixels; Cout == 192: the two waves of a pixel row group hold 96 channels each and exchange their partial sums of squares
    // through LDS (one extra barrier).  write_raw == 0 drops the un-normed store (conv1 -> norm2 -> conv2 in a residual block).
    const float* norm_gamma; bf16_t* norm_out;
    int norm_ring, norm_slot0, norm_silu, write_raw;
};

enum { EPI_BIAS = 0, EPI_RESIDUAL = 1, EPI_FINAL = 2 };
constexpr unsigned OOB = 0xFFFFFF00u;  // a buffer offset past every descriptor's range: the load returns zeros

}  // namespace fvkc3

// vae_conv3w.hip: true if it served the launch (non-upsampling, bf16-output 3x3 convs); false = not eligible, the caller runs vae_conv3.hip's kernel
bool fvk_vae_conv3w_launch(fvkc3::Conv3Args a, int epilogue, hipStream_t s, int* rc);
// vae_convout.hip: true if it served the launch (conv_out: EPI_FINAL, 3 time taps, <= 3 output channels, <= 96 input channels)
bool fvk_vae_convout_launch(fvkc3::Conv3Args a, hipStream_t s, int* rc);
