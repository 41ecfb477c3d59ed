// Classifier-free-guidance combine + FlowUniPC multistep update of one denoising step as ONE elementwise kernel (HBM-bound; the
// reference issues ~25 eager kernels and materialises ~15 temporaries for it).
// ref: fastvideo/pipelines/stages/denoising.py:575-596 (noise_pred = uncond + g * (text - uncond) on bf16 tensors, then
//      scheduler.step(noise_pred, t, latents)),  fastvideo/models/schedulers/scheduling_flow_unipc_multistep.py:296-347
//      (convert_model_output), :364-489 (UniP B(h) predictor), :491-617 (UniC corrector), :649-724 (step).
// The scalar coefficients come from the host (fastvideo_amd/scheduler.py computes them exactly as the reference does, on 0-d fp32
// tensors); the tensor arithmetic below keeps the reference's operation order and rounding points, built with -ffp-contract=off so that no
// multiply-add is contracted: outputs are bit-identical to the eager reference in fp32.
#include "fvk_common.h"

namespace {

// uncontracted fp32 primitives (this file is built with -ffp-contract=off; hipcc fuses __fmul_rn + __fadd_rn into v_fma_f32 otherwise,
// which is 1 ulp away from the eager reference's separately rounded ops)
__device__ __forceinline__ float mul_(float a, float b) { return a * b; }
__device__ __forceinline__ float add_(float a, float b) { return a + b; }
__device__ __forceinline__ float sub_(float a, float b) { return a - b; }
__device__ __forceinline__ float div_(float a, float b) { return a / b; }

struct StepArgs {
    const bf16_t* text; const bf16_t* uncond;         // model outputs (uncond NULL = no guidance)
    const float* sample; const float* last_sample;    // latents x_t (fp32); sample before the previous predictor (corrector only)
    const float* m0; const float* m1;                 // converted model outputs (x0 predictions) of the previous two steps
    float* x0_out; float* sample_c_out; float* next_out; bf16_t* next_bf16;
    long n;
    float g, sigma_t;
    float cc_x, cc_m0, cc_B, c_rho0, c_rho_last, c_rk;  // corrector
    float pc_x, pc_m0, pc_B, p_rho0, p_rk;              // predictor
    int corr_order, pred_order;                          // 0 = no corrector; predictor order 1 or 2
    int recip;                                           // 1: x / rk is evaluated as x * (1.0f / rk) — torch's eager GPU kernels do that when the
    float c_irk, p_irk;                                  //    divisor is a 0-d CPU tensor (BinaryDivTrueKernel.cu: "may lose one bit"); 0: true division
};

__global__ __launch_bounds__(256) void cfg_unipc_step_kernel(StepArgs a) {
#pragma clang fp contract(off)  // hipcc contracts __fmul_rn + __fadd_rn into v_fma_f32 otherwise (1-ulp differences from the eager reference)
    const long i0 = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i0 >= a.n) return;
    const int cnt = (a.n - i0) < 4 ? (int)(a.n - i0) : 4;
    for (int e = 0; e < cnt; ++e) {
        const long i = i0 + e;
        // CFG combine on bf16 tensors: three ops, three roundings
        float np = (float)a.text[i];
        if (a.uncond) {
            const float u = (float)a.uncond[i];
            const float d = (float)(bf16_t)sub_(np, u);
            const float gd = (float)(bf16_t)mul_(a.g, d);
            np = (float)(bf16_t)add_(u, gd);
        }
        // convert_model_output: sigma_t (0-d fp32) * bf16 tensor -> bf16; fp32 sample - bf16 -> fp32
        float x = a.sample[i];
        const float x0 = sub_(x, (float)(bf16_t)mul_(a.sigma_t, np));
        const float m0 = a.m0 ? a.m0[i] : 0.f, m1 = a.m1 ? a.m1[i] : 0.f;
        if (a.corr_order > 0) {  // UniC: x <- c_x*last - c_m0*m0 - c_B*(rho0*(m1-m0)/rk + rho_last*(x0-m0))
            const float xt_ = sub_(mul_(a.cc_x, a.last_sample[i]), mul_(a.cc_m0, m0));
            const float d1t = mul_(a.c_rho_last, sub_(x0, m0));
            const float dm = sub_(m1, m0);
            const float corr = a.corr_order > 1 ? add_(mul_(a.c_rho0, a.recip ? mul_(dm, a.c_irk) : div_(dm, a.c_rk)), d1t) : d1t;
            x = sub_(xt_, mul_(a.cc_B, corr));
        }
        a.x0_out[i] = x0;
        a.sample_c_out[i] = x;
        // UniP with the shifted history (m0 <- x0, m1 <- old m0)
        float xn = sub_(mul_(a.pc_x, x), mul_(a.pc_m0, x0));
        if (a.pred_order > 1) {
            const float dp = sub_(m0, x0);
            xn = sub_(xn, mul_(a.pc_B, mul_(a.p_rho0, a.recip ? mul_(dp, a.p_irk) : div_(dp, a.p_rk))));
        }
        a.next_out[i] = xn;
        if (a.next_bf16) a.next_bf16[i] = (bf16_t)xn;  // latent_model_input = latents.to(bf16) of the next step (denoising.py:404)
    }
}

// DMD few-step sampling (DmdDenoisingStage, CausalDMDDenosingStage): x0 prediction + re-noising to the next timestep, one pass.
// ref: fastvideo/models/utils.py:138-175 (pred_noise_to_pred_video: fp64 arithmetic, cast back to the model-output dtype),
//      fastvideo/models/schedulers/scheduling_flow_match_euler_discrete.py:601-635 (add_noise: fp32, type_as(noise)),
//      call sites fastvideo/pipelines/stages/denoising.py:1382-1395, causal_denoising.py:273-310.
struct DmdArgs {
    const bf16_t* pred; const void* noisy; const bf16_t* noise; bf16_t* video; bf16_t* next;
    const double* sigma_t; const float* sigma_n;  // per frame (device): fp32 table values, sigma_t widened to fp64 by the host
    long per_frame, n;
    int noisy_f32;
};

__global__ __launch_bounds__(256) void dmd_step_kernel(DmdArgs a) {
#pragma clang fp contract(off)
    const long i0 = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i0 >= a.n) return;
    const int cnt = (a.n - i0) < 4 ? (int)(a.n - i0) : 4;
    for (int e = 0; e < cnt; ++e) {
        const long i = i0 + e;
        const long f = i / a.per_frame;
        const double x = a.noisy_f32 ? (double)((const float*)a.noisy)[i] : (double)(float)((const bf16_t*)a.noisy)[i];
        const double pv = x - a.sigma_t[f] * (double)(float)a.pred[i];  // product and difference rounded separately in fp64
        const bf16_t v = (bf16_t)(float)pv;                             // torch's double -> bf16 conversion goes through float
        a.video[i] = v;
        if (a.next) {
            const float sn = a.sigma_n[f];
            a.next[i] = (bf16_t)add_(mul_(sub_(1.0f, sn), (float)v), mul_(sn, (float)a.noise[i]));
        }
    }
}

}  // namespace

extern "C" int fvk_dmd_step(const void* pred_noise, const void* noisy_latent, int noisy_is_f32, const double* sigma_t, const void* noise,
                            const float* sigma_next, void* pred_video_out, void* next_out, long frames, long per_frame, void* stream) {
    FVK_CHECK(pred_noise && noisy_latent && sigma_t && pred_video_out && frames > 0 && per_frame > 0, FVK_ERR_ARG,
              "fvk_dmd_step: null pointer / empty");
    FVK_CHECK((next_out == nullptr) == (noise == nullptr) && (next_out == nullptr) == (sigma_next == nullptr), FVK_ERR_ARG,
              "fvk_dmd_step: noise, sigma_next and next_out come together (all NULL on the last step)");
    DmdArgs a{(const bf16_t*)pred_noise, noisy_latent, (const bf16_t*)noise, (bf16_t*)pred_video_out, (bf16_t*)next_out, sigma_t, sigma_next,
              per_frame, frames * per_frame, noisy_is_f32};
    const unsigned blocks = (unsigned)((a.n + 1023) / 1024);
    hipLaunchKernelGGL(dmd_step_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
    FVK_LAUNCH_CHECK();
    return FVK_OK;
}

// coef (HOST pointer, 13 floats): g, sigma_t, cc_x, cc_m0, cc_B, c_rho0, c_rho_last, c_rk, pc_x, pc_m0, pc_B, p_rho0, p_rk
extern "C" int fvk_cfg_unipc_step(const void* noise_text, const void* noise_uncond, const float* sample, const float* last_sample,
                                  const float* m0, const float* m1, float* x0_out, float* sample_c_out, float* next_out,
                                  void* next_bf16_out, long n, const float* coef_host, int corr_order, int pred_order, void* stream) {
    FVK_CHECK(noise_text && sample && x0_out && sample_c_out && next_out && coef_host && n > 0, FVK_ERR_ARG, "fvk_cfg_unipc_step: null pointer / empty");
    const int recip = (pred_order >> 8) & 1;  // bit 8 of pred_order: accelerator-eager division (multiply by the fp32 reciprocal of the 0-d divisor)
    pred_order &= 0xff;
    FVK_CHECK(corr_order >= 0 && corr_order <= 2 && pred_order >= 1 && pred_order <= 2, FVK_ERR_ARG,
              "fvk_cfg_unipc_step: corrector order %d / predictor order %d unsupported (solver_order <= 2)", corr_order, pred_order);
    FVK_CHECK(corr_order == 0 || (last_sample && m0), FVK_ERR_ARG, "fvk_cfg_unipc_step: corrector needs last_sample and m0");
    FVK_CHECK((corr_order < 2 && pred_order < 2) || m0, FVK_ERR_ARG, "fvk_cfg_unipc_step: order 2 needs the previous converted output");
    FVK_CHECK(corr_order < 2 || m1, FVK_ERR_ARG, "fvk_cfg_unipc_step: corrector order 2 needs two previous converted outputs");
    StepArgs a{};
    a.text = (const bf16_t*)noise_text; a.uncond = (const bf16_t*)noise_uncond; a.sample = sample; a.last_sample = last_sample;
    a.m0 = m0; a.m1 = m1; a.x0_out = x0_out; a.sample_c_out = sample_c_out; a.next_out = next_out; a.next_bf16 = (bf16_t*)next_bf16_out;
    a.n = n;
    const float* c = coef_host;
    a.g = c[0]; a.sigma_t = c[1]; a.cc_x = c[2]; a.cc_m0 = c[3]; a.cc_B = c[4]; a.c_rho0 = c[5]; a.c_rho_last = c[6]; a.c_rk = c[7];
    a.pc_x = c[8]; a.pc_m0 = c[9]; a.pc_B = c[10]; a.p_rho0 = c[11]; a.p_rk = c[12];
    a.corr_order = corr_order; a.pred_order = pred_order;
    a.recip = recip;
    a.c_irk = a.c_rk != 0.f ? 1.0f / a.c_rk : 0.f;
    a.p_irk = a.p_rk != 0.f ? 1.0f / a.p_rk : 0.f;
    const unsigned blocks = (unsigned)((n + 1023) / 1024);
    hipLaunchKernelGGL(cfg_unipc_step_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
    FVK_LAUNCH_CHECK();
    return FVK_OK;
}
