"""Denoising-step tail on MI355X: classifier-free-guidance combine + FlowUniPC multistep scheduler step
(fastvideo/pipelines/stages/denoising.py:575-596 -> FlowUniPCMultistepScheduler.step,
fastvideo/models/schedulers/scheduling_flow_unipc_multistep.py:649-724) as one fused HIP kernel per step (fvk_cfg_unipc_step).

The host computes the step's scalar coefficients exactly as the reference does — on 0-d fp32 torch tensors on the CPU, where the
reference keeps ``self.sigmas`` (``:133``, ``:262``) — and the kernel applies them with the reference's operation order, so the fp32
latents are bit-identical to the eager scheduler (tests/test_gpu_sched.py).  solver_order <= 2, solver_type "bh2", predict_x0,
flow_prediction, lower_order_final: the configuration the Wan pipelines use (fastvideo/pipelines/basic/wan/wan_pipeline.py)."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib, ops

BF16 = torch.bfloat16


class FlowUniPCStepper:

    def __init__(self, num_inference_steps: int, shift: float = 3.0, num_train_timesteps: int = 1000, solver_order: int = 2,
                 lower_order_final: bool = True, scalar_rounding: str = "bf16"):
        # scalar_rounding: how a 0-d fp32 / Python scalar meets a bf16 tensor in `sigma_t * model_output` and `g * (text - uncond)`.
        #   "bf16": the scalar is first cast to the tensors' dtype — what the reference's eager path does when the operands live on the
        #           CPU (TensorIterator common-dtype cast); this is the behaviour the oracle and the golden vectors pin.
        #   "fp32": the scalar stays fp32 inside the kernel, as the eager CUDA/HIP elementwise kernels do with a CPU scalar operand; the
        #           divisions by the 0-d `rk` become multiplications by its fp32 reciprocal, as torch's GPU division kernel does for a CPU
        #           scalar divisor (pinned bit-exactly against torch eager on the device: tests/test_gpu_sched.py).
        # The two differ by at most one bf16 ulp of the product (and one fp32 ulp of the quotient).
        if scalar_rounding not in ("bf16", "fp32"):
            raise ValueError("scalar_rounding must be 'bf16' or 'fp32'")
        self.scalar_rounding = scalar_rounding
        if solver_order not in (1, 2):
            raise ValueError("FlowUniPCStepper: solver_order must be 1 or 2")
        # __init__ (:94-107) with shift, then set_timesteps (:207-232) which shifts again — the reference's behaviour when the
        # pipeline builds the scheduler with shift=flow_shift and calls set_timesteps(num_inference_steps)
        alphas = np.linspace(1, 1 / num_train_timesteps, num_train_timesteps)[::-1].copy()
        s0 = torch.from_numpy(1.0 - alphas).to(dtype=torch.float32)
        s0 = shift * s0 / (1 + (shift - 1) * s0)
        sig = np.linspace(s0[0].item(), s0[-1].item(), num_inference_steps + 1).copy()[:-1]
        sig = shift * sig / (1 + (shift - 1) * sig)
        self.timesteps = torch.from_numpy(sig * num_train_timesteps).to(dtype=torch.int64)
        self.sigmas = torch.from_numpy(np.concatenate([sig, [0]]).astype(np.float32))
        self.order, self.lower_order_final = solver_order, lower_order_final
        self.reset()

    def reset(self):
        self.step_index, self.lower_order_nums, self.this_order = 0, 0, 0
        self._m = [None, None]     # converted model outputs: [older, newest]
        self._last = None
        self._bufs = None

    @staticmethod
    def _lam(sigma):
        return torch.log(torch.clamp(1 - sigma, min=1e-12)) - torch.log(torch.clamp(sigma, min=1e-12))

    def _bh(self, i_t, i_s0, i_hist, order, corrector):
        """(c_x, c_m0, c_B, rho0, rho_last, rk) of a B(h) update (:413-470 / :553-612)."""
        sigma_t, sigma_s0 = self.sigmas[i_t], self.sigmas[i_s0]
        alpha_t = 1 - sigma_t
        h = self._lam(sigma_t) - self._lam(sigma_s0)
        rk = (self._lam(self.sigmas[i_hist]) - self._lam(sigma_s0)) / h if order > 1 else torch.tensor(1.0)
        rks = torch.tensor(([rk] if order > 1 else []) + [1.0])
        hh = -h
        h_phi_1 = torch.expm1(hh)
        h_phi_k = h_phi_1 / hh - 1
        B_h = torch.expm1(hh)
        R, b, fact = [], [], 1
        for i in range(1, order + 1):
            R.append(torch.pow(rks, i - 1))
            b.append(h_phi_k * fact / B_h)
            fact *= i + 1
            h_phi_k = h_phi_k / hh - 1 / fact
        if corrector:
            rhos = torch.tensor([0.5]) if order == 1 else torch.linalg.solve(torch.stack(R), torch.tensor(b))
            rho0, rho_last = (rhos[0] if order > 1 else torch.tensor(0.0)), rhos[-1]
        else:
            rho0, rho_last = torch.tensor(0.5), torch.tensor(0.0)   # order 2: rhos_p = [0.5] (:461-462)
        return [float(v) for v in (sigma_t / sigma_s0, alpha_t * h_phi_1, alpha_t * B_h, rho0, rho_last, rk)]

    @torch.no_grad()
    def step(self, noise_pred_text, latents, noise_pred_uncond=None, guidance_scale: float = 1.0, want_bf16: bool = True):
        """latents fp32 [..]; noise predictions bf16 (the DiT outputs).  Returns (next latents fp32, next latents bf16 or None)."""
        k = self.step_index
        if k >= len(self.timesteps):
            raise RuntimeError("FlowUniPCStepper: stepped past the schedule")
        if latents.dtype != torch.float32 or not latents.is_cuda or noise_pred_text.dtype != BF16:
            raise RuntimeError("FlowUniPCStepper.step: latents must be fp32 and noise predictions bf16 ROCm tensors (no CPU fallback)")
        latents = latents.contiguous()
        n = latents.numel()
        corr_order = self.this_order if (k > 0 and self._last is not None) else 0
        cc = self._bh(k, k - 1, k - 2, corr_order, True) if corr_order else [0.0] * 6
        this_order = min(self.order, len(self.timesteps) - k) if self.lower_order_final else self.order
        self.this_order = min(this_order, self.lower_order_nums + 1)
        pc = self._bh(k + 1, k, k - 1, self.this_order, False)
        rnd = (lambda v: float(torch.tensor(float(v), dtype=torch.float32).bfloat16())) if self.scalar_rounding == "bf16" else float
        coef = (C.c_float * 13)(rnd(guidance_scale), rnd(self.sigmas[k]), cc[0], cc[1], cc[2], cc[3], cc[4], cc[5],
                                pc[0], pc[1], pc[2], pc[3], pc[5])
        x0, sample_c, nxt = (torch.empty_like(latents) for _ in range(3))
        nxt16 = torch.empty(latents.shape, dtype=BF16, device=latents.device) if want_bf16 else None
        p = lambda t: None if t is None else C.c_void_p(t.data_ptr())
        m0, m1 = self._m[1], self._m[0]
        _lib.call("fvk_cfg_unipc_step", p(noise_pred_text.contiguous()), p(None if noise_pred_uncond is None else noise_pred_uncond.contiguous()),
                  p(latents), p(self._last), p(m0), p(m1), p(x0), p(sample_c), p(nxt), p(nxt16), n, coef, int(corr_order),
                  int(self.this_order) | (0x100 if self.scalar_rounding == "fp32" else 0), ops._stream())
        self._m = [m0, x0]
        self._last = sample_c
        if self.lower_order_nums < self.order:
            self.lower_order_nums += 1
        self.step_index += 1
        return nxt, nxt16


class DenoisingLoopHip:
    """The per-step body of the reference's ``DenoisingStage.forward`` (fastvideo/pipelines/stages/denoising.py:372-596) for Wan T2V:
    latents (fp32) -> bf16 model input -> DiT forward(s) (conditional [+ unconditional for classifier-free guidance]) -> fused
    CFG + FlowUniPC step.  ``transformer`` is a WanTransformer3DModelHip; everything on the token axis and the step tail are HIP
    kernels, the loop itself is host control flow."""

    def __init__(self, transformer, num_inference_steps: int, flow_shift: float = 3.0, guidance_scale: float = 1.0,
                 transformer_2=None, boundary_ratio: float | None = None, guidance_scale_2: float | None = None,
                 num_train_timesteps: int = 1000, scalar_rounding: str = "fp32", cfg_batch: bool = False):
        """``scalar_rounding``: "fp32" (default) keeps the 0-d ``sigma_t`` / Python ``guidance_scale`` operands in fp32 inside the step
        kernel — what the reference's eager elementwise kernels do ON THE GPU with a CPU-scalar operand, i.e. what a user of the
        reference on an accelerator gets; "bf16" reproduces the reference's CPU eager path (the scalar is cast to the tensor dtype
        first), which is what ``oracle/sched_oracle.py`` and ``tests/golden/unipc.pt`` pin bit-exactly (``FlowUniPCStepper``'s own
        default, used by the golden tests).  The two differ by at most one bf16 ulp of the product per step.

        ``transformer_2`` / ``boundary_ratio`` / ``guidance_scale_2``: Wan2.2-A14B's two experts — steps with
        t >= boundary_ratio * num_train_timesteps run the high-noise expert with ``guidance_scale``, the rest run ``transformer_2``
        with ``guidance_scale_2`` (denoising.py:251-256, 377-403).  Both experts stay resident (2 x 28 GB of bf16 weights in 288 GB
        of HBM), so the reference's per-boundary CPU offload shuffle has no counterpart here."""
        self.model, self.g = transformer, float(guidance_scale)
        # cfg_batch: the classifier-free-guidance pair as one batch-2 forward (see ``step``); off = the reference's two forwards per step
        # (denoising.py:497-560).  Needs equal-length prompt / negative embeddings (the pipelines pad both to 512 text tokens).
        self.cfg_batch = bool(cfg_batch)
        if self.cfg_batch and any(getattr(m, "quant", None) == "fp8" for m in (transformer, transformer_2) if m is not None):
            raise ValueError("DenoisingLoopHip(cfg_batch=True): a per-tensor fp8 model quantises its activations with ONE absmax over the whole "
                             "[B*S, d] matrix, so the batched pair would share a scale and differ from the reference's two forwards; use "
                             "quantization='fp8_channel' (per-token scales) or cfg_batch=False")
        self.model_2 = transformer_2
        self.g2 = float(guidance_scale if guidance_scale_2 is None else guidance_scale_2)
        self.boundary_timestep = None if boundary_ratio is None else boundary_ratio * num_train_timesteps
        if self.boundary_timestep is not None and transformer_2 is None:
            raise ValueError("DenoisingLoopHip: boundary_ratio given without transformer_2 (the low-noise expert)")
        self.stepper = FlowUniPCStepper(num_inference_steps, shift=flow_shift, num_train_timesteps=num_train_timesteps,
                                        scalar_rounding=scalar_rounding)

    def expert_for(self, t: float):
        """(model, guidance scale) of the step at timestep t (denoising.py:377-403)."""
        if self.boundary_timestep is None or t >= self.boundary_timestep:
            return self.model, self.g
        return self.model_2, self.g2

    @torch.no_grad()
    def step(self, i: int, x, x16, prompt_embeds, negative_prompt_embeds=None):
        """Denoising step i of the schedule: DiT forward(s) on the bf16 latent ``x16`` + the fused CFG / UniPC tail on the fp32 latent ``x``.
        Returns the next (x, x16).  With ``cfg_batch`` the conditional / unconditional pair runs as ONE batch-2 forward (latent repeated,
        embeddings stacked): every kernel of a bf16 / fp8_channel model is row- and batch-independent (per-token scales; the split-KV count of
        an under-filled attention grid is taken from one sample's grid), so each half equals its stand-alone forward bit for bit
        (tests/test_gpu_sched.py), while the pair shares every launch — half the launches, twice the grid per launch.  A per-TENSOR fp8 model
        is refused at construction: its activation scale is one absmax over the whole [B*S, d] matrix, which the pair would share."""
        model, g = self.expert_for(float(self.stepper.timesteps[i]))
        use_cfg = negative_prompt_embeds is not None and self.g > 1.0  # batch-level switch (pipeline_batch_info.py:270-272)
        t = self.stepper.timesteps[i].to(device=x.device, dtype=torch.float32).reshape(1)
        if use_cfg and self.cfg_batch:
            pair = model(x16.expand(2, *x16.shape[1:]).contiguous(), torch.cat([prompt_embeds, negative_prompt_embeds], 0), t.repeat(2))
            cond, uncond = pair[0:1], pair[1:2]
        else:
            cond = model(x16, prompt_embeds, t)
            uncond = model(x16, negative_prompt_embeds, t) if use_cfg else None
        return self.stepper.step(cond, x, uncond, g)

    @torch.no_grad()
    def run(self, latents, prompt_embeds, negative_prompt_embeds=None, num_steps: int | None = None):
        """latents fp32 [1,C,T,H,W]; embeds bf16 [1,L,text_dim].  Returns the denoised latents (fp32)."""
        self.stepper.reset()
        x = latents.float()
        x16 = x.to(BF16)
        n = len(self.stepper.timesteps) if num_steps is None else num_steps
        for i in range(n):
            x, x16 = self.step(i, x, x16, prompt_embeds, negative_prompt_embeds)
        return x


class FlowMatchEulerTables:
    """``FlowMatchEulerDiscreteScheduler.__init__`` tables (scheduling_flow_match_euler_discrete.py:140-158): fp32 ``timesteps`` /
    ``sigmas`` over the 1000 training timesteps with the static shift — what the DMD stages look sigmas up in (they build the scheduler
    with shift=8.0 and never call set_timesteps: denoising.py:1257)."""

    def __init__(self, shift: float = 8.0, num_train_timesteps: int = 1000):
        t = np.linspace(1, num_train_timesteps, num_train_timesteps, dtype=np.float32)[::-1].copy()
        t = torch.from_numpy(t).to(dtype=torch.float32)
        sig = t / num_train_timesteps
        sig = shift * sig / (1 + (shift - 1) * sig)
        self.timesteps = sig * num_train_timesteps
        self.sigmas = sig
        self.num_train_timesteps = num_train_timesteps
        self._t64 = self.timesteps.double()

    def index_of(self, timestep) -> torch.Tensor:
        """``argmin(|timesteps - t|)`` in fp64 per entry of ``timestep`` (utils.py:170-172; add_noise does it in the timestep's dtype,
        :632 — identical for the integer / table-valued timesteps the stages pass)."""
        t = torch.as_tensor(timestep).reshape(-1).double().cpu()
        return torch.argmin((self._t64.unsqueeze(0) - t.unsqueeze(1)).abs(), dim=1)

    def warp(self, steps) -> torch.Tensor:
        """``warp_denoising_step`` (causal_denoising.py:81-83): integer DMD steps -> the shifted schedule's timesteps."""
        table = torch.cat((self.timesteps, torch.tensor([0], dtype=torch.float32)))
        return table[self.num_train_timesteps - torch.as_tensor(steps, dtype=torch.long)]


class DmdStepper:
    """One DMD sampling step on the GPU (fvk_dmd_step): ``pred_noise_to_pred_video`` (fastvideo/models/utils.py:138-175) and, unless it
    is the last step, ``scheduler.add_noise(pred_video, noise, next_timestep)`` (scheduling_flow_match_euler_discrete.py:601-635) —
    the step tail of ``DmdDenoisingStage`` (denoising.py:1382-1395) and ``CausalDMDDenosingStage`` (causal_denoising.py:273-310).
    Bit-identical to the eager reference ops (tests/test_gpu_sched.py)."""

    def __init__(self, shift: float = 8.0, num_train_timesteps: int = 1000):
        self.tables = FlowMatchEulerTables(shift, num_train_timesteps)

    def step(self, pred_noise, noisy_latent, timestep, noise=None, next_timestep=None):
        """pred_noise bf16 [F, ...] (frames first, as the stages' ``flatten(0, 1)`` of [B, T, C, H, W]); noisy_latent same shape, bf16 or fp32;
        timestep: scalar or [F].  Returns (pred_video bf16, next_latent bf16 or None)."""
        if pred_noise.device.type != "cuda":
            raise RuntimeError("DmdStepper runs on a ROCm device only (no CPU fallback)")
        if pred_noise.dtype != BF16 or noisy_latent.dtype not in (BF16, torch.float32) or noisy_latent.shape != pred_noise.shape:
            raise ValueError("DmdStepper.step: pred_noise bf16 and noisy_latent (bf16 | fp32) of one shape expected")
        dev = pred_noise.device
        pred_noise, noisy_latent = pred_noise.contiguous(), noisy_latent.contiguous()
        frames = pred_noise.shape[0]
        per_frame = pred_noise.numel() // frames
        expand = lambda idx: idx.expand(frames) if idx.numel() == 1 else idx
        it = expand(self.tables.index_of(timestep))
        if it.numel() != frames:
            raise ValueError(f"timestep has {it.numel()} entries for {frames} frames")
        sigma_t = self.tables.sigmas[it].double().to(dev)
        video = torch.empty_like(pred_noise)
        nxt = sn = None
        if noise is not None:
            if next_timestep is None or noise.dtype != BF16 or noise.shape != pred_noise.shape:
                raise ValueError("DmdStepper.step: bf16 noise of the latent's shape and next_timestep come together")
            noise = noise.to(dev).contiguous()
            sn = self.tables.sigmas[expand(self.tables.index_of(next_timestep))].to(dev).contiguous()
            nxt = torch.empty_like(pred_noise)
        _lib.call("fvk_dmd_step", ops._p(pred_noise), ops._p(noisy_latent), int(noisy_latent.dtype == torch.float32), ops._p(sigma_t),
                  ops._p(noise), ops._p(sn), ops._p(video), ops._p(nxt), frames, per_frame, ops._stream())
        return video, nxt


class DmdDenoisingLoopHip:
    """Few-step DMD sampling loop of ``DmdDenoisingStage.forward`` (fastvideo/pipelines/stages/denoising.py:1322-1401; the FastWan
    pipelines): per step one DiT forward on the bf16 latent, ``pred_noise_to_pred_video``, and — except after the last step — re-noising
    to the next timestep with a fresh draw.  The step tail is one HIP kernel (``DmdStepper``), bit-identical to the eager ops; the latent
    stays on the GPU.  ``noise_fn(shape_btchw, dtype) -> tensor`` supplies the draws (the reference: ``torch.randn(latents.shape,
    dtype=pred_video.dtype, generator=batch.generator[0])`` moved to the device, ``:1388-1391``)."""

    def __init__(self, transformer, dmd_denoising_steps, flow_shift: float = 8.0):
        self.model = transformer
        self.stepper = DmdStepper(flow_shift)
        self.timesteps = torch.tensor(list(dmd_denoising_steps), dtype=torch.long)

    @torch.no_grad()
    def run(self, latents, prompt_embeds, noise_fn):
        """latents [1, C, T, H, W] (initial noise, fp32 or bf16) -> denoised latents [1, C, T, H, W] bf16 (the dtype the reference's loop ends
        in: ``pred_video`` takes the model output's dtype)."""
        dev = self.model.device
        B, C, T, H, W = latents.shape
        if B != 1:
            raise ValueError("DmdDenoisingLoopHip: one sample per call")
        prompt_embeds = prompt_embeds.to(dev)
        cur = latents.to(dev).permute(0, 2, 1, 3, 4).flatten(0, 1).contiguous()  # frames first [T, C, H, W], as the stage's flatten(0, 1)
        n = len(self.timesteps)
        for i in range(n):
            t = self.timesteps[i]
            x_in = cur.view(1, T, C, H, W).permute(0, 2, 1, 3, 4).to(BF16)
            pred = self.model(x_in, prompt_embeds, t.reshape(1).float().to(dev))
            pred_f = pred.permute(0, 2, 1, 3, 4).flatten(0, 1).contiguous()
            if i < n - 1:
                noise = noise_fn((1, T, C, H, W), BF16).to(dev).flatten(0, 1)
                _, cur = self.stepper.step(pred_f, cur, t, noise, self.timesteps[i + 1])
            else:
                cur, _ = self.stepper.step(pred_f, cur, t)
        return cur.view(1, T, C, H, W).permute(0, 2, 1, 3, 4).contiguous()
