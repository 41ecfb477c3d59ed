"""TEST INFRASTRUCTURE — CPU restatement of the DMD few-step sampling arithmetic: the FlowMatchEuler tables
(fastvideo/models/schedulers/scheduling_flow_match_euler_discrete.py:140-158), ``pred_noise_to_pred_video`` (fastvideo/models/utils.py:138-175)
and ``FlowMatchEulerDiscreteScheduler.add_noise`` (:601-635).  Pinned bit-exactly against the real reference functions
(tests/test_dmd_oracle.py: live import through oracle/_diffusers_shim.py, and tests/golden/dmd.pt from oracle/make_golden_dmd.py)."""
from __future__ import annotations

import numpy as np
import torch


def tables(shift: float = 8.0, num_train_timesteps: int = 1000):
    t = np.linspace(1, num_train_timesteps, num_train_timesteps, dtype=np.float32)[::-1].copy()
    t = torch.from_numpy(t).to(dtype=torch.float32)
    sig = t / num_train_timesteps
    sig = shift * sig / (1 + (shift - 1) * sig)
    return sig * num_train_timesteps, sig  # timesteps, sigmas (fp32)


def _expand(timestep, n):
    if timestep.ndim == 2:
        timestep = timestep.flatten(0, 1)
    if timestep.ndim == 1 and timestep.shape[0] == 1:
        timestep = timestep.expand(n)
    assert timestep.numel() == n
    return timestep


def pred_noise_to_pred_video(pred_noise, noise_input_latent, timestep, timesteps, sigmas):
    """utils.py:138-175: everything in float64, index = argmin |timesteps - t|, cast back to pred_noise's dtype."""
    timestep = _expand(timestep, noise_input_latent.shape[0])
    dtype = pred_noise.dtype
    tid = torch.argmin((timesteps.double().unsqueeze(0) - timestep.unsqueeze(1)).abs(), dim=1)
    sigma_t = sigmas.double()[tid].reshape(-1, 1, 1, 1)
    return (noise_input_latent.double() - sigma_t * pred_noise.double()).to(dtype)


def add_noise(clean_latent, noise, timestep, timesteps, sigmas):
    """scheduling_flow_match_euler_discrete.py:601-635: fp32 sigma [B,1,1,1] against the latents' dtype, ``type_as(noise)``."""
    timestep = _expand(timestep, clean_latent.shape[0])
    tid = torch.argmin((timesteps.unsqueeze(0) - timestep.unsqueeze(1)).abs(), dim=1)
    sigma = sigmas[tid].reshape(-1, 1, 1, 1)
    return ((1 - sigma) * clean_latent + sigma * noise).type_as(noise)


def dmd_rollout(model_fn, latents_bcthw, dmd_steps, noise_list, shift: float = 8.0):
    """The loop of ``DmdDenoisingStage.forward`` (denoising.py:1322-1401) around ``model_fn(latent_bcthw_bf16, timestep[1]) -> pred_bcthw``
    (e.g. the oracle DiT) with the re-noising draws given in call order ([1, T, C, H, W] bf16 each).  Returns [1, C, T, H, W]."""
    ts_tab, sg_tab = tables(shift)
    timesteps = torch.tensor(list(dmd_steps), dtype=torch.long)
    latents = latents_bcthw.permute(0, 2, 1, 3, 4)  # the stage keeps [B, T, C, H, W]
    it = iter(noise_list)
    for i, t in enumerate(timesteps):
        noise_latents = latents.clone()
        t_expand = t.repeat(1)
        pred = model_fn(latents.to(torch.bfloat16).permute(0, 2, 1, 3, 4), t_expand).permute(0, 2, 1, 3, 4)
        video = pred_noise_to_pred_video(pred.flatten(0, 1), noise_latents.flatten(0, 1), t_expand, ts_tab, sg_tab).unflatten(0, pred.shape[:2])
        if i < len(timesteps) - 1:
            nxt = timesteps[i + 1] * torch.ones([1], dtype=torch.long)
            latents = add_noise(video.flatten(0, 1), next(it).flatten(0, 1), nxt, ts_tab, sg_tab).unflatten(0, video.shape[:2])
        else:
            latents = video
    return latents.permute(0, 2, 1, 3, 4)
