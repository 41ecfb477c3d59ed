"""CPU restatement of one denoising-step tail of the reference: classifier-free-guidance combine + FlowUniPC multistep update.
TEST INFRASTRUCTURE ONLY (import rule as oracle/wan_oracle.py).

Restates
  fastvideo/pipelines/stages/denoising.py:575-596    noise_pred = uncond + g * (text - uncond)  (bf16 tensor arithmetic: every op
                                                     rounds to bf16), then scheduler.step(noise_pred, t, latents_fp32)
  fastvideo/models/schedulers/scheduling_flow_unipc_multistep.py
      :71-133   __init__ (sigmas = shift*s/(1+(shift-1)*s)),  :164-262 set_timesteps (linspace sigmas, shift, final sigma 0)
      :296-347  convert_model_output   x0 = sample - sigma_t * model_output   (flow_prediction, predict_x0; the product of a 0-d fp32
                                       tensor and a bf16 tensor is a bf16 tensor)
      :364-489  multistep_uni_p_bh_update (B(h) = expm1(-h) "bh2", order <= 2: rhos_p = [0.5])
      :491-617  multistep_uni_c_bh_update (order 1: rhos_c = [0.5]; order 2: rhos_c = solve(R, b))
      :649-724  step (corrector when step_index > 0, history shift, order warm-up, lower_order_final)
Scalars are computed exactly as the reference does (0-d fp32 torch tensors on the CPU); tensor updates keep its operation order.
Pinned: tests/test_sched_oracle.py runs it against the REAL scheduler class (oracle/ref_loader.load_unipc_scheduler) and against
tests/golden/unipc.pt."""
from __future__ import annotations

import numpy as np
import torch


def cfg_combine(text: torch.Tensor, uncond: torch.Tensor | None, g: float) -> torch.Tensor:
    """denoising.py:580: tensors keep their (bf16) dtype, so each of the three ops rounds."""
    if uncond is None:
        return text
    return uncond + g * (text - uncond)


class FlowUniPCOracle:

    def __init__(self, num_inference_steps: int, shift: float = 3.0, num_train_timesteps: int = 1000, solver_order: int = 2,
                 lower_order_final: bool = True):
        # __init__: training sigmas with shift 1.0 -> sigma_max / sigma_min;  set_timesteps: linspace + shift + terminal 0
        alphas = np.linspace(1, 1 / num_train_timesteps, num_train_timesteps)[::-1].copy()
        s0 = torch.from_numpy(1.0 - alphas).to(dtype=torch.float32)
        s0 = shift * s0 / (1 + (shift - 1) * s0)  # the constructor already applies `shift` (:105-107); set_timesteps applies it AGAIN
        sigma_min, sigma_max = s0[-1].item(), s0[0].item()  # (:127-128) — the pipeline builds the scheduler with shift=flow_shift
        sig = np.linspace(sigma_max, sigma_min, num_inference_steps + 1).copy()[:-1]
        sig = shift * sig / (1 + (shift - 1) * sig)
        self.timesteps = torch.from_numpy(sig * num_train_timesteps).to(dtype=torch.int64)
        self.sigmas = torch.from_numpy(np.concatenate([sig, [0]]).astype(np.float32))
        self.order, self.lower_order_final = solver_order, lower_order_final
        self.model_outputs = [None] * solver_order
        self.lower_order_nums, self.last_sample, self.step_index, self.this_order = 0, None, 0, 0

    @staticmethod
    def _lam(sigma):
        eps = 1e-12
        return torch.log(torch.clamp(1 - sigma, min=eps)) - torch.log(torch.clamp(sigma, min=eps))

    def _coeffs(self, i_t, i_s0, hist_idx, order, corrector):
        """Scalars of one B(h) update from sigma index i_s0 to i_t.  hist_idx: sigma indices of the older model outputs."""
        sigma_t, sigma_s0 = self.sigmas[i_t], self.sigmas[i_s0]
        alpha_t = 1 - sigma_t
        h = self._lam(sigma_t) - self._lam(sigma_s0)
        rks = [(self._lam(self.sigmas[si]) - self._lam(sigma_s0)) / h for si in hist_idx[:order - 1]]
        rks_t = torch.tensor(rks + [1.0])
        hh = -h
        h_phi_1 = torch.expm1(hh)
        h_phi_k = h_phi_1 / hh - 1
        B_h = torch.expm1(hh)
        R, b, fact = [], [], 1
        for i in range(1, order + 1):
            R.append(torch.pow(rks_t, i - 1))
            b.append(h_phi_k * fact / B_h)
            fact *= i + 1
            h_phi_k = h_phi_k / hh - 1 / fact
        R, b = torch.stack(R), torch.tensor(b)
        if corrector:
            rhos = torch.tensor([0.5]) if order == 1 else torch.linalg.solve(R, b)
        else:
            rhos = torch.tensor([0.5]) if order == 2 else (torch.linalg.solve(R[:-1, :-1], b[:-1]) if order > 2 else None)
        return dict(c_x=sigma_t / sigma_s0, c_m0=alpha_t * h_phi_1, c_B=alpha_t * B_h, rks=rks, rhos=rhos)

    def step(self, model_output: torch.Tensor, sample: torch.Tensor) -> torch.Tensor:
        k = self.step_index
        sigma_t = self.sigmas[k]
        x0 = sample - sigma_t * model_output                                         # convert_model_output
        if k > 0 and self.last_sample is not None:                                    # multistep_uni_c_bh_update
            order = self.this_order
            c = self._coeffs(k, k - 1, [k - (i + 1) for i in range(1, order)], order, corrector=True)
            m0, x = self.model_outputs[-1], self.last_sample
            x_t_ = c["c_x"] * x - c["c_m0"] * m0
            corr = 0
            if order > 1:
                D1s = torch.stack([(self.model_outputs[-(i + 1)] - m0) / c["rks"][i - 1] for i in range(1, order)], dim=1)
                corr = torch.einsum("k,bkc...->bc...", c["rhos"][:-1].to(D1s.device), D1s)  # .to(): a no-op on the CPU; lets the
                #                       same restatement run as torch GPU-eager ops (tests/test_gpu_sched.py, scalar_rounding="fp32")
            sample = (x_t_ - c["c_B"] * (corr + c["rhos"][-1] * (x0 - m0))).to(x.dtype)
        for i in range(self.order - 1):
            self.model_outputs[i] = self.model_outputs[i + 1]
        self.model_outputs[-1] = x0
        this_order = min(self.order, len(self.timesteps) - k) if self.lower_order_final else self.order
        self.this_order = min(this_order, self.lower_order_nums + 1)
        self.last_sample = sample
        order = self.this_order                                                       # multistep_uni_p_bh_update
        c = self._coeffs(k + 1, k, [k - i for i in range(1, order)], order, corrector=False)
        m0 = x0
        x_t = c["c_x"] * sample - c["c_m0"] * m0
        if order > 1:
            D1s = torch.stack([(self.model_outputs[-(i + 1)] - m0) / c["rks"][i - 1] for i in range(1, order)], dim=1)
            x_t = x_t - c["c_B"] * torch.einsum("k,bkc...->bc...", c["rhos"].to(D1s.device), D1s)
        else:
            x_t = x_t - c["c_B"] * 0
        if self.lower_order_nums < self.order:
            self.lower_order_nums += 1
        self.step_index += 1
        return x_t.to(sample.dtype)
