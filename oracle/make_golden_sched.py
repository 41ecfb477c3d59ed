"""tests/golden/unipc.pt: the REAL reference FlowUniPCMultistepScheduler (CPU) driven for 8 steps with seeded bf16 model outputs and
the denoising stage's CFG combine (fastvideo/pipelines/stages/denoising.py:580).  Run in the build container only."""
import os, sys
import torch
HERE = os.path.dirname(os.path.abspath(__file__)); sys.path.insert(0, os.path.dirname(HERE))
from oracle import ref_loader as R
Sched = R.load_unipc_scheduler()
steps, shift, g_scale = 8, 3.0, 5.0
s = Sched(shift=shift)
s.set_timesteps(steps, device="cpu", shift=shift)
gen = torch.Generator().manual_seed(0)
x = torch.randn((1, 16, 3, 8, 10), generator=gen)
lat0 = x.clone()
text, unc, lats = [], [], []
for t in s.timesteps:
    nt, nu = torch.randn(x.shape, generator=gen).bfloat16(), torch.randn(x.shape, generator=gen).bfloat16()
    noise_pred = nu + g_scale * (nt - nu)
    x = s.step(noise_pred, t, x, return_dict=False)[0]
    text.append(nt); unc.append(nu); lats.append(x.clone())
out = os.path.join(os.path.dirname(HERE), "tests", "golden", "unipc.pt")
torch.save({"steps": steps, "shift": shift, "guidance": g_scale, "latents0": lat0, "text": torch.stack(text), "uncond": torch.stack(unc),
            "latents": torch.stack(lats), "timesteps": s.timesteps.clone(), "sigmas": s.sigmas.clone()}, out)
print("wrote", out, os.path.getsize(out) // 1024, "KiB")
