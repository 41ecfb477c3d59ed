"""Generate tests/golden/dmd.pt from the REAL reference: ``FlowMatchEulerDiscreteScheduler(shift=8.0)`` tables, ``pred_noise_to_pred_video``
and ``scheduler.add_noise`` on seeded bf16 / fp32 latents (frames-first [F, C, H, W] as the DMD stages pass them).
Run in the build container only:  ``python oracle/make_golden_dmd.py``."""
from __future__ import annotations

import importlib
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import ref_loader as R  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


def reference():
    R.load_unipc_scheduler()  # installs the diffusers shim
    sch = importlib.import_module("fastvideo.models.schedulers.scheduling_flow_match_euler_discrete").FlowMatchEulerDiscreteScheduler(shift=8.0)
    from fastvideo.models.utils import pred_noise_to_pred_video
    return sch, pred_noise_to_pred_video


def cases():
    g = torch.Generator().manual_seed(11)
    out = []
    for F, shape, ndt, t, tn in [(3, (16, 6, 10), torch.float32, torch.tensor([1000]), torch.tensor([750])),
                                 (3, (16, 6, 10), torch.bfloat16, torch.tensor([750]), torch.tensor([500])),
                                 (4, (16, 5, 7), torch.bfloat16, torch.tensor([937.5, 937.5, 250.0, 3.0]), torch.tensor([833.3, 250.0, 250.0, 0.0])),
                                 (2, (16, 6, 10), torch.bfloat16, torch.tensor([250]), None)]:
        pred = (torch.randn(F, *shape, generator=g) * 1.3).bfloat16()
        noisy = (torch.randn(F, *shape, generator=g) * 1.1).to(ndt)
        noise = torch.randn(F, *shape, generator=g).bfloat16() if tn is not None else None
        out.append(dict(pred=pred, noisy=noisy, noise=noise, t=t, t_next=tn))
    return out


def main():
    sch, p2v = reference()
    rec = []
    for c in cases():
        video = p2v(pred_noise=c["pred"], noise_input_latent=c["noisy"], timestep=c["t"], scheduler=sch)
        nxt = sch.add_noise(video, c["noise"], c["t_next"]) if c["noise"] is not None else None
        rec.append(dict(**c, video=video, next=nxt))
    torch.save(dict(shift=8.0, timesteps=sch.timesteps.clone(), sigmas=sch.sigmas.clone(), cases=rec), os.path.join(OUT, "dmd.pt"))
    print("dmd.pt", os.path.getsize(os.path.join(OUT, "dmd.pt")) / 1e3, "kB")


if __name__ == "__main__":
    main()
