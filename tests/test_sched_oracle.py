"""Pins oracle/sched_oracle.py (CFG combine + FlowUniPC step) against the REAL reference scheduler: golden fixture everywhere, live class
when /root/reference is present."""
import os
import pytest
import torch
from oracle import ref_loader as R
from oracle.sched_oracle import FlowUniPCOracle, cfg_combine

GOLD = os.path.join(os.path.dirname(__file__), "golden", "unipc.pt")


def _run(g):
    o = FlowUniPCOracle(g["steps"], shift=g["shift"])
    assert torch.equal(o.timesteps, g["timesteps"]) and torch.equal(o.sigmas, g["sigmas"])
    x, outs = g["latents0"].clone(), []
    for i in range(g["steps"]):
        x = o.step(cfg_combine(g["text"][i], g["uncond"][i], g["guidance"]), x)
        outs.append(x)
    return torch.stack(outs)


def test_oracle_matches_reference_golden_bit_exact():
    g = torch.load(GOLD)
    assert torch.equal(_run(g), g["latents"])


@pytest.mark.skipif(not R.available(), reason="needs the reference checkout")
@pytest.mark.parametrize("steps,shift", [(4, 5.0), (11, 8.0), (3, 1.0)])
def test_oracle_matches_live_reference(steps, shift):
    Sched = R.load_unipc_scheduler()
    s = Sched(shift=shift)
    s.set_timesteps(steps, device="cpu", shift=shift)
    o = FlowUniPCOracle(steps, shift=shift)
    gen = torch.Generator().manual_seed(steps)
    x = torch.randn((1, 4, 2, 6, 6), generator=gen)
    xo = x.clone()
    for t in s.timesteps:
        mo = torch.randn(x.shape, generator=gen).bfloat16()
        x = s.step(mo, t, x, return_dict=False)[0]
        xo = o.step(mo, xo)
        assert torch.equal(x, xo)


def test_denoising_loop_expert_selection_is_host_logic_only():
    """DenoisingLoopHip.expert_for: t >= boundary_ratio * num_train_timesteps -> (transformer, guidance_scale), else
    (transformer_2, guidance_scale_2) — denoising.py:251-256, 377-403.  No kernels involved."""
    from fastvideo_amd.scheduler import DenoisingLoopHip
    a, b = object(), object()
    loop = DenoisingLoopHip(a, 40, flow_shift=12.0, guidance_scale=4.0, transformer_2=b, boundary_ratio=0.875, guidance_scale_2=3.0)
    assert loop.boundary_timestep == 875.0
    picks = [loop.expert_for(float(t))[0] for t in loop.stepper.timesteps]
    n_hi = sum(1 for p in picks if p is a)
    assert 0 < n_hi < 40 and all(p is a for p in picks[:n_hi]) and all(p is b for p in picks[n_hi:])
    assert loop.expert_for(875.0) == (a, 4.0) and loop.expert_for(874.0) == (b, 3.0)
    assert DenoisingLoopHip(a, 4).expert_for(1.0) == (a, 1.0)
    with pytest.raises(ValueError):
        DenoisingLoopHip(a, 4, boundary_ratio=0.9)
