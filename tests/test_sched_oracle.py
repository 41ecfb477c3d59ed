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
