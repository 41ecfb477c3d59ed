"""Pins oracle/dmd_oracle.py and the host tables of fastvideo_amd.scheduler (FlowMatchEulerTables) against the REAL reference:
tests/golden/dmd.pt everywhere, the live FlowMatchEulerDiscreteScheduler / pred_noise_to_pred_video when /root/reference exists."""
import os

import pytest
import torch

from oracle import dmd_oracle as D
from oracle import ref_loader as R


@pytest.fixture(scope="module")
def fx(golden_dir):
    return torch.load(os.path.join(golden_dir, "dmd.pt"), weights_only=False)


def test_tables_bit_exact(fx):
    ts, sg = D.tables(fx["shift"])
    assert torch.equal(ts, fx["timesteps"]) and torch.equal(sg, fx["sigmas"])
    from fastvideo_amd.scheduler import FlowMatchEulerTables
    T = FlowMatchEulerTables(fx["shift"])
    assert torch.equal(T.timesteps, fx["timesteps"]) and torch.equal(T.sigmas, fx["sigmas"])
    assert T.index_of(torch.tensor([1000, 750, 3, 0])).tolist() == torch.argmin(
        (fx["timesteps"].double().unsqueeze(0) - torch.tensor([1000., 750., 3., 0.]).double().unsqueeze(1)).abs(), dim=1).tolist()
    # warp_denoising_step (causal_denoising.py:81-83)
    table = torch.cat((fx["timesteps"], torch.tensor([0.0])))
    assert torch.equal(T.warp([1000, 750, 500, 250]), table[1000 - torch.tensor([1000, 750, 500, 250])])


def test_step_functions_bit_exact_vs_golden(fx):
    for c in fx["cases"]:
        v = D.pred_noise_to_pred_video(c["pred"], c["noisy"], c["t"], fx["timesteps"], fx["sigmas"])
        assert v.dtype == torch.bfloat16 and torch.equal(v, c["video"])
        if c["noise"] is not None:
            n = D.add_noise(v, c["noise"], c["t_next"], fx["timesteps"], fx["sigmas"])
            assert n.dtype == torch.bfloat16 and torch.equal(n, c["next"])


@pytest.mark.skipif(not R.available(), reason="needs the reference checkout (/root/reference)")
def test_live_reference(fx):
    from oracle.make_golden_dmd import reference
    sch, p2v = reference()
    ts, sg = D.tables(8.0)
    assert torch.equal(ts, sch.timesteps) and torch.equal(sg, sch.sigmas)
    g = torch.Generator().manual_seed(5)
    pred, noisy, noise = (torch.randn(5, 16, 4, 6, generator=g).bfloat16() for _ in range(3))
    t, tn = torch.tensor([612.0, 1000.0, 40.0, 999.9, 0.0]), torch.tensor([500])
    v = p2v(pred_noise=pred, noise_input_latent=noisy.float(), timestep=t, scheduler=sch)
    assert torch.equal(v, D.pred_noise_to_pred_video(pred, noisy.float(), t, ts, sg))
    assert torch.equal(sch.add_noise(v, noise, tn), D.add_noise(v, noise, tn, ts, sg))
