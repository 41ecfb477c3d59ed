"""Full-geometry parity of the kernel CHAIN: one Wan2.1-T2V-1.3B-geometry transformer block (12 heads x 128, ffn 8960, text 512 x 4096,
fused QKV stride 4608, 12-head RMS norm across 1536, cross-attention with 512 keys) inside a 1-layer model, at BASELINE.json's cfg1
(latent [1,16,9,64,64] -> S = 9 216) and cfg2 (latent [1,16,21,60,104] -> S = 32 760) — HIP path vs ``oracle.WanOracle`` (the
restatement pinned bit-exactly to the real reference at small geometry, tests/test_oracle_golden.py) on the same seeded inputs.

Every other model-level test runs 2 heads / d = 256; per-op tests cover full sizes one kernel at a time.  This one closes the gap
between them: the whole chain at the real strides and reduction lengths.

Tolerance: the reference's own DiT bound atol = 1e-1, rtol = 1e-2 (fastvideo/tests/transformers/test_wanvideo.py:109) on every
compared tensor (residual stream after self-attention, block output, final norm, model output) plus mean-error bounds an order of
magnitude tighter.  CPU cost of the oracle: one block at S = 32 760 ~ 10 s on the GPU box's host cores."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def _cmp(y, ref, what, atol=1e-1, rtol=1e-2, mean_tol=1.5e-2):
    y, ref = y.float().cpu(), ref.float()
    assert torch.isfinite(y).all(), what
    err = (y - ref).abs()
    bad = err > atol + rtol * ref.abs()
    print(f"{what}: max|err|={err.max().item():.4g} mean|err|={err.mean().item():.4g} ref_absmean={ref.abs().mean().item():.4g}")
    assert not bad.any(), f"{what}: {int(bad.sum())}/{bad.numel()} outside tolerance, max {err.max().item():.4g}"
    assert err.mean().item() < mean_tol, f"{what}: mean error {err.mean().item():.4g}"


@pytest.fixture(scope="module")
def block_1_3b():
    from fastvideo_amd import wan_config as WC
    cfg = WC.WanConfig("Wan2.1-T2V-1.3B geometry, 1 layer", 12, 128, 8960, 1)
    sd = WC.random_state_dict(cfg, seed=0, device="cpu")
    # random_state_dict draws small biases / tables; widen the AdaLN tables so that shift / scale / gate matter
    gen = torch.Generator().manual_seed(5)
    for k in ("blocks.0.scale_shift_table", "scale_shift_table"):
        sd[k] = (torch.randn(sd[k].shape, generator=gen) * 0.3).to(sd[k].dtype)
    return cfg, sd


@pytest.mark.parametrize("name,latent_shape", [("cfg1", (1, 16, 9, 64, 64)), ("cfg2", (1, 16, 21, 60, 104))])
def test_one_block_at_full_geometry_matches_oracle(block_1_3b, name, latent_shape):
    from fastvideo_amd.wan_dit import WanTransformer3DModelHip
    from oracle import wan_oracle as W
    cfg, sd = block_1_3b
    gen = torch.Generator().manual_seed(1)
    latent = torch.randn(latent_shape, generator=gen).bfloat16()
    ctx = torch.randn((1, 512, cfg.text_dim), generator=gen).bfloat16()
    t = torch.tensor([500.0])
    torch.set_num_threads(min(64, os.cpu_count() or 8))  # torch's CPU GEMM / SDPA stop scaling well before 256 threads
    orc = W.WanOracle(sd, num_heads=cfg.num_heads)
    tr_ref = {}
    with torch.no_grad():
        y_ref = orc.forward(latent, ctx, t, trace=tr_ref)
    model = WanTransformer3DModelHip(sd, cfg.num_heads, cfg.head_dim, cfg.patch_size, cfg.eps, cfg.freq_dim)
    tr = {}
    y = model(latent.cuda(), ctx.cuda(), t.cuda(), trace=tr)
    for key in ("blocks.0.after_self_attn", "blocks.0.out", "norm_out"):
        _cmp(tr[key], tr_ref[key], f"{name} {key}")
    _cmp(y, y_ref, f"{name} model output")
    assert y.shape == latent.shape and y.dtype == torch.bfloat16


def test_one_block_at_a14b_geometry_matches_oracle():
    """Wan2.2-T2V-A14B block geometry (BASELINE cfg4: 40 heads x 128 = 5120, ffn 13 824, text 512 x 4096: QKV row stride 15 360, RMS norm
    across 5120 columns, 54 / 20 column tiles per GEMM) in a 1-layer model on a reduced latent [1,16,5,32,32] (S = 1 280) vs the oracle."""
    from fastvideo_amd import wan_config as WC
    from fastvideo_amd.wan_dit import WanTransformer3DModelHip
    from oracle import wan_oracle as W
    cfg = WC.WanConfig("Wan2.2-A14B geometry, 1 layer", 40, 128, 13824, 1)
    sd = WC.random_state_dict(cfg, seed=3, device="cpu")
    gen = torch.Generator().manual_seed(7)
    for k in ("blocks.0.scale_shift_table", "scale_shift_table"):
        sd[k] = (torch.randn(sd[k].shape, generator=gen) * 0.3).to(sd[k].dtype)
    latent = torch.randn((1, 16, 5, 32, 32), generator=gen).bfloat16()
    ctx = torch.randn((1, 512, cfg.text_dim), generator=gen).bfloat16()
    t = torch.tensor([321.0])
    torch.set_num_threads(min(64, os.cpu_count() or 8))
    tr_ref, tr = {}, {}
    with torch.no_grad():
        y_ref = W.WanOracle(sd, num_heads=cfg.num_heads).forward(latent, ctx, t, trace=tr_ref)
    model = WanTransformer3DModelHip(sd, cfg.num_heads, cfg.head_dim, cfg.patch_size, cfg.eps, cfg.freq_dim)
    y = model(latent.cuda(), ctx.cuda(), t.cuda(), trace=tr)
    for key in ("blocks.0.after_self_attn", "blocks.0.out", "norm_out"):
        _cmp(tr[key], tr_ref[key], f"A14B geometry {key}")
    _cmp(y, y_ref, "A14B geometry model output")


def test_thirty_layers_at_1_3b_geometry_match_oracle():
    """The WHOLE Wan2.1-T2V-1.3B stack (30 blocks, 12 heads x 128, ffn 8960) at BASELINE cfg1 (latent [1,16,9,64,64], S = 9 216) vs
    ``WanOracle.forward`` — the reference's own DiT test runs the whole model too (fastvideo/tests/transformers/test_wanvideo.py:29-109).
    The one-block tests above check every kernel at the real strides; this one checks that 30 layers of bf16 rounding do not DRIFT at
    d = 1536: the residual stream is compared every 5 layers (error relative to the stream's own magnitude, which grows with depth) and
    the model output against the reference's DiT bound (atol 1e-1, rtol 1e-2: at most 1e-4 of the elements outside, none by more than
    3x) plus a mean bound.  CPU cost of the oracle: ~1.5 s per block on the GPU box's host cores."""
    from fastvideo_amd import wan_config as WC
    from fastvideo_amd.wan_dit import WanTransformer3DModelHip
    from oracle import wan_oracle as W
    cfg = WC.WAN21_T2V_1_3B
    assert cfg.num_layers == 30 and cfg.num_heads == 12 and cfg.ffn_dim == 8960
    sd = WC.random_state_dict(cfg, seed=0, device="cpu")
    gen = torch.Generator().manual_seed(11)
    for k in [k for k in sd if k.endswith("scale_shift_table")]:
        sd[k] = (torch.randn(sd[k].shape, generator=gen) * 0.3).to(sd[k].dtype)
    latent = torch.randn((1, 16, 9, 64, 64), generator=gen).bfloat16()
    ctx = torch.randn((1, 512, cfg.text_dim), generator=gen).bfloat16()
    t = torch.tensor([700.0])
    torch.set_num_threads(min(64, os.cpu_count() or 8))
    tr_ref, tr = {}, {}
    with torch.no_grad():
        y_ref = W.WanOracle(sd, num_heads=cfg.num_heads).forward(latent, ctx, t, trace=tr_ref)
    model = WanTransformer3DModelHip(sd, cfg.num_heads, cfg.head_dim, cfg.patch_size, cfg.eps, cfg.freq_dim)
    y = model(latent.cuda(), ctx.cuda(), t.cuda(), trace=tr)
    rels = []
    for i in (0, 4, 9, 14, 19, 24, 29):
        a, b = tr[f"blocks.{i}.out"].float().cpu(), tr_ref[f"blocks.{i}.out"].float()
        assert torch.isfinite(a).all()
        err = (a - b).abs()
        rel = err.mean().item() / b.abs().mean().item()
        rels.append(rel)
        print(f"30 layers, residual stream after block {i}: mean|err| / mean|ref| = {rel:.4g} (mean|ref| {b.abs().mean().item():.4g}, max|err| {err.max().item():.4g})")
        assert rel < 3e-2, f"block {i}: relative mean error {rel:.4g}"
    assert rels[-1] < 6 * max(rels[0], 2e-3), f"error grew from {rels[0]:.3g} (block 0) to {rels[-1]:.3g} (block 29): drift"
    yc, ref = y.float().cpu(), y_ref.float()
    err = (yc - ref).abs()
    lim = 1e-1 + 1e-2 * ref.abs()
    frac_out = (err > lim).float().mean().item()
    print(f"30 layers, model output: max|err|={err.max().item():.4g} mean|err|={err.mean().item():.4g} ref_absmean={ref.abs().mean().item():.4g} "
          f"outside the DiT bound: {frac_out:.3g}")
    # the reference's own DiT bound, un-softened (fastvideo/tests/transformers/test_wanvideo.py:109 allows NO element outside atol 1e-1 /
    # rtol 1e-2); rounds 1-3 allowed 1e-4 of the elements up to 3x the limit here — never needed: measured max 6.3e-2
    assert frac_out == 0.0, f"{int((err > lim).sum())} elements outside the reference's DiT bound"
    assert err.mean().item() < 2e-2


def test_full_contract_vs_reference_fixture(golden_dir):
    """THE contract workload end to end: Wan2.1-T2V-1.3B, 30 layers, latent [1,16,21,60,104] = 32 760 tokens, 512 text tokens — the HIP
    forward against the REAL reference's CPU forward (bf16 weights under bf16 autocast, SDPA backend) of the same seeded weights and
    inputs, stored in tests/golden/contract_full_ref.pt by ``scripts/full_contract_parity.py reference`` (run in the build container, log in
    profiles/r04_full_contract_reference.log).  Every one of the 2 096 640 output elements must be inside the reference's DiT bound (atol 1e-1,
    rtol 1e-2), for the default long-key attention kernel AND the other two selectable ones (attn_autotune may keep either of the first
    two: whichever it keeps has passed this)."""
    import importlib.util
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    fix = os.path.join(golden_dir, "contract_full_ref.pt")
    spec = importlib.util.spec_from_file_location("full_contract_parity", os.path.join(os.path.dirname(golden_dir), "..", "scripts", "full_contract_parity.py"))
    fcp = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fcp)
    out_json = os.path.join(os.path.dirname(golden_dir), "..", "gpurun_out", "full_contract_parity.json")
    res = fcp.leg_hip(fix, out_json)
    for name, r in res["kernels"].items():
        print(name, r)
        assert r["elements"] == 16 * 21 * 60 * 104
        assert r["outside_dit_bound_atol1e-1_rtol1e-2"] == 0, f"{name}: {r}"
        assert r["mean_err"] < 2e-2 and r["worst_ratio_to_bound"] < 1.0, f"{name}: {r}"
