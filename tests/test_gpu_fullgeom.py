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
    # round 5: the V projection as its own GEMM that writes V^T (ops.gemm_vt) is the shipped dense path; the fused QKV GEMM + V^T layout pass
    # it replaced computes the same bytes
    assert model.vt_gemm
    model.vt_gemm = False
    assert torch.equal(model(latent.cuda(), ctx.cuda(), t.cuda()), y), "V^T-GEMM path differs from fused QKV + v_transpose"
    model.vt_gemm = True
    # ... and so does the cross-attention residual add moved from the norm pass into the out-projection's epilogue
    model.fuse_cross_residual = False
    assert torch.equal(model(latent.cuda(), ctx.cuda(), t.cuda()), y), "cross-attention residual in the GEMM epilogue differs from the norm-pass form"
    model.fuse_cross_residual = True
    pair = model(latent.cuda().expand(2, -1, -1, -1, -1).contiguous(), ctx.cuda().expand(2, -1, -1).contiguous(), t.cuda().repeat(2))
    assert torch.equal(pair[0:1], y) and torch.equal(pair[1:2], y), "batch-2 forward (V^T GEMM batched over samples) differs from the single forward"


def _cfg2_inputs(cfg, seed=1):
    gen = torch.Generator().manual_seed(seed)
    latent = torch.randn((1, 16, 21, 60, 104), generator=gen).bfloat16()
    ctx = torch.randn((1, 512, cfg.text_dim), generator=gen).bfloat16()
    return latent, ctx, torch.tensor([500.0])


def _cmp_fp8(y, ref_q, ref_b16, what):
    """The bound of a dynamically quantised forward, DERIVED from the quantisation itself (as the VAE bound is derived from the reference's own
    bf16 error): fp8 activations are re-quantised from bf16 values that differ between two correct implementations by accumulation order
    — a value on the other side of an e4m3 rounding boundary moves by a whole fp8 ulp (6-12 %), and with per-tensor scales ONE bf16 ulp in
    the tensor's absmax moves every boundary at once — so element-wise the device cannot sit inside the bf16 DiT bound against ANY fp8
    oracle.  What it must do: stay closer to the fp8 oracle than the fp8 oracle is to the bf16 oracle, in the mean, in the fraction of
    elements outside the reference's DiT bound (atol 1e-1, rtol 1e-2) and in the maximum."""
    y, ref_q, ref_b16 = y.float().cpu(), ref_q.float(), ref_b16.float()
    assert torch.isfinite(y).all(), what
    err, dq = (y - ref_q).abs(), (ref_q - ref_b16).abs()
    f_err = (err > 1e-1 + 1e-2 * ref_q.abs()).float().mean().item()
    f_dq = (dq > 1e-1 + 1e-2 * ref_b16.abs()).float().mean().item()
    print(f"{what}: device vs fp8 oracle mean {err.mean().item():.4g} max {err.max().item():.4g} outside the DiT bound {f_err:.3g} | "
          f"fp8 oracle vs bf16 oracle mean {dq.mean().item():.4g} max {dq.max().item():.4g} outside {f_dq:.3g} | ratio of means {err.mean().item() / dq.mean().item():.3f}")
    assert err.mean().item() <= dq.mean().item(), f"{what}: implementation noise {err.mean().item():.4g} above the quantisation's own effect {dq.mean().item():.4g}"
    assert f_err <= f_dq + 1e-6, f"{what}: {f_err:.3g} of the elements outside the DiT bound (the quantisation itself: {f_dq:.3g})"
    assert err.max().item() <= max(dq.max().item(), 1e-1), f"{what}: max error {err.max().item():.4g} (quantisation effect max {dq.max().item():.4g})"


@pytest.mark.parametrize("quant", ["fp8", "fp8_channel"])
def test_one_block_at_cfg2_fp8_matches_oracle(block_1_3b, quant):
    """BASELINE config 5's linears (fp8 e4m3, tensor / per-token granularity: fp8_config.py:55-68,119-157) at the REAL strides: one
    12-head x 128 block, 32 760 tokens, K = 1536 / 8960, fused QKV with one scale per original matrix, the LayerNorm pass that writes the
    e4m3 row itself (fp8_channel) — vs ``WanOracle(quantization=...)`` (reference quantisers, _scaled_mm restated, pinned bit-exact at
    small geometry).  The sub-block before any re-quantised activation matters (after self-attention) meets the bf16 DiT bound outright;
    from there on the bound is _cmp_fp8's (measured in round 5: device-vs-oracle mean error 0.27 x the quantisation
    effect for per-token scales, 0.63 x for per-tensor scales)."""
    from fastvideo_amd.wan_dit import WanTransformer3DModelHip
    from oracle import wan_oracle as W
    cfg, sd = block_1_3b
    latent, ctx, t = _cfg2_inputs(cfg)
    torch.set_num_threads(min(64, os.cpu_count() or 8))
    tr_ref, tr_b16, tr = {}, {}, {}
    with torch.no_grad():
        y_ref = W.WanOracle(sd, num_heads=cfg.num_heads, quantization=quant).forward(latent, ctx, t, trace=tr_ref)
        y_b16 = W.WanOracle(sd, num_heads=cfg.num_heads).forward(latent, ctx, t, trace=tr_b16)
    model = WanTransformer3DModelHip(sd, cfg.num_heads, cfg.head_dim, cfg.patch_size, cfg.eps, cfg.freq_dim, quantization=quant)
    y = model(latent.cuda(), ctx.cuda(), t.cuda(), trace=tr)
    _cmp(tr["blocks.0.after_self_attn"], tr_ref["blocks.0.after_self_attn"], f"cfg2 {quant} blocks.0.after_self_attn")
    for key in ("blocks.0.after_self_attn", "blocks.0.out", "norm_out"):
        _cmp_fp8(tr[key], tr_ref[key], tr_b16[key], f"cfg2 {quant} {key}")
    _cmp_fp8(y, y_ref, y_b16, f"cfg2 {quant} model output")


def test_one_block_at_cfg2_sta_matches_oracle(block_1_3b):
    """BASELINE config 3 at its real geometry: sliding-tile attention, window (3,3,3) on tile (6,8,8), token grid (21,30,52) = 4 x 4 x 7
    ragged tiles, 12 heads — the shipped path (_sta_fused: q / k scattered by the norm pass through row maps at stride 4608, V^T
    gathered, grouped 256-row KV block lists, output rows scattered) vs the oracle with exact fp32 attention under the sliding-tile
    mask (oracle.vsa_oracle.sta_attention_ragged == attention_fp32_ref + sta_mask_ragged, tests/test_oracle_chunked.py; window rule of
    fastvideo-kernel/tests/support_flex_sta.py:29-59).  Same DiT bound as dense."""
    from fastvideo_amd.wan_dit import WanTransformer3DModelHip
    from oracle import vsa_oracle as V
    from oracle import wan_oracle as W
    cfg, sd = block_1_3b
    latent, ctx, t = _cfg2_inputs(cfg)
    grid, window, tile = (21, 30, 52), (3, 3, 3), (6, 8, 8)
    torch.set_num_threads(min(64, os.cpu_count() or 8))

    def sta_attention(q, k, v, scale):   # [B,S,H,D]
        return V.sta_attention_ragged(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2), scale, grid, window, tile).transpose(1, 2).to(q.dtype)

    orc = W.WanOracle(sd, num_heads=cfg.num_heads)
    orc.attention = sta_attention
    tr_ref, tr = {}, {}
    with torch.no_grad():
        y_ref = orc.forward(latent, ctx, t, trace=tr_ref)
    model = WanTransformer3DModelHip(sd, cfg.num_heads, cfg.head_dim, cfg.patch_size, cfg.eps, cfg.freq_dim, attention="sta",
                                     sta_window=window, sta_tile=tile)
    y = model(latent.cuda(), ctx.cuda(), t.cuda(), trace=tr)
    assert model.sta_lists == "grouped" and model.sta_fold
    for key in ("blocks.0.after_self_attn", "blocks.0.out", "norm_out"):
        _cmp(tr[key], tr_ref[key], f"cfg2 sta {key}")
    _cmp(y, y_ref, "cfg2 sta model output")
    # the window really bites: right after the self-attention sub-block the dense oracle is >= 20 x further from the sliding-tile oracle than
    # the device is (measured: device 3.6e-5; at the model output the later sub-blocks' own rounding dominates both)
    tr_dense = {}
    with torch.no_grad():
        W.WanOracle(sd, num_heads=cfg.num_heads).forward(latent, ctx, t, trace=tr_dense)
    key = "blocks.0.after_self_attn"
    d_window = (tr_dense[key].float() - tr_ref[key].float()).abs().mean().item()
    d_device = (tr[key].float().cpu() - tr_ref[key].float()).abs().mean().item()
    print(f"cfg2 sta: dense oracle vs sliding-tile oracle after self-attention {d_window:.4g}, device vs sliding-tile oracle {d_device:.4g}")
    assert d_window > 20 * d_device


def test_one_block_at_cfg2_vsa_matches_oracle():
    """The VSA model line of the bench at its real geometry: sparsity 0.8 -> top-125 of 624 blocks (6 x 8 x 13 tiles of 4x4x4, ragged in
    every axis), 12 heads, compress gate from the fourth column block of the fused QKV+gate GEMM (row stride 6144), the gather-free path
    (_vsa_fused: tile(q), tile(k), tile(gate), untile(out) folded into neighbouring kernels through row maps) — vs the oracle's
    ``video_sparse_attn`` (fastvideo_kernel/ops.py:65-133 restated) evaluated with the device's OWN block selection, as
    tests/test_gpu_kernels.py does at kernel level (one bf16 ulp in a coarse score flips a near-tie; the selection itself is bit-exact
    against the reference's top-k kernel in test_gpu_ref_triton.py).  Same DiT bound as dense."""
    from fastvideo_amd import wan_config as WC
    from fastvideo_amd.wan_dit import WanTransformer3DModelHip
    from oracle import vsa_oracle as V
    from oracle import wan_oracle as W
    cfg = WC.WanConfig("Wan2.1-T2V-1.3B geometry, 1 layer, VSA", 12, 128, 8960, 1)
    sd = WC.random_state_dict(cfg, seed=0, device="cpu", with_vsa_gate=True)
    gen = torch.Generator().manual_seed(5)
    for k in ("blocks.0.scale_shift_table", "scale_shift_table"):
        sd[k] = (torch.randn(sd[k].shape, generator=gen) * 0.3).to(sd[k].dtype)
    assert "blocks.0.to_gate_compress.weight" in sd
    latent, ctx, t = _cfg2_inputs(cfg)
    torch.set_num_threads(min(64, os.cpu_count() or 8))
    model = WanTransformer3DModelHip(sd, cfg.num_heads, cfg.head_dim, cfg.patch_size, cfg.eps, cfg.freq_dim, attention="vsa", vsa_sparsity=0.8)
    assert model.vsa_fold and model.vsa_gate
    model.vsa_trace = []
    tr = {}
    y = model(latent.cuda(), ctx.cuda(), t.cuda(), trace=tr)
    masks, model.vsa_trace = [m_.cpu().numpy() for m_ in model.vsa_trace], None
    md = V.build_metadata(tuple(latent.shape[2:]))
    vbs = md["variable_block_sizes"]
    topk = V.compute_topk(0.8, len(vbs))
    assert len(vbs) == 624 and topk == 125 and len(masks) == 1 and masks[0].shape == (1, 12, 624, 624)
    assert (masks[0].sum(-1) == topk).all()
    it = iter(masks)

    def vsa_attention(q, k, v, scale, gate):
        tq, tk, tv, tg = (V.tile(x_, md).transpose(1, 2).contiguous() for x_ in (q, k, v, gate))
        o, info = V.video_sparse_attn(tq, tk, tv, vbs, vbs, topk, 64, tg, mask_override=next(it), gathered=True)
        # the oracle's own selection from ITS coarse scores: nearly the device's (near-ties only)
        own = V.topk_mask_bisect(info["scores"].float().numpy(), topk)
        agree = (own == info["mask"]).mean()
        print(f"vsa block selection: oracle's own top-k agrees with the device's on {agree:.6f} of the 12 x 624 x 624 entries")
        assert agree > 0.999
        return V.untile(o.transpose(1, 2), md)

    orc = W.WanOracle(sd, num_heads=cfg.num_heads)
    orc.attention, orc.vsa_gate = vsa_attention, True
    tr_ref = {}
    with torch.no_grad():
        y_ref = orc.forward(latent, ctx, t, trace=tr_ref)
    for key in ("blocks.0.after_self_attn", "blocks.0.out", "norm_out"):
        _cmp(tr[key], tr_ref[key], f"cfg2 vsa {key}")
    _cmp(y, y_ref, "cfg2 vsa model output")


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
