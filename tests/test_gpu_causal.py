"""Parity of the causal (KV-cached) HIP Wan DiT path against the REAL reference's rollouts (tests/golden/wan_causal.pt, produced by
oracle/make_golden_causal.py from /root/reference on CPU) and against the oracle restating the reference's GPU autocast policy
(oracle/causal_oracle.py, ln_policy="cuda").  Tolerance: the reference's own DiT bound atol=1e-1, rtol=1e-2
(fastvideo/tests/transformers/test_wanvideo.py:109) plus a much tighter mean error; cache bookkeeping (integers) must be identical."""
import os

import pytest
import torch

from oracle import causal_oracle as CO

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fx(golden_dir):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.load(os.path.join(golden_dir, "wan_causal.pt"), weights_only=False)


def _cmp(y, ref, what, atol=1e-1, rtol=1e-2, mean_tol=1.5e-2):
    y, ref = y.float().cpu(), ref.float()
    assert torch.isfinite(y).all(), what
    err = (y - ref).abs()
    bad = err > atol + rtol * ref.abs()
    print(f"{what}: max|err|={err.max().item():.4g} mean|err|={err.mean().item():.4g} ref_absmean={ref.abs().mean().item():.4g}")
    assert not bad.any(), f"{what}: {int(bad.sum())}/{bad.numel()} outside tolerance, max {err.max().item():.4g}"
    assert err.mean().item() < mean_tol, f"{what}: mean error {err.mean().item():.4g}"


def _model(fx, case):
    from fastvideo_amd.wan_causal import CausalWanTransformer3DModelHip
    return CausalWanTransformer3DModelHip(fx["state_dict"], num_heads=fx["config"]["num_heads"], local_attn_size=case["local_attn_size"],
                                          sink_size=case["sink_size"], rope_cache_policy=case["rope_cache_policy"], num_frames_per_block=2)


@pytest.mark.parametrize("ci", [0, 1, 2])
def test_rollout_matches_reference_and_oracle(fx, ci):
    case = fx["cases"][ci]
    fs = fx["frame_seqlen"]
    model = _model(fx, case)
    kv = model.init_kv_cache(case["cache_frames"] * fs)
    cc = model.init_crossattn_cache()
    orc = CO.CausalWanOracle(fx["state_dict"], num_heads=fx["config"]["num_heads"], local_attn_size=case["local_attn_size"],
                             sink_size=case["sink_size"], rope_cache_policy=case["rope_cache_policy"], ln_policy="cuda")
    kv_o = orc.init_kv_cache(1, case["cache_frames"] * fs)
    for j, call in enumerate(case["calls"]):
        y = model(call["latent"].cuda(), case["ctx"].cuda(), call["timestep"].cuda(), kv_cache=kv, crossattn_cache=cc,
                  current_start=call["start_frame"] * fs, start_frame=call["start_frame"])
        with torch.no_grad():
            y_o = orc.forward_inference(call["latent"], case["ctx"], call["timestep"], kv_o, current_start=call["start_frame"] * fs,
                                        start_frame=call["start_frame"])
        assert y.shape == call["out"].shape and y.dtype == torch.bfloat16
        assert (kv[0]["global_end_index"], kv[0]["local_end_index"]) == (call["global_end"], call["local_end"])
        _cmp(y, y_o, f"{case['name']} call {j} vs oracle (GPU autocast policy)")
        _cmp(y, call["out"], f"{case['name']} call {j} vs reference (CPU run)")
    # the last layer's cache after the whole rollout (evictions included): same slots, values within bf16 noise of the oracle's
    for key in ("k", "v"):
        _cmp(kv[-1][key], kv_o[-1][key], f"{case['name']} {key} cache vs oracle", atol=6e-2, rtol=2e-2, mean_tol=5e-3)
        _cmp(kv[-1][key], case["calls"][-1][f"{key}_cache"], f"{case['name']} {key} cache vs reference", atol=1e-1, rtol=2e-2, mean_tol=8e-3)
    assert all(c["is_init"] for c in cc)


def test_crossattn_cache_is_transparent(fx):
    """Keeping the text K / V across calls (wanvideo.py:202-214) must not change any output bit."""
    case = fx["cases"][1]
    fs = fx["frame_seqlen"]
    outs = []
    for use_cc in (False, True):
        model = _model(fx, case)
        kv = model.init_kv_cache(case["cache_frames"] * fs)
        cc = model.init_crossattn_cache() if use_cc else None
        ys = [model(c["latent"].cuda(), case["ctx"].cuda(), c["timestep"].cuda(), kv_cache=kv, crossattn_cache=cc,
                    current_start=c["start_frame"] * fs, start_frame=c["start_frame"]) for c in case["calls"][:4]]
        outs.append(torch.stack(ys))
    assert torch.equal(outs[0], outs[1])


def test_first_block_equals_bidirectional_model(fx):
    """With an empty cache and one timestep row the first block attends only to itself at positions 0..F-1: it must equal the
    bidirectional model on the same latent up to the causal block's rounding points (bf16 modulation vectors)."""
    from fastvideo_amd.wan_dit import WanTransformer3DModelHip
    case = fx["cases"][0]
    fs = fx["frame_seqlen"]
    call = case["calls"][0]
    model = _model(fx, case)
    kv = model.init_kv_cache(case["cache_frames"] * fs)
    y = model(call["latent"].cuda(), case["ctx"].cuda(), call["timestep"].cuda(), kv_cache=kv, current_start=0, start_frame=0)
    ctx = torch.zeros(1, 512, case["ctx"].shape[2], dtype=torch.bfloat16)
    ctx[:, :case["ctx"].shape[1]] = case["ctx"]
    bi = WanTransformer3DModelHip(fx["state_dict"], num_heads=fx["config"]["num_heads"])
    y_bi = bi(call["latent"].cuda(), ctx.cuda(), call["timestep"].reshape(-1)[:1].float().cuda())
    _cmp(y, y_bi.cpu(), "first causal block vs bidirectional forward", atol=6e-2, rtol=1e-2, mean_tol=6e-3)


def test_refuses_cpu_and_bad_arguments(fx):
    case = fx["cases"][0]
    model = _model(fx, case)
    kv = model.init_kv_cache(64)
    with pytest.raises(RuntimeError):
        model(case["calls"][0]["latent"], case["ctx"], case["calls"][0]["timestep"], kv_cache=kv)
    with pytest.raises(NotImplementedError):
        model(case["calls"][0]["latent"].cuda(), case["ctx"].cuda(), case["calls"][0]["timestep"].cuda())
    with pytest.raises(ValueError):  # global attention keeps at most 21 latent frames (causal_wanvideo.py:131-136)
        model(case["calls"][0]["latent"].cuda(), case["ctx"].cuda(), case["calls"][0]["timestep"].cuda(), kv_cache=model.init_kv_cache(64),
              current_start=21 * fx["frame_seqlen"], start_frame=21)


def test_causal_dmd_rollout_matches_oracle_loop(fx):
    """CausalDenoisingLoopHip (block loop of CausalDMDDenosingStage, causal_denoising.py:205-349) vs the same loop on the oracle model with
    the oracle DMD arithmetic and the SAME re-noising draws: 3 blocks x 3 DMD steps + context re-runs, local window with a sink frame."""
    from fastvideo_amd.wan_causal import CausalDenoisingLoopHip
    case = fx["cases"][1]
    g = torch.Generator().manual_seed(21)
    lat = torch.randn(1, 16, 6, 8, 8, generator=g)
    steps = [1000, 750, 400]
    draws = [torch.randn(1, 2, 16, 8, 8, generator=g).bfloat16() for _ in range(3 * 2)]
    orc = CO.CausalWanOracle(fx["state_dict"], num_heads=fx["config"]["num_heads"], local_attn_size=case["local_attn_size"],
                             sink_size=case["sink_size"], ln_policy="cuda")
    with torch.no_grad():
        ref = CO.causal_dmd_rollout(orc, lat, case["ctx"], steps, draws, 2, case["local_attn_size"])
    model = _model(fx, case)
    it = iter(draws)
    loop = CausalDenoisingLoopHip(model, steps)
    y = loop.run(lat.cuda(), case["ctx"].cuda(), lambda shape, dtype: next(it))
    assert y.shape == lat.shape and y.dtype == lat.dtype
    _cmp(y, ref, "causal DMD rollout", atol=1.5e-1, rtol=2e-2, mean_tol=2e-2)
    assert next(it, None) is None  # every draw consumed, in order
