"""Pins oracle/causal_oracle.py (KV-cached causal Wan DiT inference) against the REAL reference: the committed rollouts of
tests/golden/wan_causal.pt (oracle/make_golden_causal.py) everywhere, and the live reference class on the same host when
/root/reference is present (bit-exact there; the cross-host fixture comparison allows host-dependent bf16 GEMM rounding)."""
import os

import pytest
import torch

from oracle import causal_oracle as CO
from oracle import ref_loader as R
from tests.test_oracle_golden import _same_or_host_rounding


@pytest.fixture(scope="module")
def fx(golden_dir):
    return torch.load(os.path.join(golden_dir, "wan_causal.pt"), weights_only=False)


def _oracle(fx, case):
    return CO.CausalWanOracle(fx["state_dict"], num_heads=fx["config"]["num_heads"], local_attn_size=case["local_attn_size"],
                              sink_size=case["sink_size"], rope_cache_policy=case["rope_cache_policy"])


def test_rollouts_vs_golden(fx):
    fs = fx["frame_seqlen"]
    assert [c["name"] for c in fx["cases"]] == ["global", "local_sink", "relativistic"]
    for case in fx["cases"]:
        o = _oracle(fx, case)
        kv = o.init_kv_cache(1, case["cache_frames"] * fs)
        for j, call in enumerate(case["calls"]):
            with torch.no_grad():
                y = o.forward_inference(call["latent"], case["ctx"], call["timestep"], kv, current_start=call["start_frame"] * fs,
                                        start_frame=call["start_frame"])
            assert (kv[0]["global_end_index"], kv[0]["local_end_index"]) == (call["global_end"], call["local_end"]), (case["name"], j)
            _same_or_host_rounding(y, call["out"], f"{case['name']} call {j}")
        _same_or_host_rounding(kv[-1]["k"], case["calls"][-1]["k_cache"], case["name"] + " k cache")
        _same_or_host_rounding(kv[-1]["v"], case["calls"][-1]["v_cache"], case["name"] + " v cache")


def test_cache_plan_integer_cases():
    """causal_wanvideo.py:123-173 bookkeeping: growth, rewrite of the same positions, eviction behind the sink, window clamp."""
    fs = 16
    p = CO.cache_update_plan(4, 1, fs, 4 * fs, 2 * fs, 0, 0, 0)
    assert p["evict"] is None and p["write"] == (0, 32) and p["window"] == (0, 32) and (p["global_end"], p["local_end"]) == (32, 32)
    p = CO.cache_update_plan(4, 1, fs, 4 * fs, 2 * fs, 0, 32, 32)  # context re-run over the same positions
    assert p["evict"] is None and p["write"] == (0, 32) and (p["global_end"], p["local_end"]) == (32, 32)
    p = CO.cache_update_plan(4, 1, fs, 4 * fs, 2 * fs, 64, 64, 64)  # cache full: evict 32 tokens behind the 16 sink tokens
    assert p["evict"] == (16 + 32, 16, 64 - 32 - 16) and p["write"] == (32, 64) and p["window"] == (0, 64) and p["local_end"] == 64
    p = CO.cache_update_plan(2, 0, fs, 4 * fs, 2 * fs, 32, 32, 32)  # window shorter than the cache
    assert p["window"] == (32, 64) and p["evict"] is None
    with pytest.raises(ValueError):
        CO.cache_update_plan(-1, 0, fs, 64 * fs, fs, 21 * fs, 21 * fs, 21 * fs)


@pytest.mark.skipif(not R.available(), reason="needs the reference checkout (/root/reference)")
def test_rollouts_vs_live_reference_bit_exact(fx):
    R.install()
    R.init_distributed()
    from fastvideo.forward_context import set_forward_context
    fs, cfg = fx["frame_seqlen"], fx["config"]
    for case in fx["cases"]:
        m = R.build_causal_wan(**cfg, seed=0, modulation_std=0.05, dtype=torch.bfloat16, local_attn_size=case["local_attn_size"],
                               sink_size=case["sink_size"], num_frames_per_block=2, rope_cache_policy=case["rope_cache_policy"])
        sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
        assert all(torch.equal(sd[k], fx["state_dict"][k]) for k in sd)
        o = _oracle(fx, case)
        n = case["cache_frames"] * fs
        kv_o = o.init_kv_cache(1, n)
        kv_r = [dict(k=torch.zeros(1, n, cfg["num_heads"], cfg["head_dim"], dtype=torch.bfloat16),
                     v=torch.zeros(1, n, cfg["num_heads"], cfg["head_dim"], dtype=torch.bfloat16),
                     global_end_index=torch.tensor([0]), local_end_index=torch.tensor([0])) for _ in range(cfg["num_layers"])]
        for j, call in enumerate(case["calls"]):
            with torch.no_grad(), torch.autocast("cpu", dtype=torch.bfloat16), set_forward_context(current_timestep=0, attn_metadata=None):
                y_ref = m(call["latent"], case["ctx"], call["timestep"], kv_cache=kv_r, crossattn_cache=None,
                          current_start=call["start_frame"] * fs, start_frame=call["start_frame"])
            with torch.no_grad():
                y = o.forward_inference(call["latent"], case["ctx"], call["timestep"], kv_o, current_start=call["start_frame"] * fs,
                                        start_frame=call["start_frame"])
            assert torch.equal(y, y_ref), f"{case['name']} call {j}: max diff {(y.float() - y_ref.float()).abs().max().item()}"
            for a, b in zip(kv_o, kv_r):
                assert torch.equal(a["k"], b["k"]) and torch.equal(a["v"], b["v"])


def test_product_cache_plan_equals_oracle_plan():
    """The host's own copy of the cache bookkeeping (fastvideo_amd/wan_causal.py) against the oracle's restatement of
    causal_wanvideo.py:123-173 over randomised rollouts: window sizes, sink frames, block sizes, context re-runs, cache sizes."""
    import itertools
    from fastvideo_amd import wan_causal as WCa
    n = 0
    for las, sink, fs, cf, nb in itertools.product([-1, 2, 4, 6], [0, 1, 2], [4, 16], [4, 6, 21], [1, 2, 3]):
        if las != -1 and sink >= las:
            continue
        cache, ge, le, start = cf * fs, 0, 0, 0
        for _ in range(12):
            stop = False
            for _rerun in (0, 1):
                args = (las, sink, fs, cache, nb * fs, start * fs, ge, le)
                try:
                    b = CO.cache_update_plan(*args)
                except ValueError:
                    with pytest.raises(ValueError):
                        WCa.cache_update_plan(*args)
                    stop = True
                    break
                try:
                    a = WCa.cache_update_plan(*args)
                except ValueError:  # the product additionally refuses writes past the cache (the reference would raise inside torch)
                    assert b["write"][1] > cache or b["write"][0] < 0
                    stop = True
                    break
                assert a == b
                ge, le = a["global_end"], a["local_end"]
                n += 1
            if stop:
                break
            start += nb
    assert n > 3000
