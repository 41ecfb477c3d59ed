"""Generate tests/golden/wan_causal.pt by running the REAL reference ``CausalWanTransformer3DModel`` (imported from /root/reference, CPU,
bf16 autocast) through KV-cached block-by-block rollouts — the call pattern of ``CausalDMDDenosingStage``
(fastvideo/pipelines/stages/causal_denoising.py:205-349): per block one forward at a noisy timestep, then one forward at the context
timestep over the SAME positions (cache rewrite), then the next block.

Run in the build container only:  ``python oracle/make_golden_causal.py``.
Same tiny weights as wan_tiny.pt (2 heads x 128, ffn 512, 2 layers, text_dim 64; seed 0, modulation_std 0.05): the causal model has the
same parameter names.  Rollout cases (latent blocks [1,16,2,8,8]: 16 tokens per frame):
  "global"        local_attn_size -1, 3 blocks, cache of 8 frames
  "local_sink"    local_attn_size 4, sink_size 1: the 3rd block evicts by a left shift behind the sink frame; per-frame timesteps [1, 2]
  "relativistic"  as local_sink with rope_cache_policy="relativistic" (raw keys cached, window re-roped at attention time)"""
from __future__ import annotations

import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import ref_loader as R  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
TINY = dict(num_heads=2, head_dim=128, ffn_dim=512, num_layers=2, text_dim=64)
CASES = [
    dict(name="global", local_attn_size=-1, sink_size=0, rope_cache_policy="absolute", blocks=3, cache_frames=8, per_frame_t=False),
    dict(name="local_sink", local_attn_size=4, sink_size=1, rope_cache_policy="absolute", blocks=4, cache_frames=4, per_frame_t=True),
    dict(name="relativistic", local_attn_size=4, sink_size=1, rope_cache_policy="relativistic", blocks=4, cache_frames=4, per_frame_t=False),
]
FRAMES, HH, WW = 2, 8, 8


def rollout_inputs(case, seed=3):
    g = torch.Generator().manual_seed(seed)
    ctx = torch.randn(1, 16, TINY["text_dim"], generator=g).bfloat16()
    calls = []
    start = 0
    for b in range(case["blocks"]):
        x = torch.randn(1, 16, FRAMES, HH, WW, generator=g).bfloat16()
        for t in (750 - 100 * b, 0):  # noisy step, then the context re-run over the same positions
            ts = torch.tensor([[t, max(t - 7, 0)]]) if case["per_frame_t"] else torch.tensor([[t]])
            calls.append(dict(latent=x, timestep=ts.long(), start_frame=start))
        start += FRAMES
    return ctx, calls


def main():
    R.install()
    R.init_distributed()
    from fastvideo.forward_context import set_forward_context
    fs = (HH // 2) * (WW // 2)
    out = dict(config=TINY, frame_seqlen=fs, cases=[])
    for case in CASES:
        m = R.build_causal_wan(**TINY, seed=0, modulation_std=0.05, dtype=torch.bfloat16, local_attn_size=case["local_attn_size"],
                               sink_size=case["sink_size"], num_frames_per_block=FRAMES, rope_cache_policy=case["rope_cache_policy"])
        if "state_dict" not in out:
            out["state_dict"] = {k: v.detach().clone() for k, v in m.state_dict().items()}
        ctx, calls = rollout_inputs(case)
        n = case["cache_frames"] * fs
        kv = [dict(k=torch.zeros(1, n, TINY["num_heads"], TINY["head_dim"], dtype=torch.bfloat16),
                   v=torch.zeros(1, n, TINY["num_heads"], TINY["head_dim"], dtype=torch.bfloat16),
                   global_end_index=torch.tensor([0]), local_end_index=torch.tensor([0])) for _ in range(TINY["num_layers"])]
        rec = []
        for c in calls:
            with torch.no_grad(), torch.autocast("cpu", dtype=torch.bfloat16), set_forward_context(current_timestep=0, attn_metadata=None):
                y = m(c["latent"], ctx, c["timestep"], kv_cache=kv, crossattn_cache=None, current_start=c["start_frame"] * fs,
                      start_frame=c["start_frame"])
            rec.append(dict(**c, out=y.detach().clone(), global_end=int(kv[0]["global_end_index"]), local_end=int(kv[0]["local_end_index"])))
        rec[-1].update(k_cache=kv[-1]["k"].clone(), v_cache=kv[-1]["v"].clone())  # the last layer's cache after the whole rollout
        out["cases"].append(dict(**{k: v for k, v in case.items()}, ctx=ctx, calls=rec))
        print(case["name"], [(r["global_end"], r["local_end"]) for r in rec])
    torch.save(out, os.path.join(OUT, "wan_causal.pt"))
    print("wan_causal.pt", os.path.getsize(os.path.join(OUT, "wan_causal.pt")) / 1e6, "MB")


if __name__ == "__main__":
    main()
