"""Causal (KV-cached) Wan2.1-1.3B rollout timing at 480p — not the contract bench.  21 latent frames in 7 blocks of 3, per block
`--steps` DMD forwards + the clean-context re-run (CausalDMDDenosingStage's call pattern), KV cache of 21 frames (global attention).
usage: python scripts/causal_bench.py [--steps 3] [--frames 21] [--local 0] [--sink 0]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--frames", type=int, default=21)
ap.add_argument("--local", type=int, default=0, help="local_attn_size in frames (0 = global, 21-frame window)")
ap.add_argument("--sink", type=int, default=0)
args = ap.parse_args()

import __graft_entry__ as G
G.build()
from fastvideo_amd import wan_config as WC
from fastvideo_amd.wan_causal import CausalWanTransformer3DModelHip, CausalDenoisingLoopHip

dev = torch.device("cuda:0")
cfg = WC.WAN21_T2V_1_3B
sd = WC.random_state_dict(cfg, seed=0, device=dev)
model = CausalWanTransformer3DModelHip(sd, cfg.num_heads, cfg.head_dim, cfg.patch_size, cfg.eps, cfg.freq_dim,
                                       local_attn_size=(args.local or -1), sink_size=args.sink, num_frames_per_block=3, device=dev)
del sd
g = torch.Generator(device=dev).manual_seed(1)
lat = torch.randn((1, 16, args.frames, 60, 104), generator=g, device=dev)
ctx = torch.randn((1, 512, cfg.text_dim), generator=g, device=dev).bfloat16()
steps = [1000, 750, 500, 250][:args.steps]
loop = CausalDenoisingLoopHip(model, steps)
noise_fn = lambda shape, dtype: torch.randn(shape, generator=g, device=dev, dtype=torch.float32).to(dtype)
loop.run(lat[:, :, :6], ctx, noise_fn)  # warm-up (two blocks)
torch.cuda.synchronize()
model.attn_events = []
t0 = time.perf_counter()
out = loop.run(lat, ctx, noise_fn)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
ev, model.attn_events = model.attn_events, None
assert torch.isfinite(out).all()
blocks = args.frames // 3
fwd = blocks * (len(steps) + 1)
attn_ms = sum(e0.elapsed_time(e1) for e0, e1, *_ in ev)
flops = sum(4.0 * sq * skv * h * cfg.head_dim for _, _, sq, skv, h in ev)
print(json.dumps({"what": "causal Wan2.1-1.3B rollout, 480p", "latent_frames": args.frames, "blocks": blocks, "dmd_steps": len(steps),
                  "model_forwards": fwd, "tokens_per_forward": 3 * 1560, "total_ms": round(dt * 1e3, 1), "ms_per_forward": round(dt * 1e3 / fwd, 2),
                  "latent_frames_per_s": round(args.frames / dt, 2), "pixel_frames_per_s": round((1 + 4 * (args.frames - 1)) / dt, 1),
                  "self_attention_ms": round(attn_ms, 1), "self_attention_tflops": round(flops / (attn_ms * 1e-3) / 1e12, 1),
                  "kv_cache_gb": round(sum(c["k"].numel() * 4 for c in model.init_kv_cache(1)) * 0 + 30 * 2 * 2 * loop.cache_tokens(1560) * 1536 / 1e9, 2)}))
