"""Wan2.1 VAE decode timing at the 81f x 480p latent ([1,16,21,60,104] -> [1,3,81,480,832]) — not the contract bench.
usage: python scripts/vae_bench.py [--frames 21] [--h 60] [--w 104] [--iters 2] [--mode cached|tiled|spatial|plain|stream] [--u8]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=21)
ap.add_argument("--h", type=int, default=60)
ap.add_argument("--w", type=int, default=104)
ap.add_argument("--iters", type=int, default=2)
ap.add_argument("--mode", default="cached", choices=["cached", "tiled", "spatial", "plain", "stream"],
                help="cached = frame-chunked decode (default); tiled/spatial/plain = the cache-less family at the default 256-px tiles; "
                     "stream = streaming_decode in 3-latent-frame calls")
ap.add_argument("--u8", action="store_true", help="also time the uint8 post-processing kernel")
args = ap.parse_args()

from fastvideo_amd.wan_vae import WanVaeDecoderHip
from fastvideo_amd.wan_config import wan_vae_param_spec, vae_decode_flops

spec = wan_vae_param_spec(base_dim=96)
g = torch.Generator().manual_seed(0)
sd = {}
for n, s in spec:
    fan_in = 1
    for d in s[1:]:
        fan_in *= d
    sd[n] = ((torch.rand(s, generator=g) * 2 - 1) * (3.0 / fan_in)**0.5) if len(s) >= 4 and "gamma" not in n else (
        torch.ones(s) if "gamma" in n else torch.zeros(s))
dec = WanVaeDecoderHip(sd, device="cuda", use_feature_cache=args.mode in ("cached", "stream"))
if args.mode in ("tiled", "spatial"):
    dec.enable_tiling(use_temporal_tiling=args.mode == "tiled")
z = torch.randn((1, 16, args.frames, args.h, args.w), generator=g).cuda()


def run():
    if args.mode == "stream":
        cache, outs = dec.get_streaming_cache(), []
        for i in range(0, args.frames, 3):
            o, cache = dec.streaming_decode(z[:, :, i:i + 3], cache, i == 0)
            outs.append(o)
        return torch.cat(outs, 2)
    if args.mode == "tiled":
        dec.blend_num_frames = 4  # undo the reference's per-call doubling so that every timed call does the same work
    return dec.decode(z)


y = run()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(args.iters):
    y = run()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / args.iters
if args.u8:
    u = dec.postprocess_u8(y)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(10):
        u = dec.postprocess_u8(y)
    torch.cuda.synchronize()
    du = (time.perf_counter() - t1) / 10
    print(json.dumps({"postprocess_u8_ms": round(du * 1e3, 3), "GB/s": round(y.numel() * 5 / du / 1e9, 1), "frames": list(u.shape)}))
fl = vae_decode_flops(args.frames, args.h, args.w)
print(json.dumps({"mode": args.mode, "vae_decode_ms": round(dt * 1e3, 2), "latent": [1, 16, args.frames, args.h, args.w], "pixels": list(y.shape),
                  "algorithmic_tflop": round(fl / 1e12, 2), "tflops": round(fl / dt / 1e12, 1), "finite": bool(torch.isfinite(y).all()),
                  "peak_mem_gb": round(torch.cuda.max_memory_allocated() / 2**30, 2)}))
