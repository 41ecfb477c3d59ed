"""Where a Wan2.1 VAE decode spends its time, by conv shape: per (taps, Cin, Cout, output H x W) the launch count, total ms and TFLOP/s
(HIP events around each conv launch, ops.VAE_CONV_EVENTS), plus everything that is not a conv as the remainder of the wall time.
usage: python scripts/vae_conv_breakdown.py [--frames 21 --h 60 --w 104 --frames-per-pass 4]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=21)
ap.add_argument("--h", type=int, default=60)
ap.add_argument("--w", type=int, default=104)
ap.add_argument("--frames-per-pass", type=int, default=4)
ap.add_argument("--impl", type=int, default=0, help="vae_conv_impl tunable (measurement build, FVK_PROBE_LIB=1): 3 = the 8-wave vae_conv3.hip kernels everywhere")
args = ap.parse_args()

from fastvideo_amd import ops
from fastvideo_amd.wan_config import wan_vae_param_spec
from fastvideo_amd.wan_vae import WanVaeDecoderHip

g = torch.Generator().manual_seed(0)
sd = {}
for n, s in wan_vae_param_spec(base_dim=96):
    fan_in = 1
    for d in s[1:]:
        fan_in *= d
    sd[n] = torch.ones(s) if "gamma" in n else (((torch.rand(s, generator=g) * 2 - 1) * (3.0 / fan_in)**0.5) if len(s) >= 4 else torch.zeros(s))
if args.impl:
    ops.set_tunable("vae_conv_impl", args.impl)
dec = WanVaeDecoderHip(sd, frames_per_pass=args.frames_per_pass)
z = torch.randn((1, 16, args.frames, args.h, args.w), generator=g).cuda()
dec.decode(z)
torch.cuda.synchronize()
t0 = time.perf_counter()
dec.decode(z)
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) * 1e3

_orig = ops._conv_timed
shapes = []


def _timed(kt, ks, Cin, Cout, T, H, W, launch):
    shapes.append((T, H, W))
    return _orig(kt, ks, Cin, Cout, T, H, W, launch)


ops._conv_timed = _timed
ops.VAE_CONV_EVENTS = []
dec.decode(z)
torch.cuda.synchronize()
ev, ops.VAE_CONV_EVENTS = ops.VAE_CONV_EVENTS, None
rows = {}
for (taps, cin, cout, fl, e0, e1), (T, H, W) in zip(ev, shapes):
    r = rows.setdefault((taps, cin, cout, H, W), [0, 0.0, 0.0, set()])
    r[0] += 1
    r[1] += e0.elapsed_time(e1)
    r[2] += fl
    r[3].add(T)
tot = sum(r[1] for r in rows.values())
print(f"decode wall {wall:.1f} ms; conv launches {len(ev)}, conv time {tot:.1f} ms ({tot / wall:.1%}); the rest (norms, mid attention, "
      f"GEMMs, copies, host gaps) {wall - tot:.1f} ms")
print(f"{'taps':8} {'Cin':>4} {'Cout':>4} {'HxW':>10} {'T/launch':>9} {'n':>5} {'ms':>8} {'share':>6} {'TFLOP/s':>8}")
for (taps, cin, cout, H, W), (n, ms, fl, ts) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
    print(f"{taps:8} {cin:4d} {cout:4d} {H:4d}x{W:<5d} {'/'.join(map(str, sorted(ts))):>9} {n:5d} {ms:8.2f} {ms / wall:6.1%} {fl / ms / 1e9:8.1f}")
