"""VAE decode A/B of the 96-channel conv schedule: staggered wave groups (shipped, "vae_conv_impl" 0) vs lockstep ("vae_conv_impl" 2),
interleaved on one box; the two must produce bit-identical pixels (same per-wave MFMA order)."""
import os as _os
_os.environ.setdefault("FVK_PROBE_LIB", "1")  # A/B switches exist only in the measurement build (scripts/probes/libfvk_probe.so)
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fastvideo_amd import ops
from fastvideo_amd.wan_config import wan_vae_param_spec
from fastvideo_amd.wan_vae import WanVaeDecoderHip
g = torch.Generator().manual_seed(0)
sd = {}
for n, s in wan_vae_param_spec(base_dim=96):
    fan_in = 1
    for d in s[1:]: fan_in *= d
    sd[n] = torch.ones(s) if "gamma" in n else (((torch.rand(s, generator=g) * 2 - 1) * (3.0 / fan_in)**0.5) if len(s) >= 4 else torch.zeros(s))
shape = tuple(int(x) for x in sys.argv[1:4]) if len(sys.argv) >= 4 else (21, 60, 104)
z = torch.randn((1, 16, *shape), generator=g).cuda()
dec = WanVaeDecoderHip(sd)
res, outs = {0: [], 2: []}, {}
for r in range(4):
    for impl in (0, 2):
        ops.set_tunable("vae_conv_impl", impl)
        y = dec.decode(z); torch.cuda.synchronize()
        t0 = time.perf_counter(); dec.decode(z); dec.decode(z); torch.cuda.synchronize()
        res[impl].append(round((time.perf_counter() - t0) / 2 * 1e3, 2))
        if r == 0: outs[impl] = y
ops.set_tunable("vae_conv_impl", 0)
print(json.dumps({"latent": shape, "staggered_ms": res[0], "lockstep_ms": res[2], "bit_identical": bool(torch.equal(outs[0], outs[2]))}))
