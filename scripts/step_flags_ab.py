"""Round 5 A/B of the contract forward (30 layers, cfg2) under model switches, interleaved in one process: each switch off against everything on.
usage: python scripts/step_flags_ab.py [reps=6]"""
import json, os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fastvideo_amd import wan_config as WC
from fastvideo_amd.wan_dit import WanTransformer3DModelHip
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
quant = sys.argv[2] if len(sys.argv) > 2 and sys.argv[2] != "bf16" else None      # bf16 | fp8 | fp8_channel
config = sys.argv[3] if len(sys.argv) > 3 else "cfg2"                              # cfg2 | cfg1
dev = torch.device("cuda")
cfg = WC.WAN21_T2V_1_3B
sd = WC.random_state_dict(cfg, seed=0, device=dev)
model = WanTransformer3DModelHip(sd, cfg.num_heads, cfg.head_dim, cfg.patch_size, cfg.eps, cfg.freq_dim, device=dev, quantization=quant)
del sd
g = torch.Generator(device=dev).manual_seed(1)
lat = torch.randn(WC.LATENT_CFG1 if config == "cfg1" else WC.LATENT_81F_480P, generator=g, device=dev).bfloat16()
ctx = torch.randn((1, 512, cfg.text_dim), generator=g, device=dev).bfloat16()
ts = torch.tensor([500.0], device=dev)
FLAGS = ("vt_gemm", "fuse_cross_residual")
configs = [("all on (shipped)", {})] + [(f"{f} off", {f: False}) for f in FLAGS] + [("all off (round 4's path)", {f: False for f in FLAGS})]


def apply(c):
    for f in FLAGS:
        setattr(model, f, c.get(f, True))


outs = {}
for name, c in configs:
    apply(c)
    for _ in range(3): outs[name] = model(lat, ctx, ts)
torch.cuda.synchronize()
ms = {name: [] for name, _ in configs}
for rep in range(reps):
    for name, c in configs:
        apply(c)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(4): model(lat, ctx, ts)
        e.record(); torch.cuda.synchronize()
        ms[name].append(round(s.elapsed_time(e) / 4, 3))
ref = outs["all on (shipped)"]
print(json.dumps({"quant": quant or "bf16", "config": config, "forward_ms": ms, "median_ms": {n: statistics.median(v) for n, v in ms.items()},
                  "bit_identical_to_shipped": {n: bool(torch.equal(o, ref)) for n, o in outs.items()}}), flush=True)
