"""A/B of the contract forward (30 layers, cfg2, bf16) under measurement-build tunables, interleaved in one process.
usage: python scripts/step_tunable_ab.py '<json: [[name, {tunable: value, ...}], ...]>' [reps=6]
Under `rocprofv3 --kernel-trace --stats` the variants' kernels carry different template arguments, so the per-kernel averages of ONE run are the
in-step A/B of the launch itself (V^T GEMM with streaming stores <5,143> vs plain stores <5,15>)."""
import os as _os
_os.environ.setdefault("FVK_PROBE_LIB", "1")
import json, os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fastvideo_amd import ops, wan_config as WC
from fastvideo_amd.wan_dit import WanTransformer3DModelHip
variants = json.loads(sys.argv[1])
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
dev = torch.device("cuda")
cfg = WC.WAN21_T2V_1_3B
sd = WC.random_state_dict(cfg, seed=0, device=dev)
model = WanTransformer3DModelHip(sd, cfg.num_heads, cfg.head_dim, cfg.patch_size, cfg.eps, cfg.freq_dim, device=dev)
del sd
g = torch.Generator(device=dev).manual_seed(1)
lat = torch.randn(WC.LATENT_81F_480P, generator=g, device=dev).bfloat16()
ctx = torch.randn((1, 512, cfg.text_dim), generator=g, device=dev).bfloat16()
ts = torch.tensor([500.0], device=dev)
names = sorted({k for _, t in variants for k in t})


def apply(t):
    for k in names:
        ops.set_tunable(k, int(t.get(k, 0)))


outs = {}
for name, t in variants:
    apply(t)
    for _ in range(2): outs[name] = model(lat, ctx, ts)
torch.cuda.synchronize()
ms = {name: [] for name, _ in variants}
for rep in range(reps):
    for name, t in variants:
        apply(t)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(3): model(lat, ctx, ts)
        e.record(); torch.cuda.synchronize()
        ms[name].append(round(s.elapsed_time(e) / 3, 3))
apply({})
ref = outs[variants[0][0]]
print(json.dumps({"forward_ms": ms, "median_ms": {n: statistics.median(v) for n, v in ms.items()},
                  "bit_identical_to_first": {n: bool(torch.equal(o, ref)) for n, o in outs.items()}}), flush=True)
