"""The VSA forward at the contract geometry with / without the split last round of attn_bs16 (ops._bs_workspace), interleaved in one process.
(The version of commit f214e3e also toggled the combine pass fused into the sparse kernel's store — measured slower and reverted:
profiles/r06g_vsa_step_ab_fused_combine_and_split.log.)"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fastvideo_amd import ops, wan_config as WC
from fastvideo_amd.wan_dit import WanTransformer3DModelHip
dev = torch.device("cuda")
cfg = WC.WanConfig("wan2.1-1.3b", 12, 128, 8960, 30)
sd = WC.random_state_dict(cfg, seed=0, device=dev, with_vsa_gate=True)
model = WanTransformer3DModelHip(sd, cfg.num_heads, attention="vsa", device=dev)
g = torch.Generator(device=dev).manual_seed(1)
lat = torch.randn(WC.LATENT_81F_480P, generator=g, device=dev).bfloat16()
ctx = torch.randn((1, 512, cfg.text_dim), generator=g, device=dev).bfloat16()
t = torch.tensor([500.0], device=dev)
orig_ws = ops._bs_workspace
modes = {"split last round (shipped)": True, "every list whole": False}
res, outs = {k: [] for k in modes}, {}
for rep in range(int(os.environ.get("REPS", "4"))):
    for name, split in modes.items():
        ops._bs_workspace = orig_ws if split else (lambda a, m, d: (None, 0))
        o = model(lat, ctx, t); torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(3): model(lat, ctx, t)
        e.record(); torch.cuda.synchronize()
        res[name].append(round(s.elapsed_time(e) / 3, 3))
        outs[name] = o
ops._bs_workspace = orig_ws
print(json.dumps({"ms_per_forward": res, "best": {k: min(v) for k, v in res.items()},
                  "split_vs_whole_max_abs": float((outs["split last round (shipped)"].float() - outs["every list whole"].float()).abs().max())}, indent=1))
