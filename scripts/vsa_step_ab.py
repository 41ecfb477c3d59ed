"""The VSA forward at the contract geometry with this round's host-visible switches, interleaved in one process: the combine pass fused into the
sparse kernel's store (kernel_api.FUSE_SPARSE_COMBINE) and the split last round of attn_bs16 (ops.attn_block_sparse / vsa_sparse_combine workspace)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fastvideo_amd import ops, kernel_api, wan_config as WC
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
modes = {"fused combine + split last round (shipped)": (True, True), "two launches + split": (False, True), "fused, every list whole": (True, False),
         "two launches, every list whole (the round's start)": (False, False)}
res = {k: [] for k in modes}
outs = {}
for rep in range(int(os.environ.get("REPS", "4"))):
    for name, (fuse, split) in modes.items():
        kernel_api.FUSE_SPARSE_COMBINE = fuse
        ops._bs_workspace = orig_ws if split else (lambda a, m, d: (None, 0))
        o = model(lat, ctx, t); torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(3): model(lat, ctx, t)
        e.record(); torch.cuda.synchronize()
        res[name].append(round(s.elapsed_time(e) / 3, 3))
        outs[name] = o
base = outs["two launches, every list whole (the round's start)"]
print(json.dumps({"ms_per_forward": res, "best": {k: min(v) for k, v in res.items()},
                  "fused_equals_two_launches_bitwise": bool(torch.equal(outs["fused, every list whole"], base)),
                  "split_vs_whole_max_abs": float((outs["fused combine + split last round (shipped)"].float() - base.float()).abs().max())}, indent=1))
