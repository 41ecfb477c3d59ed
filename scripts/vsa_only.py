"""The shipped block-sparse (VSA) kernel alone at the contract geometry (cfg2: 624 blocks of 64, top-125, 12 heads) on the block selection a
random-init model makes in its second layer — for PMC passes (round 5; round 6: scripts/vsa_bs16_ab.py + scripts/vsa_pmc.sh).  N_LAUNCH launches (default 3)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fastvideo_amd import ops, wan_config as WC
from fastvideo_amd.wan_dit import WanTransformer3DModelHip
dev = torch.device("cuda")
cfg = WC.WanConfig("vsa-only", 12, 128, 8960, 2)
sd = WC.random_state_dict(cfg, seed=0, device=dev, with_vsa_gate=True)
model = WanTransformer3DModelHip(sd, cfg.num_heads, attention="vsa", device=dev)
g = torch.Generator(device=dev).manual_seed(1)
lat = torch.randn(WC.LATENT_81F_480P, generator=g, device=dev).bfloat16()
ctx = torch.randn((1, 512, cfg.text_dim), generator=g, device=dev).bfloat16()
model.vsa_trace = []
model(lat, ctx, torch.tensor([500.0], device=dev))
mask, model.vsa_trace = model.vsa_trace[1], None
m = next(v for k_, v in model._vsa_cache.items() if isinstance(k_, tuple) and len(k_) == 3 and all(isinstance(x, int) for x in k_))
vbs = m["variable_block_sizes"]
S_pad = vbs.numel() * 64
q, k, v = (torch.randn((1, S_pad, 12, 128), generator=g, device=dev).bfloat16() for _ in range(3))
idx, num = ops.map_to_index(mask)
torch.cuda.synchronize()
for _ in range(int(os.environ.get("N_LAUNCH", "3"))):
    o = ops.attn_block_sparse(q, k, v, idx, num, vbs, layout="bshd")
torch.cuda.synchronize()
blocks = float(num.sum())
print("ok", float(o.float().abs().mean()), "selected (q block, kv block) pairs", blocks, "algorithmic FLOP per launch", blocks * 4 * 64 * 64 * 128,
      "MFMA 32x32x16 instructions per launch", blocks * 2 * 64 * 64 * 128 / (32 * 32 * 16))
