"""Where attn_bs16's time goes: timing ablations of the kernel at the cfg2 lists of the model's second layer (results of the ablated variants are
WRONG; only their time means something).  "attn_impl" 0 = shipped, 66 = only every second LDS-DMA piece issued (half the L2 -> LDS traffic),
67 = no softmax VALU, 69 = no counted vmcnt waits, 68 = half the pieces + no softmax, 72 = MFMAs + fragment reads + half the pieces only."""
import os as _os
_os.environ.setdefault("FVK_PROBE_LIB", "1")
import json, os, sys
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
n = vbs.numel()
del model, sd
q, k, v = (torch.randn((1, n * 64, 12, 128), generator=g, device=dev).bfloat16() for _ in range(3))
idx, num = ops.map_to_index(mask)
V = {"shipped": 0, "half the LDS-DMA pieces": 66, "no softmax VALU": 67, "no counted waits": 69, "half the pieces, no softmax": 68,
     "MFMAs + fragment reads + half the pieces": 72}
t = {k_: [] for k_ in V}
for r in range(4):
    for name, impl in V.items():
        ops.set_tunable("attn_impl", impl)
        ops.attn_block_sparse(q, k, v, idx, num, vbs, layout="bshd"); torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(5): ops.attn_block_sparse(q, k, v, idx, num, vbs, layout="bshd")
        e.record(); torch.cuda.synchronize()
        t[name].append(round(s.elapsed_time(e) / 5, 4))
ops.set_tunable("attn_impl", 0)
pairs = float(num.sum())
print(json.dumps({"ms (incl. the 0.04-ms V^T layout pass)": t, "best_ms": {k_: min(v_) for k_, v_ in t.items()},
                  "mfma_only_floor_ms_at_2GHz": round(pairs * 136 * 16 / 1024 / 2.0e9 * 1e3 * (8 / 7.3125), 3)}, indent=1))
