"""Block-sparse (VSA) attention at the contract geometry (cfg2: 624 blocks, top-125, 12 heads): the two-list workgroup against the union walk
(round 4) on the block selections the MODEL makes — layer by layer of a 6-layer random-init model, randn latent — and on uniformly random
selections.  Interleaved timing, outputs compared for equality.  usage: python scripts/vsa_union_ab.py"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fastvideo_amd import ops, wan_config as WC
from fastvideo_amd.wan_dit import WanTransformer3DModelHip
dev = torch.device("cuda")
cfg = WC.WanConfig("vsa-ab", 12, 128, 8960, 6)
sd = WC.random_state_dict(cfg, seed=0, device=dev, with_vsa_gate=True)
model = WanTransformer3DModelHip(sd, cfg.num_heads, attention="vsa", device=dev)
g = torch.Generator(device=dev).manual_seed(1)
lat = torch.randn(WC.LATENT_81F_480P, generator=g, device=dev).bfloat16()
ctx = torch.randn((1, 512, cfg.text_dim), generator=g, device=dev).bfloat16()
model.vsa_trace = []
model(lat, ctx, torch.tensor([500.0], device=dev))
masks, model.vsa_trace = model.vsa_trace, None
m = next(v for k_, v in model._vsa_cache.items() if isinstance(k_, tuple) and len(k_) == 3 and all(isinstance(x, int) for x in k_))
vbs = m["variable_block_sizes"]
nb = vbs.numel()
S_pad = nb * 64
q, k, v = (torch.randn((1, S_pad, 12, 128), generator=g, device=dev).bfloat16() for _ in range(3))
rand_mask = torch.zeros((1, 12, nb, nb), dtype=torch.bool, device=dev)
rand_mask.scatter_(-1, torch.rand((1, 12, nb, nb), generator=g, device=dev).topk(125, dim=-1).indices, True)
for name, mask in [(f"layer {i}", mk) for i, mk in enumerate(masks)] + [("uniformly random", rand_mask)]:
    idx, num = ops.map_to_index(mask)
    res = {}
    outs = {}
    for rep in range(3):
        for uni in (False, True):
            o = ops.attn_block_sparse(q, k, v, idx, num, vbs, layout="bshd", pair_union=uni); torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(5): o = ops.attn_block_sparse(q, k, v, idx, num, vbs, layout="bshd", pair_union=uni)
            e.record(); torch.cuda.synchronize()
            res.setdefault(uni, []).append(s.elapsed_time(e) / 5)
            outs[uni] = o
    a_, b_ = mask[0, :, 0::2].float(), mask[0, :, 1::2].float()
    n = min(a_.shape[1], b_.shape[1])
    saved = 1 - ((a_[:, :n] + b_[:, :n]) > 0).sum().item() / (a_[:, :n].sum().item() + b_[:, :n].sum().item())
    print(f"{name:18s}: two lists {sorted(res[False])[1]:.3f} ms, union walk (incl. the merge kernel) {sorted(res[True])[1]:.3f} ms "
          f"({sorted(res[False])[1] / sorted(res[True])[1]:.3f}x), tiles saved {saved:.3f}, outputs equal {torch.equal(outs[False], outs[True])}", flush=True)
