"""How much could a UNION walk of neighbouring query blocks' KV lists save the block-sparse (VSA) kernel?  The kernel sits on the L2 -> LDS ingest
rate of its access pattern: its time follows the bytes it fetches, so two 64-row query blocks sharing one walk over the union of
their top-k lists would cut the time by 1 - |A u B| / (|A| + |B|).  This measures that ratio on the contract geometry (cfg2: 21 x 30 x 52 tokens, 624
blocks, top-125) for (a) a randn latent through random-init weights (the benchmark's input) and (b) a SMOOTH latent (coarse noise upsampled 4x in
every axis: neighbouring tokens nearly equal, the regime of real video) through the same weights — per layer, for pairs of consecutive query
blocks in tile-major order.  usage: python scripts/vsa_overlap.py [--layers 6]"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fastvideo_amd import wan_config as WC
from fastvideo_amd.wan_dit import WanTransformer3DModelHip

ap = argparse.ArgumentParser()
ap.add_argument("--layers", type=int, default=6)
args = ap.parse_args()
cfg = WC.WanConfig("vsa-overlap", 12, 128, 8960, args.layers)
dev = torch.device("cuda")
sd = WC.random_state_dict(cfg, seed=0, device=dev, with_vsa_gate=True)
model = WanTransformer3DModelHip(sd, cfg.num_heads, attention="vsa", device=dev)
g = torch.Generator(device=dev).manual_seed(1)
shape = WC.LATENT_81F_480P
ctx = torch.randn((1, 512, cfg.text_dim), generator=g, device=dev).bfloat16()
ts = torch.tensor([500.0], device=dev)
coarse = torch.randn((1, 16, 6, 15, 26), generator=g, device=dev)
smooth = torch.nn.functional.interpolate(coarse, size=shape[2:], mode="trilinear", align_corners=False)
smooth = smooth / smooth.std()
out = {}
for name, lat in (("randn", torch.randn(shape, generator=g, device=dev)), ("smooth", smooth)):
    model.vsa_trace = []
    model(lat.bfloat16(), ctx, ts)
    masks, model.vsa_trace = model.vsa_trace, None
    rows = []
    for li, m in enumerate(masks):                     # [1, H, nq, nk] bool
        m = m[0].float()
        a, b = m[:, 0::2], m[:, 1::2]
        n = min(a.shape[1], b.shape[1])
        a, b = a[:, :n], b[:, :n]
        inter = (a * b).sum(-1)
        union = ((a + b) > 0).float().sum(-1)
        tot = a.sum(-1) + b.sum(-1)
        rows.append(dict(layer=li, topk=int(m[0, 0].sum().item()), mean_overlap_frac=round((inter / a.sum(-1)).mean().item(), 4),
                         bytes_saved_by_union_walk=round((1 - union.sum() / tot.sum()).item(), 4)))
    out[name] = rows
    print(name, json.dumps(rows))
print(json.dumps({k: {"mean_bytes_saved": round(sum(r["bytes_saved_by_union_walk"] for r in v) / len(v), 4)} for k, v in out.items()}))
