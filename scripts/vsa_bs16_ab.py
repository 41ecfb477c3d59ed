"""Block-sparse (VSA) kernel at the cfg2 geometry (624 blocks, top-125, 12 heads, real block sizes) on the model's second-layer block selection
and on uniformly random lists: attn_bs16 (round 6, shipped: "attn_impl" 0, with the split last round) vs every list whole (59) vs attn_bs16 on hardware workgroup ids (56) vs the round-1 kernel (55,
with "vsa_impl" 2 = on XCD-contiguous ids), interleaved; outputs compared with each other and with exact fp32 attention on sampled query blocks.
PMC=1: N_LAUNCH launches of ONE variant (ATTN_IMPL) for rocprofv3 --pmc passes."""
import os as _os
_os.environ.setdefault("FVK_PROBE_LIB", "1")
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fastvideo_amd import ops, wan_config as WC
from fastvideo_amd.wan_dit import WanTransformer3DModelHip
dev = torch.device("cuda")
GRID = os.environ.get("GRID", "cfg2")
lat_shape = WC.LATENT_81F_480P if GRID == "cfg2" else (1, 16, 33, 90, 160)
cfg = WC.WanConfig("vsa-only", 12, 128, 8960, 2)
sd = WC.random_state_dict(cfg, seed=0, device=dev, with_vsa_gate=True)
model = WanTransformer3DModelHip(sd, cfg.num_heads, attention="vsa", device=dev)
g = torch.Generator(device=dev).manual_seed(1)
lat = torch.randn(lat_shape, generator=g, device=dev).bfloat16()
ctx = torch.randn((1, 512, cfg.text_dim), generator=g, device=dev).bfloat16()
model.vsa_trace = []
model(lat, ctx, torch.tensor([500.0], device=dev))
mask_model, model.vsa_trace = model.vsa_trace[1], None
m = next(v for k_, v in model._vsa_cache.items() if isinstance(k_, tuple) and len(k_) == 3 and all(isinstance(x, int) for x in k_))
vbs = m["variable_block_sizes"]
n = vbs.numel()
topk = m["topk"]
S_pad = n * 64
del model, sd
q, k, v = (torch.randn((1, S_pad, 12, 128), generator=g, device=dev).bfloat16() for _ in range(3))
mask_rand = ops.topk_mask(torch.randn((1, 12, n, n), generator=g, device=dev), topk)
lists = {"model_layer1": ops.map_to_index(mask_model), "uniform_random": ops.map_to_index(mask_rand)}
VARIANTS = {"bs16 (shipped)": (0, 0), "bs16, every list whole (no split last round)": (59, 0), "bs16, hardware ids": (56, 0), "bs16, nt pieces": (57, 0), "bs16, sc0 pieces": (58, 0), "round-1 kernel": (55, 0),
            "round-1 kernel, XCD-contiguous ids": (55, 2)}


def run(idx, num):
    return ops.attn_block_sparse(q, k, v, idx, num, vbs, layout="bshd")


if os.environ.get("PMC") == "1":
    ops.set_tunable("attn_impl", int(os.environ.get("ATTN_IMPL", "0")))
    ops.set_tunable("vsa_impl", int(os.environ.get("VSA_IMPL", "0")))
    idx, num = lists[os.environ.get("LISTS", "model_layer1")]
    for _ in range(int(os.environ.get("N_LAUNCH", "3"))):
        o = run(idx, num)
    torch.cuda.synchronize()
    print("ok", float(o.float().abs().mean()))
    sys.exit(0)


def exact_blocks(idx, num, blocks):
    """fp32 softmax attention of the sampled query blocks over their lists' valid keys (block_sparse_attn_triton.py:124-158 semantics)."""
    out = {}
    vb = vbs.cpu()
    for (h, i) in blocks:
        sel = idx[0, h, i, :int(num[0, h, i])].long()
        ok = (torch.arange(64, device=dev)[None, :] < vbs[sel][:, None]).reshape(-1)
        ks = k[0, :, h].view(n, 64, 128)[sel].reshape(-1, 128)[ok].float()
        vs = v[0, :, h].view(n, 64, 128)[sel].reshape(-1, 128)[ok].float()
        s_ = (q[0, i * 64:(i + 1) * 64, h].float() @ ks.T) * 128**-0.5
        out[(h, i)] = torch.softmax(s_, dim=-1) @ vs
    return out


res = {}
for name, (idx, num) in lists.items():
    t, outs = {vn: [] for vn in VARIANTS}, {}
    for r in range(4):
        for vn, (ai, vi) in VARIANTS.items():
            ops.set_tunable("attn_impl", ai); ops.set_tunable("vsa_impl", vi)
            o = run(idx, num); torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(5): run(idx, num)
            e.record(); torch.cuda.synchronize()
            t[vn].append(round(s.elapsed_time(e) / 5, 4))
            outs[vn] = o
    ops.set_tunable("attn_impl", 0); ops.set_tunable("vsa_impl", 0)
    pairs = float(num.sum())
    blocks = [(0, 0), (3, 1), (11, n - 1), (5, n // 2), (7, 17), (2, n - 2)]
    ref = exact_blocks(idx, num, blocks)
    new, old = outs["bs16 (shipped)"], outs["round-1 kernel"]
    e_new = max((new[0, i * 64:(i + 1) * 64, h].float() - ref[(h, i)]).abs().max().item() for (h, i) in blocks)
    e_old = max((old[0, i * 64:(i + 1) * 64, h].float() - ref[(h, i)]).abs().max().item() for (h, i) in blocks)
    d = (new.float() - old.float()).abs()
    res[name] = {"ms": t, "tflops_real_pairs": {vn: round(4 * pairs * 64 * 64 * 128 / (min(v_) * 1e-3) / 1e12, 1) for vn, v_ in t.items()},
                 "bs16_ids_bit_identical": bool(torch.equal(outs["bs16, every list whole (no split last round)"], outs["bs16, hardware ids"])),
                 "split_vs_whole_max_abs": round((new.float() - outs["bs16, every list whole (no split last round)"].float()).abs().max().item(), 6),
                 "split_vs_whole_rows_changed": int(((new.float() - outs["bs16, every list whole (no split last round)"].float()).abs().amax(-1) > 0).sum()),
                 "bs16_vs_round1_max_abs": round(d.max().item(), 6), "bs16_vs_round1_mean_abs": float(f"{d.mean().item():.3g}"),
                 "finite": bool(torch.isfinite(new.float()).all()),
                 "max_abs_err_vs_exact_fp32_on_sampled_blocks": {"bs16": round(e_new, 6), "round-1 kernel": round(e_old, 6)}}
print(json.dumps(res, indent=1))
