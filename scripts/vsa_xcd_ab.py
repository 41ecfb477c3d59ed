"""Block-sparse (VSA) kernel at the cfg2 geometry on the block selection a random-init model makes in its second layer AND on uniformly random
lists: the round-robin workgroup deal (hardware ids, "vsa_impl" 0) vs the XCD-contiguous deal ("vsa_impl" 2, attn_fwd.hip), interleaved.
PMC=1: N_LAUNCH launches of ONE variant (VSA_IMPL) for a rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum / FETCH_SIZE pass."""
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
mask_model, model.vsa_trace = model.vsa_trace[1], None
m = next(v for k_, v in model._vsa_cache.items() if isinstance(k_, tuple) and len(k_) == 3 and all(isinstance(x, int) for x in k_))
vbs = m["variable_block_sizes"]
n = vbs.numel()
S_pad = n * 64
q, k, v = (torch.randn((1, S_pad, 12, 128), generator=g, device=dev).bfloat16() for _ in range(3))
mask_rand = ops.topk_mask(torch.randn((1, 12, n, n), generator=g, device=dev), 125)
lists = {"model_layer1": ops.map_to_index(mask_model), "uniform_random": ops.map_to_index(mask_rand)}
if os.environ.get("PMC") == "1":
    ops.set_tunable("vsa_impl", int(os.environ.get("VSA_IMPL", "0")))
    idx, num = lists[os.environ.get("LISTS", "model_layer1")]
    for _ in range(int(os.environ.get("N_LAUNCH", "3"))):
        o = ops.attn_block_sparse(q, k, v, idx, num, vbs, layout="bshd")
    torch.cuda.synchronize()
    print("ok", float(o.float().abs().mean()))
    sys.exit(0)
res = {}
for name, (idx, num) in lists.items():
    fn = lambda: ops.attn_block_sparse(q, k, v, idx, num, vbs, layout="bshd")
    t, outs = {0: [], 2: []}, {}
    for r in range(4):
        for impl in (0, 2):
            ops.set_tunable("vsa_impl", impl)
            o = fn(); torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(5): fn()
            e.record(); torch.cuda.synchronize()
            t[impl].append(round(s.elapsed_time(e) / 5, 4))
            outs[impl] = o
    ops.set_tunable("vsa_impl", 0)
    pairs = float(num.sum())
    res[name] = {"round_robin_ms": t[0], "xcd_contiguous_ms": t[2], "bit_identical": bool(torch.equal(outs[0], outs[2])),
                 "tflops_real_pairs_round_robin": round(4 * pairs * 64 * 64 * 128 / (min(t[0]) * 1e-3) / 1e12, 1),
                 "tflops_real_pairs_xcd_contiguous": round(4 * pairs * 64 * 64 * 128 / (min(t[2]) * 1e-3) / 1e12, 1)}
print(json.dumps(res, indent=1))
