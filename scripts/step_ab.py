"""Same-box A/B of the contract DiT step under kernel generations (measurement build: FVK_PROBE_LIB=1): interleaved forwards with
(gemm_impl, attn_impl) = (228, 99) the round-2 kernels (gemm_ph + attn_pp2), (228, 200) gemm_ph + attn_w64, (0, 200) gemm_w1 + attn_w64,
(0, 0) the shipped pair (gemm_w1 + attn_w16).  Boxes differ by up to 10 % in sustained clock: only numbers from ONE run compare."""
import os as _os
_os.environ.setdefault("FVK_PROBE_LIB", "1")
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fastvideo_amd import ops, wan_config as WC
from fastvideo_amd.wan_dit import WanTransformer3DModelHip

dev = torch.device("cuda", 0)
cfg = WC.WAN21_T2V_1_3B
sd = WC.random_state_dict(cfg, seed=0, device=dev)
model = WanTransformer3DModelHip(sd, cfg.num_heads, cfg.head_dim, cfg.patch_size, cfg.eps, cfg.freq_dim, attention="dense", device=dev)
del sd
g = torch.Generator(device=dev).manual_seed(1)
latent = torch.randn(WC.LATENT_81F_480P, generator=g, device=dev).bfloat16()
ctx = torch.randn((1, 512, cfg.text_dim), generator=g, device=dev).bfloat16()
ts = torch.tensor([500.0], device=dev)
arms = [("r2: gemm_ph + attn_pp2", 228, 99), ("gemm_ph + attn_w64", 228, 200), ("gemm_w1 + attn_w64", 0, 200), ("shipped: gemm_w1 + attn_w16", 0, 0)]
res = {a[0]: [] for a in arms}
attn = {a[0]: [] for a in arms}
for r in range(4):
    for name, gi, ai in arms:
        ops.set_tunable("gemm_impl", gi); ops.set_tunable("attn_impl", ai)
        model(latent, ctx, ts); torch.cuda.synchronize()
        model.attn_events = []
        t0 = time.perf_counter()
        for _ in range(3): model(latent, ctx, ts)
        torch.cuda.synchronize()
        res[name].append((time.perf_counter() - t0) / 3 * 1e3)
        ev, model.attn_events = model.attn_events, None
        attn[name].append(sum(e0.elapsed_time(e1) for e0, e1, *_ in ev) / len(ev))
ops.set_tunable("gemm_impl", 0); ops.set_tunable("attn_impl", 0)
for name, v in res.items():
    print(json.dumps({"arm": name, "ms_per_step_median": round(sorted(v)[len(v) // 2], 2), "all": [round(x, 2) for x in v],
                      "self_attention_ms_per_launch": [round(x, 4) for x in attn[name]]}))
# the same two attention kernels on randn inputs of the same shape, interleaved, in this process
S, H = 32760, cfg.num_heads
q, k, v = (torch.randn(1, S, H, 128, device=dev).bfloat16() for _ in range(3))
vt = ops.v_transpose(v); o = torch.empty_like(q)
for ai in (200, 0, 200, 0):
    ops.set_tunable("attn_impl", ai)
    ops.attn_dense(q, k, vt=vt, out=o); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): ops.attn_dense(q, k, vt=vt, out=o)
    e1.record(); torch.cuda.synchronize()
    print(json.dumps({"randn attention, attn_impl": ai, "ms": round(e0.elapsed_time(e1) / 5, 4)}))
ops.set_tunable("attn_impl", 0)
