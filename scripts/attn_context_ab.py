"""Why does attn_w16 lose to attn_w64 INSIDE the step on some boxes while winning in isolation?  One process: (1) in-step self-attention time per
arm, (2) the same kernels on captured layer-15 inputs back to back, (3) with a GEMM between launches, (4) with an idle gap between launches."""
import os as _os
_os.environ.setdefault("FVK_PROBE_LIB", "1")
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fastvideo_amd import ops, wan_config as WC
from fastvideo_amd.wan_dit import WanTransformer3DModelHip
import fastvideo_amd.wan_dit as WD
dev = torch.device("cuda", 0)
cfg = WC.WAN21_T2V_1_3B
sd = WC.random_state_dict(cfg, seed=0, device=dev)
model = WanTransformer3DModelHip(sd, cfg.num_heads, cfg.head_dim, cfg.patch_size, cfg.eps, cfg.freq_dim, attention="dense", device=dev)
del sd
g = torch.Generator(device=dev).manual_seed(1)
latent = torch.randn(WC.LATENT_81F_480P, generator=g, device=dev).bfloat16()
ctx = torch.randn((1, 512, cfg.text_dim), generator=g, device=dev).bfloat16()
ts = torch.tensor([500.0], device=dev)
cap = []
orig = ops.attn_dense
def spy(q, k, v=None, vt=None, **kw):
    if k.shape[1] > 4096 and len(cap) < 16:
        keep = lambda t: torch.empty_strided(t.shape, t.stride(), dtype=t.dtype, device=t.device).copy_(t)
        cap.append((keep(q), keep(k), keep(vt), kw))
    return orig(q, k, v, vt=vt, **kw)
ops.attn_dense = spy; WD.ops.attn_dense = spy
model(latent, ctx, ts); torch.cuda.synchronize()
ops.attn_dense = orig; WD.ops.attn_dense = orig
out = {}
# (1) in step
for r in range(2):
    for name, ai in (("w64", 200), ("w16", 0)):
        ops.set_tunable("attn_impl", ai)
        model(latent, ctx, ts); torch.cuda.synchronize()
        model.attn_events = []
        for _ in range(2): model(latent, ctx, ts)
        torch.cuda.synchronize()
        ev, model.attn_events = model.attn_events, None
        out.setdefault("in_step_" + name, []).append(round(sum(e0.elapsed_time(e1) for e0, e1, *_ in ev) / len(ev), 4))
q, k, vt, kw = cap[15]
x = torch.randn(32760, 1536, device=dev).bfloat16(); w = torch.randn(4608, 1536, device=dev).bfloat16() * 0.02; y = torch.empty(32760, 4608, device=dev, dtype=torch.bfloat16)
def run(mode, ai, n=12):
    ops.set_tunable("attn_impl", ai)
    evs = []
    for i in range(n + 2):
        if mode == "gemm":
            for _ in range(4): ops.gemm(x, w, None, out=y)
        elif mode == "idle":
            torch.cuda.synchronize(); time.sleep(0.002)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); orig(q, k, vt=vt, **kw); e1.record()
        if i >= 2: evs.append((e0, e1))
    torch.cuda.synchronize()
    return round(sum(a.elapsed_time(b) for a, b in evs) / len(evs), 4)
for r in range(2):
    for mode in ("back_to_back", "gemm", "idle"):
        for name, ai in (("w64", 200), ("w16", 300)):
            out.setdefault(f"{mode}_{name}", []).append(run(mode, ai))
ops.set_tunable("attn_impl", 0)
print(json.dumps(out))
