"""The dense self-attention of the contract model on ITS OWN q / k / v (captured from a forward, layers 0 / 15 / 29): attn_w64 (200) vs attn_w16
(300) vs attn_w16 without the exact recompute (319), interleaved — against the same kernels on randn inputs of the same shape."""
import os as _os
_os.environ.setdefault("FVK_PROBE_LIB", "1")
import json, os, sys
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
cap = []
orig = ops.attn_dense
def spy(q, k, v=None, vt=None, **kw):
    if k.shape[1] > 4096:
        keep = lambda t: torch.empty_strided(t.shape, t.stride(), dtype=t.dtype, device=t.device).copy_(t)
        cap.append((keep(q), keep(k), keep(vt), kw))
    return orig(q, k, v, vt=vt, **kw)
ops.attn_dense = spy
import fastvideo_amd.wan_dit as WD
WD.ops.attn_dense = spy
model(latent, ctx, ts); torch.cuda.synchronize()
ops.attn_dense = orig; WD.ops.attn_dense = orig
print("captured", len(cap), "self-attention calls; q stride", cap[0][0].stride(), "scores c2*(q.k) std per layer:",
      [round(float((cap[i][0][0, :2048, 0].float() @ cap[i][1][0, :2048, 0].float().t()).std()) * 128**-0.5 * 1.4427, 2) for i in (0, 15, 29)])
S, H = cap[0][0].shape[1], cap[0][0].shape[2]
qr, kr, vr = (torch.randn(1, S, H, 128, device=dev).bfloat16() for _ in range(3))
sets = {"layer0": cap[0], "layer15": cap[15], "layer29": cap[29], "randn": (qr, kr, ops.v_transpose(vr), {"scale": 128**-0.5, "layout": "bshd"})}
for name, (q, k, vt, kw) in sets.items():
    res = {}
    for r in range(3):
        for ai in (200, 300, 319):
            ops.set_tunable("attn_impl", ai)
            orig(q, k, vt=vt, **kw); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): o = orig(q, k, vt=vt, **kw)
            e1.record(); torch.cuda.synchronize()
            res.setdefault(ai, []).append(round(e0.elapsed_time(e1) / 5, 4))
    ops.set_tunable("attn_impl", 200); o64 = orig(q, k, vt=vt, **kw).float()
    ops.set_tunable("attn_impl", 300); o16 = orig(q, k, vt=vt, **kw).float()
    ops.set_tunable("attn_impl", 319); o16n = orig(q, k, vt=vt, **kw).float()
    ops.set_tunable("attn_impl", 0)
    print(json.dumps({"data": name, "ms w64 / w16 / w16-no-redo": [sorted(res[a])[1] for a in (200, 300, 319)],
                      "max|w16-w64|": round(float((o16 - o64).abs().max()), 5), "rows changed by the redo": int(((o16 - o16n).abs().amax(-1) > 0).sum())}))
