"""Short key axes (Sq = 32 760 queries, L = CROSS_L keys, 12 heads; the DiT cross-attention has 512): attn_w16 forced (attn_impl 300), attn_pp2 (99), the 4-wave 128-row kernel (1)."""
import os as _os
_os.environ.setdefault("FVK_PROBE_LIB", "1")  # A/B switches exist only in the measurement build (scripts/probes/libfvk_probe.so)
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fastvideo_amd import ops
S, L, H, D = 32760, int(os.environ.get("CROSS_L", "512")), 12, 128
g = torch.Generator(device="cuda").manual_seed(0)
q = torch.randn((1, S, H, D), generator=g, device="cuda").bfloat16()
k, v = (torch.randn((1, L, H, D), generator=g, device="cuda").bfloat16() for _ in range(2))
vt = ops.v_transpose(v); o = torch.empty_like(q)
fl = 4.0 * S * L * H * D
res = {}
for r in range(5):
    for i in (300, 99, 1):
        ops.set_tunable("attn_impl", i)
        ops.attn_dense(q, k, vt=vt, out=o); torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10): ops.attn_dense(q, k, vt=vt, out=o)
        e.record(); torch.cuda.synchronize()
        res.setdefault(i, []).append(s.elapsed_time(e) / 10)
ops.set_tunable("attn_impl", 0)
print(json.dumps({f"impl{i}": {"us": round(sorted(v_)[2] * 1e3, 1), "tflops": round(fl / sorted(v_)[2] / 1e9, 1)} for i, v_ in res.items()}))
