"""attn_w64 (attn_impl 200) vs attn_w16 (300) at the cfg2 self-attention shape, interleaved, sustained (30 launches per sample): boxes differ."""
import os as _os
_os.environ.setdefault("FVK_PROBE_LIB", "1")
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fastvideo_amd import ops
S, H, D = 32760, 12, 128
q, k, v = (torch.randn(1, S, H, D, device="cuda").bfloat16() for _ in range(3))
vt = ops.v_transpose(v); o = torch.empty_like(q)
fl = 4.0 * S * S * H * D
res = {200: [], 300: []}
for r in range(5):
    for i in (200, 300):
        ops.set_tunable("attn_impl", i)
        ops.attn_dense(q, k, vt=vt, out=o); torch.cuda.synchronize()
        s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s_.record()
        for _ in range(30): ops.attn_dense(q, k, vt=vt, out=o)
        e_.record(); torch.cuda.synchronize()
        res[i].append(round(fl / (s_.elapsed_time(e_) / 30) / 1e9, 1))
ops.set_tunable("attn_impl", 0)
print(json.dumps({"w64_tflops": res[200], "w16_tflops": res[300]}))
