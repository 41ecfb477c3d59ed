"""One attention kernel variant at the cfg2 shape for PMC passes: ATTN_IMPL=<attn_impl> (0 = the shipped default attn_w16; 200 = attn_w64; 99 = attn_pp2), N_LAUNCH launches."""
import os as _os
_os.environ.setdefault("FVK_PROBE_LIB", "1")
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fastvideo_amd import ops
S, H, D = 32760, 12, 128
g = torch.Generator(device="cuda").manual_seed(0)
q, k, v = (torch.randn((1, S, H, D), generator=g, device="cuda").bfloat16() for _ in range(3))
vt = ops.v_transpose(v)
ops.set_tunable("attn_impl", int(os.environ.get("ATTN_IMPL", "0")))
for _ in range(int(os.environ.get("N_LAUNCH", "3"))):
    o = ops.attn_dense(q, k, vt=vt)
torch.cuda.synchronize()
print("ok", float(o.float().abs().mean()))
