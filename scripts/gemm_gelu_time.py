import os, sys, json
sys.path.insert(0, "/root/repo")
import torch
from fastvideo_amd import ops
S, d, F = 32760, 1536, 8960
x = torch.randn(S, d, device="cuda").bfloat16(); w = (torch.randn(F, d, device="cuda") * d**-0.5).bfloat16(); b = torch.randn(F, device="cuda").bfloat16()
out = torch.empty(S, F, device="cuda", dtype=torch.bfloat16)
r = {}
for rep in range(4):
    for name, kw in (("gelu", dict(epilogue=ops.EPI_GELU_TANH)), ("plain", {})):
        ops.gemm(x, w, b, out=out, **kw); torch.cuda.synchronize()
        s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s_.record()
        for _ in range(10): ops.gemm(x, w, b, out=out, **kw)
        e_.record(); torch.cuda.synchronize()
        r.setdefault(name, []).append(round(s_.elapsed_time(e_) / 10, 4))
print(json.dumps(r))
