"""Run only the dominant kernels at cfg2 shapes (for PMC passes): dense self-attention and the FFN-in GEMM."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fastvideo_amd import ops
S, H, D, d, F = 32760, 12, 128, 1536, 8960
g = torch.Generator(device="cuda").manual_seed(0)
q, k, v = (torch.randn((1, S, H, D), generator=g, device="cuda").bfloat16() for _ in range(3))
vt = ops.v_transpose(v)
x = torch.randn((S, d), generator=g, device="cuda").bfloat16(); w = (torch.randn((F, d), generator=g, device="cuda") * d**-0.5).bfloat16()
b = torch.randn((F,), generator=g, device="cuda").bfloat16()
n = int(os.environ.get("N_LAUNCH", "2"))
for _ in range(n):
    o = ops.attn_dense(q, k, vt=vt)
    y = ops.gemm(x, w, b, epilogue=ops.EPI_GELU_TANH)
torch.cuda.synchronize()
print("ok", float(o.float().abs().mean()), float(y.float().abs().mean()))
