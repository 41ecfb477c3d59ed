"""Is the dense attention kernel power-bound?  Same binary, same shape (cfg2 self-attention), same instruction stream: random bf16 inputs
vs all-zero inputs (no operand toggling in the matrix pipe).  guides/MI355X_MICROARCH.md reports +19 % for a tuned attention kernel."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fastvideo_amd import ops
S, H, D = 32760, 12, 128
g = torch.Generator(device="cuda").manual_seed(0)
fl = 4.0 * S * S * H * D
out = {}
def run(name, q, k, v):
    vt = ops.v_transpose(v); o = torch.empty_like(q)
    ts = []
    for r in range(4):
        ops.attn_dense(q, k, vt=vt, out=o); torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(5): ops.attn_dense(q, k, vt=vt, out=o)
        e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) / 5)
    m = sorted(ts)[1]
    out[name] = {"ms": round(m, 4), "tflops": round(fl / m / 1e9, 1)}
rn = lambda: torch.randn((1, S, H, D), generator=g, device="cuda").bfloat16()
q, k, v = rn(), rn(), rn()
z = torch.zeros_like(q)
for name, args in (("random", (q, k, v)), ("zeros", (z, z, z)), ("random_again", (q, k, v)), ("small_values_1e-3", (q * 1e-3, k * 1e-3, v * 1e-3))):
    run(name, *args)
print(json.dumps(out))
