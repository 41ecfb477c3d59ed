"""Dense attention at sequence-parallel per-rank shapes: the 8-wave 256-row kernel (attn_impl 0, shipped) vs the 4-wave 128-row kernel
(attn_impl 1) — does the smaller workgroup fill 256 CUs better when a rank holds only 192-384 query blocks?"""
import os as _os
_os.environ.setdefault("FVK_PROBE_LIB", "1")  # A/B switches exist only in the measurement build (scripts/probes/libfvk_probe.so)
import json, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fastvideo_amd import ops
S, H, D = 32760, 12, 128
g = torch.Generator(device="cuda").manual_seed(0)
rn = lambda *s: torch.randn(s, generator=g, device="cuda").bfloat16()
def t(fn, n=6):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
out = {}
for P in (2, 4, 8):
    G = math.gcd(H, P); Sl = (S + P - 1) // P; hg = H // G
    q, k, v = rn(1, G * Sl, hg, D), rn(1, S, hg, D), rn(1, S, hg, D)
    vt = ops.v_transpose(v)
    r = {}
    for impl in (0, 1, 0, 1):
        ops.set_tunable("attn_impl", impl)
        r.setdefault(f"impl{impl}_us", []).append(round(t(lambda: ops.attn_dense(q, k, vt=vt, layout="bshd")), 1))
    ops.set_tunable("attn_impl", 0)
    out[f"P{P}"] = dict(q_rows=G * Sl, heads=hg, wg256=((G * Sl + 255) // 256) * hg, **r)
print(json.dumps(out))
