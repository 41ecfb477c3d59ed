"""Bit-identity of the attn_pp2 schedule variants (attn_impl 100 + k) against the 4-wave kernel on small and ragged shapes, then an
interleaved timing at the cfg2 shape.  usage: python scripts/attn_variants_check.py 103 105 107 109"""
import os as _os
_os.environ.setdefault("FVK_PROBE_LIB", "1")  # A/B switches exist only in the measurement build (scripts/probes/libfvk_probe.so)
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fastvideo_amd import ops
impls = [int(x) for x in sys.argv[1:]] or [0, 103]
g = torch.Generator(device="cuda").manual_seed(0)
ok = True
for (B, H, Sq, Skv) in [(1, 2, 256, 128), (1, 1, 300, 1), (2, 3, 700, 1999), (1, 2, 1030, 257), (1, 1, 512, 4096)]:
    q, k, v = (torch.randn((B, s_, H, 128), generator=g, device="cuda").bfloat16() for s_ in (Sq, Skv, Skv))
    ops.set_tunable("attn_impl", 1)
    ref = ops.attn_dense(q, k, v)
    for i in impls:
        ops.set_tunable("attn_impl", i)
        o = ops.attn_dense(q, k, v)
        torch.cuda.synchronize()
        d = (o.float() - ref.float()).abs().max().item()
        same = d < 2e-2
        ok &= same
        print(f"shape {(B, H, Sq, Skv)} impl {i}: max|diff| vs 4-wave kernel {d:.3g} {'ok' if same else 'MISMATCH'}", flush=True)
ops.set_tunable("attn_impl", 0)
print("ALL_OK" if ok else "FAILED", flush=True)
if not ok:
    sys.exit(1)
S, H = 32760, 12
q, k, v = (torch.randn((1, S, H, 128), generator=g, device="cuda").bfloat16() for _ in range(3))
vt = ops.v_transpose(v); o = torch.empty_like(q)
fl = 4.0 * S * S * H * 128
res = {i: [] for i in impls}; first = None
for r in range(6):
    for i in impls:
        ops.set_tunable("attn_impl", i)
        ops.attn_dense(q, k, vt=vt, out=o); torch.cuda.synchronize()
        if r == 0:
            if first is None: first = o.clone()
            else: print("impl", i, "identical to first:", bool(torch.equal(o, first)), flush=True)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(3): ops.attn_dense(q, k, vt=vt, out=o)
        e.record(); torch.cuda.synchronize()
        res[i].append(s.elapsed_time(e) / 3)
ops.set_tunable("attn_impl", 0)
for i in impls:
    m = sorted(res[i])[len(res[i]) // 2]
    print(json.dumps({"impl": i, "ms": round(m, 4), "tflops": round(fl / m / 1e9, 1), "best": round(fl / min(res[i]) / 1e9, 1)}))
