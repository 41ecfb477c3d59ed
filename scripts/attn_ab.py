"""Interleaved A/B of the dense attention variants at the cfg2 shape (S=32760, H=12): python scripts/attn_ab.py [impl ...]"""
import os as _os
_os.environ.setdefault("FVK_PROBE_LIB", "1")  # A/B switches exist only in the measurement build (scripts/probes/libfvk_probe.so)
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fastvideo_amd import ops
impls = [int(x) for x in sys.argv[1:]] or [0, 4]
S, H, D = 32760, 12, 128
q, k, v = (torch.randn(1, S, H, D, device="cuda").bfloat16() for _ in range(3))
vt = ops.v_transpose(v); o = torch.empty_like(q)
fl = 4.0 * S * S * H * D
ref = None
res = {i: [] for i in impls}
for r in range(int(os.environ.get("AB_ROUNDS", "4"))):
    for i in impls:
        ops.set_tunable("attn_impl", i)
        ops.attn_dense(q, k, vt=vt, out=o); torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(3): ops.attn_dense(q, k, vt=vt, out=o)
        e.record(); torch.cuda.synchronize()
        res[i].append(s.elapsed_time(e) / 3)
        if r == 0:
            if ref is None: ref = o.clone()
            else: print("impl", i, "max abs diff vs first impl", (o.float() - ref.float()).abs().max().item())
ops.set_tunable("attn_impl", 0)
for i in impls:
    m = sorted(res[i])[len(res[i]) // 2]
    print(json.dumps({"impl": i, "ms": round(m, 4), "tflops": round(fl / m / 1e9, 1), "best": round(fl / min(res[i]) / 1e9, 1)}))
