import os
os.environ["FVK_PROBE_LIB"]="1"
import sys, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fastvideo_amd import ops
for S in (9216, 4680, 16380):
    H, D = 12, 128
    q, k, v = (torch.randn(1, S, H, D, device="cuda").bfloat16() for _ in range(3))
    vt = ops.v_transpose(v); o = torch.empty_like(q)
    fl = 4.0 * S * S * H * D
    res = {}
    for r in range(5):
        for i in (99, 0):
            ops.set_tunable("attn_impl", i)
            ops.attn_dense(q, k, vt=vt, out=o, key_splits=1); torch.cuda.synchronize()
            s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s_.record()
            for _ in range(10): ops.attn_dense(q, k, vt=vt, out=o, key_splits=1)
            e_.record(); torch.cuda.synchronize()
            res.setdefault(i, []).append(s_.elapsed_time(e_) / 10)
    ops.set_tunable("attn_impl", 0)
    print(S, {i: (round(sorted(v_)[2]*1e3,1), round(fl/sorted(v_)[2]/1e9,1)) for i, v_ in res.items()})
