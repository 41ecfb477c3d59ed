#!/bin/bash
# round 3: attn_w16 vs attn_w64 timing ablations (x11 no DMA, x14 no softmax VALU, x17 neither + no barrier), interleaved
mkdir -p gpurun_out/r3n; cd /root/repo
FVK_PROBE_LIB=1 timeout 900 python - > gpurun_out/r3n/ablate.log 2>&1 <<'PY'
import torch, json
from fastvideo_amd import ops
S, H, D = 32760, 12, 128
q, k, v = (torch.randn(1, S, H, D, device="cuda").bfloat16() for _ in range(3))
vt = ops.v_transpose(v); o = torch.empty_like(q)
fl = 4.0 * S * S * H * D
impls = [300, 318, 311, 314, 317, 312]
res = {i: [] for i in impls}
for r in range(4):
    for i in impls:
        ops.set_tunable("attn_impl", i)
        ops.attn_dense(q, k, vt=vt, out=o); torch.cuda.synchronize()
        s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s_.record()
        for _ in range(5): ops.attn_dense(q, k, vt=vt, out=o)
        e_.record(); torch.cuda.synchronize()
        res[i].append(s_.elapsed_time(e_) / 5)
ops.set_tunable("attn_impl", 0)
for i in impls:
    ms = sorted(res[i])
    print(json.dumps({"impl": i, "ms_med": round(ms[len(ms)//2], 4), "tflops_med": round(fl / ms[len(ms)//2] / 1e9, 1), "tflops_all": [round(fl / m / 1e9, 1) for m in res[i]]}))
PY
cat gpurun_out/r3n/ablate.log
FVK_PROBE_LIB=1 timeout 300 python -m pytest scripts/probes/variant_tests.py -q -k "test_attn_dense_impls" 2>&1 | tail -2
