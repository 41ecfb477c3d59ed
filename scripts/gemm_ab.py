"""Interleaved timing of the GEMM kernels at the cfg2 shapes: python scripts/gemm_ab.py [impl ...] (0 stream, 2 ping-pong, 1 128x128)"""
import os as _os
_os.environ.setdefault("FVK_PROBE_LIB", "1")  # A/B switches exist only in the measurement build (scripts/probes/libfvk_probe.so)
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fastvideo_amd import ops
impls = [int(x) for x in sys.argv[1:]] or [0, 2]
S, d, F = 32760, 1536, 8960
S14, d14, F14 = 75600, 5120, 13824
for name, M, N, K in (("qkv", S, 3 * d, d), ("out", S, d, d), ("ffn_in", S, F, d), ("ffn_out", S, d, F), ("8k", 8192, 8192, 8192),
                      ("14b_qkv", S14, 3 * d14, d14), ("14b_ffn_in", S14, F14, d14), ("14b_ffn_out", S14, d14, F14)):
    a = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") * K**-0.5).bfloat16()
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    res = {i: [] for i in impls}
    for r in range(3):
        for i in impls:
            ops.set_tunable("gemm_impl", i)
            ops.gemm(a, w, None, out=out); torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(5): ops.gemm(a, w, None, out=out)
            e.record(); torch.cuda.synchronize()
            res[i].append(s.elapsed_time(e) / 5)
    ops.set_tunable("gemm_impl", 0)
    tl = []  # the vendor library through torch (x @ w^T, plain epilogue)
    for r in range(3):
        torch.matmul(a, w.t(), out=out); torch.cuda.synchronize()
        s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s_.record()
        for _ in range(5): torch.matmul(a, w.t(), out=out)
        e_.record(); torch.cuda.synchronize()
        tl.append(s_.elapsed_time(e_) / 5)
    res["_torch"] = tl
    print(name, json.dumps({f"impl{i}_tflops".replace("impl_torch", "torch"): round(2.0 * M * N * K / sorted(v)[1] / 1e9, 1) for i, v in res.items()}))
