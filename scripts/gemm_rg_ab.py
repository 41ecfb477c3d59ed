"""The gated-residual epilogue (FFN-out / out-projection form) at the cfg2 shapes: shipped gemm_w1 (gemm_impl 0) vs its LDS-bounce variant (61) vs gemm_ph (228); plain epilogue beside it."""
import os as _os
_os.environ.setdefault("FVK_PROBE_LIB", "1")
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fastvideo_amd import ops
S, d, F = 32760, 1536, 8960
for name, M, N, K in (("ffn_out", S, d, F), ("out", S, d, d)):
    x = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") * K**-0.5).bfloat16(); b = torch.randn(N, device="cuda").bfloat16()
    res = torch.randn(M, N, device="cuda").bfloat16(); gate = torch.randn(1, N, device="cuda")
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    r = {}
    for rep in range(3):
        for impl in (0, 61, 228):
            for epi, kw in (("rg", dict(epilogue=ops.EPI_RESIDUAL_GATE, residual=res, gate=gate)), ("plain", {})):
                ops.set_tunable("gemm_impl", impl)
                ops.gemm(x, w, b, out=out, **kw); torch.cuda.synchronize()
                s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s_.record()
                for _ in range(5): ops.gemm(x, w, b, out=out, **kw)
                e_.record(); torch.cuda.synchronize()
                r.setdefault(f"impl{impl}_{epi}", []).append(s_.elapsed_time(e_) / 5)
    ops.set_tunable("gemm_impl", 0)
    print(name, json.dumps({k: round(2.0 * M * N * K / sorted(v)[1] / 1e9, 1) for k, v in r.items()}))
