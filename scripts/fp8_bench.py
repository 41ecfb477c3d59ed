"""fp8 linear path timings at the cfg2 shapes (measurement build: gemm_impl 2 = gemm_pp's fp8 kernel, 0 = gemm_w1's): python scripts/fp8_bench.py"""
import os as _os
_os.environ.setdefault("FVK_PROBE_LIB", "1")
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fastvideo_amd import ops
def t_ms(fn, it=10):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it
S, d, F = 32760, 1536, 8960
for name, M, N, K in (("qkv", S, 3 * d, d), ("out", S, d, d), ("ffn_in", S, F, d), ("ffn_out", S, d, F), ("8k", 8192, 8192, 8192)):
    x = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") * K**-0.5).bfloat16()
    b = torch.randn(N, device="cuda").bfloat16()
    wq, ws = ops.fp8_quantize(w)
    xq, xs = ops.fp8_quantize(x)
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    tq = t_ms(lambda: ops.fp8_quantize(x))
    ops.set_tunable("gemm_impl", 2); tp = t_ms(lambda: ops.gemm_fp8(xq, xs, wq, ws, b, out=out)); o_pp = out.clone()
    ops.set_tunable("gemm_impl", 0); tg = t_ms(lambda: ops.gemm_fp8(xq, xs, wq, ws, b, out=out))
    dmax = (out.float() - o_pp.float()).abs().max().item()
    tb = t_ms(lambda: ops.gemm(x, w, b, out=out))
    fl = 2.0 * M * N * K
    print(name, json.dumps({"quantize_ms": round(tq, 4), "quantize_GBps": round(M * K * 5 / tq / 1e6, 0), "gemm_fp8_ms": round(tg, 4),
                            "gemm_fp8_tflops": round(fl / tg / 1e9, 1), "gemm_pp_fp8_tflops": round(fl / tp / 1e9, 1), "max|w1-pp|": round(dmax, 4), "fp8_total_tflops": round(fl / (tg + tq) / 1e9, 1),
                            "gemm_bf16_ms": round(tb, 4), "gemm_bf16_tflops": round(fl / tb / 1e9, 1)}))
