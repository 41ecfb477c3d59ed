"""Per-rank GEMM shapes of sequence parallelism (M = 32760 / P rows): gemm_w1 (256x256 tiles, gemm_impl 14 = gemm_w1n forbidden), gemm_w1n (256x128
tiles, gemm_impl 6 = forced; round 4) and the 128x128 kernel (1) — few-tile launches leave most of the 256 CUs idle on 256x256 tiles.  Also the plain
product dispatch (0).  Measurement build (FVK_PROBE_LIB=1)."""
import os as _os
_os.environ.setdefault("FVK_PROBE_LIB", "1")
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fastvideo_amd import ops
d, F = 1536, 8960
for P in (8, 4, 2, 1):
    M = -(-32760 // P)
    for name, N, K in (("qkv", 3 * d, d), ("out", d, d), ("ffn_in", F, d), ("ffn_out", d, F)):
        a = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") * K**-0.5).bfloat16(); b = torch.randn(N, device="cuda").bfloat16()
        out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        res = {}
        for r in range(3):
            for i in (14, 6, 1, 0):
                ops.set_tunable("gemm_impl", i)
                ops.gemm(a, w, b, out=out); torch.cuda.synchronize()
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for _ in range(10): ops.gemm(a, w, b, out=out)
                e.record(); torch.cuda.synchronize()
                res.setdefault(i, []).append(s.elapsed_time(e) / 10)
        ops.set_tunable("gemm_impl", 0)
        t256 = -(-M // 256) * -(-N // 256)
        names = {14: "gemm_w1", 6: "gemm_w1n", 1: "128x128", 0: "dispatch"}
        print(f"P={P} {name} M={M} N={N} K={K} tiles256={t256} tiles128={-(-M // 256) * -(-N // 128)}", json.dumps({f"{names[i]}_us": round(sorted(v)[1] * 1e3, 1) for i, v in res.items()}))
