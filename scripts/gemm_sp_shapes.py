"""GEMM kernels at the PER-RANK shapes of sequence-parallel runs (cfg2: M = 32760 / P tokens): 256x256 LDS-DMA kernel (impl 0) vs the
128x128 kernel (impl 1).  Wave quantisation is what matters here: M = 4095, N = 1536 is 96 tiles of 256x256 on 256 CUs.
usage: python scripts/gemm_sp_shapes.py [impl ...]"""
import os as _os
_os.environ.setdefault("FVK_PROBE_LIB", "1")  # A/B switches exist only in the measurement build (scripts/probes/libfvk_probe.so)
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fastvideo_amd import ops

impls = [int(x) for x in sys.argv[1:]] or [0, 1]
d, F = 1536, 8960
for P in (2, 4, 8):
    M = (32760 + P - 1) // P
    for name, N, K in (("qkv", 3 * d, d), ("o", d, d), ("ffn_in", F, d), ("ffn_out", d, F)):
        a = torch.randn(M, K, device="cuda").bfloat16()
        w = (torch.randn(N, K, device="cuda") * K**-0.5).bfloat16()
        out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        res = {i: [] for i in impls}
        for r in range(3):
            for i in impls:
                ops.set_tunable("gemm_impl", i)
                ops.gemm(a, w, None, out=out)
                torch.cuda.synchronize()
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for _ in range(10):
                    ops.gemm(a, w, None, out=out)
                e.record()
                torch.cuda.synchronize()
                res[i].append(s.elapsed_time(e) / 10)
        ops.set_tunable("gemm_impl", 0)
        t256 = -(-M // 256) * -(-N // 256)
        print(f"P={P} M={M} {name:8s} N={N} K={K} tiles256={t256}", json.dumps({f"impl{i}": {"us": round(sorted(v)[1] * 1e3, 1), "tflops": round(2.0 * M * N * K / sorted(v)[1] / 1e9, 1)} for i, v in res.items()}))
