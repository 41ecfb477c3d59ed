"""Stream-boundary timing probe of gemm_st (build with FVK_EXTRA_FLAGS=-DFVK_ST_PROBE): s_memtime sums of workgroup 0."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fastvideo_amd import ops, _lib
M, N, K = 32760, 1536, 8960
a = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") * K**-0.5).bfloat16()
out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
buf = torch.zeros(8 * 4, dtype=torch.int64, device="cuda")
for _ in range(2):
    _lib.call("fvk_gemm_bf16", ops._p(a), ops._p(w), None, ops._p(out), M, N, K, K, N, 0, None, C.c_void_p(buf.data_ptr()), M, ops._stream())
torch.cuda.synchronize()
steps = K // 32 - 9
t = buf.cpu().view(8, 4).double() / steps
for wv in range(8):
    print(f"wave {wv}: loopback={t[wv,0]:.0f} stream(16 MFMA)={t[wv,1]:.0f} wait_dma={t[wv,2]:.0f} barrier={t[wv,3]:.0f} | step {t[wv].sum():.0f} cycles")
