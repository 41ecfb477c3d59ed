"""Step-phase timing probe of vae_conv3 (build with FVK_EXTRA_FLAGS=-DFVK_C3_PROBE): s_memtime sums of workgroup 0."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fastvideo_amd import ops, _lib
Cin = Cout = int(sys.argv[1]) if len(sys.argv) > 1 else 96
T, H, W = 4, 480, 832
if Cin == 192: H, W = 240, 416
x = torch.randn((T + 2, H, W, Cin), device="cuda").bfloat16()
w = (torch.randn((Cout, 27 * Cin), device="cuda") * (27 * Cin)**-0.5).bfloat16()
b = torch.zeros(Cout, device="cuda").bfloat16()
out = torch.empty((T, H, W, Cout), device="cuda", dtype=torch.bfloat16)
buf = torch.zeros(8 * 4, dtype=torch.int64, device="cuda")
p = lambda t: C.c_void_p(t.data_ptr())
for _ in range(2):
    _lib.call("fvk_vae_conv_bf16", p(x), p(w), p(b), p(out), None, p(buf), T, H, W, Cin, Cout, 3, 3, 3, T + 2, 0, H * W * Cout, 0, 0, 0, 0, ops._stream())
torch.cuda.synchronize()
steps = 3 * (Cin // 32) * 3 - 1
t = buf.cpu().view(8, 4).double() / steps
for wv in range(8):
    print(f"wave {wv}: loopback={t[wv,0]:.0f} stream(36 MFMA)={t[wv,1]:.0f} wait_dma={t[wv,2]:.0f} barrier={t[wv,3]:.0f} | step {t[wv].sum():.0f} cycles")
