"""The 96 -> 96 channel 3x3x3 conv of the VAE's full-resolution stage alone (for PMC passes / timing): T output frames of 480 x 832."""
import os as _os
_os.environ.setdefault("FVK_PROBE_LIB", "1")  # A/B switches exist only in the measurement build (scripts/probes/libfvk_probe.so)
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fastvideo_amd import ops
C, T, H, W = int(os.environ.get("C", "96")), 8, 480, 832
if C == 192: H, W = 240, 416
x = torch.randn((T + 2, H, W, C), device="cuda").bfloat16()
w = (torch.randn((C, 27 * C), device="cuda") * (27 * C)**-0.5).bfloat16()
b = torch.zeros(C, device="cuda").bfloat16()
ops.set_tunable("vae_conv_impl", int(os.environ.get("IMPL", "0")))
n = int(os.environ.get("N_LAUNCH", "3"))
for _ in range(2): ops.vae_conv(x, w, b, T=T, H=H, W=W, kt=3, ks=3)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(n): y = ops.vae_conv(x, w, b, T=T, H=H, W=W, kt=3, ks=3)
e.record(); torch.cuda.synchronize()
ms = s.elapsed_time(e) / n
print(f"C={C} impl={os.environ.get('IMPL', '0')} {ms:.3f} ms/launch {2.0 * T * H * W * C * 27 * C / ms / 1e9:.1f} TFLOP/s")
