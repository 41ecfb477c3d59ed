"""Power / clock of the 96 -> 96 (and 192 -> 192) 3x3x3 conv under each kernel variant: ~2 s of back-to-back launches per variant with the amdsmi
sampler running (scripts/power_trace.py).  Measurement build.  vae_conv_impl: 0 persistent conv3w, 4 conv3w one workgroup per tile, 5 persistent +
phase stagger, 3 the 8-wave vae_conv3 kernel."""
import os as _os
_os.environ.setdefault("FVK_PROBE_LIB", "1")
import importlib.util, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fastvideo_amd import ops
spec = importlib.util.spec_from_file_location("power_trace", os.path.join(os.path.dirname(os.path.abspath(__file__)), "power_trace.py"))
pt = importlib.util.module_from_spec(spec); spec.loader.exec_module(pt)
T = 16
for C, H, W in ((96, 480, 832), (192, 240, 416), (384, 120, 208)):
    x = torch.randn((T + 2, H, W, C), device="cuda").bfloat16()
    w = (torch.randn((C, 27 * C), device="cuda") * (27 * C)**-0.5).bfloat16()
    b = torch.zeros(C, device="cuda").bfloat16()
    out = torch.empty((T, H, W, C), dtype=torch.bfloat16, device="cuda")
    fl = 2.0 * T * H * W * C * 27 * C
    for impl in (0, 5, 4, 3, 0):
        ops.set_tunable("vae_conv_impl", impl)
        for _ in range(3): ops.vae_conv(x, w, b, T=T, H=H, W=W, kt=3, ks=3, out=out)
        torch.cuda.synchronize()
        ps = pt.PowerSampler(20.0).start()
        t0 = time.time(); n = 0
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        while time.time() - t0 < 2.0:
            for _ in range(20): ops.vae_conv(x, w, b, T=T, H=H, W=W, kt=3, ks=3, out=out)
            n += 20
            torch.cuda.synchronize()
        e.record(); torch.cuda.synchronize()
        t1 = time.time()
        sm = pt.summarize(ps.stop(), t0 + 0.5, t1, "amdsmi")
        ms = s.elapsed_time(e) / n
        print(f"C={C} impl={impl}: {ms:.3f} ms/launch {fl / ms / 1e9:7.1f} TF | power {sm['power_w']['mean'] if sm['power_w'] else None} W  sclk {sm['sclk_mhz']['mean'] if sm['sclk_mhz'] else None} MHz "
              f"(min over XCDs {sm['sclk_min_over_xcds_mhz']['mean'] if sm['sclk_min_over_xcds_mhz'] else None})  -> {fl / ms / 1e9 / (2500 * (sm['sclk_mhz']['mean'] if sm['sclk_mhz'] else 2400) / 2400):.3f} of the peak at that clock", flush=True)
ops.set_tunable("vae_conv_impl", 0)
