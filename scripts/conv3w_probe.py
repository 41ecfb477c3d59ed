"""Where a tile of the one-wave-per-SIMD conv kernel (vae_conv3w.hip) spends its cycles: s_memtime stamps of every workgroup's wave 0 at kernel
entry / loop start / loop end / kernel end (measurement build only: FVK_PROBE_LIB=1), for the 96 -> 96 full-resolution conv and the 192 -> 192
half-resolution one, kt = 3 and kt = 1 (27 / 9 and 54 / 18 K-steps per tile).  usage: FVK_PROBE_LIB=1 python scripts/conv3w_probe.py"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from fastvideo_amd import _lib, ops

assert os.environ.get("FVK_PROBE_LIB") == "1"
g = torch.Generator().manual_seed(0)
IMPL = int(os.environ.get("VAE_CONV_IMPL", "0"))   # 4 = one workgroup per tile (every tile pays its cold prologue), 0 = persistent workgroups
ops.set_tunable("vae_conv_impl", IMPL)
print(f"vae_conv_impl {IMPL}")
for (Cin, Cout, T, H, W, kt) in ((96, 96, 4, 480, 832, 3), (96, 96, 4, 480, 832, 1), (192, 192, 4, 240, 416, 3), (192, 192, 4, 240, 416, 1),
                                 (384, 384, 4, 120, 208, 3)):
    ring = T + kt - 1
    x = torch.randn((ring, H, W, Cin), generator=g).cuda().bfloat16()
    w = (torch.randn((Cout, kt * 9 * Cin), generator=g) * (kt * 9 * Cin)**-0.5).cuda().bfloat16()
    b = torch.zeros((Cout,)).cuda().bfloat16()
    out = torch.empty((T, H, W, Cout), dtype=torch.bfloat16, device="cuda")
    TH, TN = (16, 96) if Cout % 192 else (8, 192)
    nwg = T * ((H + TH - 1) // TH) * ((W + 31) // 32) * (Cout // TN)
    probe = torch.zeros((nwg, 4), dtype=torch.int64, device="cuda")   # persistent launches fill the first min(nwg, CUs) rows: each workgroup's FIRST tile
    p = lambda t: None if t is None else C.c_void_p(t.data_ptr())
    for it in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _lib.call("fvk_vae_conv_bf16", p(x), p(w), p(b), p(out), None, p(probe), T, H, W, Cin, Cout, kt, 3, 3, ring, 0, H * W * Cout, 0, 0, 0, 0, ops._stream())
        e1.record()
        torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    st = probe.cpu().double()
    st = st[st[:, 3] > 0]
    pro, loop, epi = (st[:, 1] - st[:, 0]), (st[:, 2] - st[:, 1]), (st[:, 3] - st[:, 2])
    nstep = kt * (Cin // 32) * 3
    span = (st[:, 3].max() - st[:, 0].min()).item()
    fl = 2.0 * T * H * W * Cout * kt * 9 * Cin
    print(f"{Cin}->{Cout} kt={kt} {H}x{W} T={T}: {nwg} workgroups, {nstep} steps/tile, {ms:.3f} ms = {fl / ms / 1e9:.0f} TF | per tile (mean cycles of the "
          f"100-MHz-class s_memtime counter x its ratio is unknown: use RATIOS) prologue {pro.mean():.0f} loop {loop.mean():.0f} ({loop.mean() / nstep:.1f}/step) "
          f"epilogue {epi.mean():.0f} | share prologue {pro.mean() / (pro + loop + epi).mean():.3f} epilogue {epi.mean() / (pro + loop + epi).mean():.3f} | "
          f"kernel span {span:.0f} ticks, sum of tile times / 256 CUs = {(pro + loop + epi).sum().item() / 256:.0f} ticks (ratio {(pro + loop + epi).sum().item() / 256 / span:.3f})")
