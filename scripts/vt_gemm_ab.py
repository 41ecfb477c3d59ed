"""Round 5 A/B: the V projection as its own GEMM writing V^T (ops.gemm_vt, shipped) against the fused QKV GEMM + the V^T layout pass —
kernel level (per layer, cfg2 shapes) and the whole 30-layer contract forward, interleaved in one process.  usage: python scripts/vt_gemm_ab.py"""
import json, os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fastvideo_amd import ops, wan_config as WC
from fastvideo_amd.wan_dit import WanTransformer3DModelHip
dev = torch.device("cuda")
S, d = 32760, 1536
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn((S, d), generator=g, device=dev).bfloat16()
w = (torch.randn((3 * d, d), generator=g, device=dev) * d**-0.5).bfloat16()
b = torch.randn((3 * d,), generator=g, device=dev).bfloat16()


def timed(fn, n=20):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


def fused():
    qkv = ops.gemm(x, w, b)
    return ops.v_transpose(qkv[:, 2 * d:].view(1, S, 12, 128))


def split():
    qk = ops.gemm(x, w[:2 * d], b[:2 * d])
    return ops.gemm_vt(x.view(1, S, d), w[2 * d:], b[2 * d:])


assert torch.equal(fused(), split())
res = {"kernel_level_us": {}}
for rep in range(3):
    for name, fn in (("fused_qkv_gemm_plus_v_transpose", fused), ("qk_gemm_plus_vt_gemm", split),
                     ("qkv_gemm_alone", lambda: ops.gemm(x, w, b)), ("qk_gemm_alone", lambda: ops.gemm(x, w[:2 * d], b[:2 * d])),
                     ("vt_gemm_alone", lambda: ops.gemm_vt(x.view(1, S, d), w[2 * d:], b[2 * d:])),
                     ("v_transpose_alone", lambda: ops.v_transpose(x.view(1, S, 12, 128)))):
        res["kernel_level_us"].setdefault(name, []).append(round(timed(fn), 1))
print(json.dumps(res), flush=True)

cfg = WC.WAN21_T2V_1_3B
sd = WC.random_state_dict(cfg, seed=0, device=dev)
model = WanTransformer3DModelHip(sd, cfg.num_heads, cfg.head_dim, cfg.patch_size, cfg.eps, cfg.freq_dim, device=dev)
del sd
lat = torch.randn(WC.LATENT_81F_480P, generator=g, device=dev).bfloat16()
ctx = torch.randn((1, 512, cfg.text_dim), generator=g, device=dev).bfloat16()
ts = torch.tensor([500.0], device=dev)
outs = {}
for flag in (True, False):
    model.vt_gemm = flag
    for _ in range(3): outs[flag] = model(lat, ctx, ts)
torch.cuda.synchronize()
ms = {True: [], False: []}
for rep in range(6):
    for flag in (True, False):
        model.vt_gemm = flag
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(4): y = model(lat, ctx, ts)
        e.record(); torch.cuda.synchronize()
        ms[flag].append(round(s.elapsed_time(e) / 4, 3))
res["forward_ms"] = {"vt_gemm": ms[True], "fused_qkv_plus_v_transpose": ms[False], "median_vt_gemm": statistics.median(ms[True]),
                     "median_fused": statistics.median(ms[False]), "outputs_bit_identical": bool(torch.equal(outs[True], outs[False]))}
print(json.dumps(res), flush=True)
