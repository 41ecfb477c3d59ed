"""attn_w16's row sums on the matrix pipe (a ninth d block of ones: 1/17 of the MFMA work; "attn_impl" 300) against fp32 VALU
adds of the unrounded probabilities in the softmax chunks ("attn_impl" 320), at the contract shape (32 760 x 32 760, 12 heads), back to back and
interleaved; outputs and LSE against exact fp32 attention on sampled rows."""
import os as _os
_os.environ.setdefault("FVK_PROBE_LIB", "1")
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fastvideo_amd import ops
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
S, H, D = 32760, 12, 128
q, k, v = (torch.randn((1, S, H, D), generator=g, device=dev).bfloat16() for _ in range(3))
V = {"row sums on the matrix pipe (shipped)": 300, "row sums as fp32 VALU adds": 320}
t, outs = {n: [] for n in V}, {}
for r in range(5):
    for name, impl in V.items():
        ops.set_tunable("attn_impl", impl)
        o = ops.attn_dense(q, k, v, scale=D**-0.5, layout="bshd", return_lse=True); torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20): ops.attn_dense(q, k, v, scale=D**-0.5, layout="bshd")
        e.record(); torch.cuda.synchronize()
        t[name].append(round(s.elapsed_time(e) / 20, 4))
        outs[name] = o
ops.set_tunable("attn_impl", 0)
rows = torch.tensor([0, 1, 255, 256, 4095, 16383, 32759, 20000, 777], device=dev)
err, lerr = {}, {}
for name, (o, lse) in outs.items():
    e_, l_ = 0.0, 0.0
    for h in (0, 5, 11):
        s_ = (q[0, rows, h].float() @ k[0, :, h].float().T) * D**-0.5
        ref = torch.softmax(s_, -1) @ v[0, :, h].float()
        e_ = max(e_, (o[0, rows, h].float() - ref).abs().max().item())
        l_ = max(l_, (lse[0, h, rows] - torch.logsumexp(s_, -1) * 1.4426950408889634).abs().max().item())
    err[name], lerr[name] = round(e_, 6), round(l_, 6)
a, b = outs["row sums on the matrix pipe (shipped)"][0].float(), outs["row sums as fp32 VALU adds"][0].float()
fl = 4.0 * S * S * H * D
print(json.dumps({"ms": t, "tflops_best": {n: round(fl / (min(v_) * 1e-3) / 1e12, 1) for n, v_ in t.items()},
                  "max_abs_err_vs_exact_fp32_sampled_rows": err, "max_abs_lse_err_log2_units": lerr,
                  "between_variants_max_abs": round((a - b).abs().max().item(), 6), "between_variants_mean_abs": float(f"{(a - b).abs().mean().item():.3g}")}, indent=1))
