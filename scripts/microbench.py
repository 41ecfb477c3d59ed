"""Per-kernel timings at the Wan2.1-1.3B 81f x 480p (cfg2) shapes, with within-process interleaved A/B of the kernel
variants behind fvk_set_tunable (guide §5.4 rule 24).  Not the contract bench (see bench.py).
usage: python scripts/microbench.py [--quick]"""
import os as _os
_os.environ.setdefault("FVK_PROBE_LIB", "1")  # A/B switches exist only in the measurement build (scripts/probes/libfvk_probe.so)
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fastvideo_amd import ops

dev = "cuda"
QUICK = "--quick" in sys.argv


def t_ms(fn, it=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / it


def ab(fn, tunable, variants, rounds=3, it=5):
    """Interleaved rounds over the variants; returns {variant: (median_ms, min_ms)}."""
    times = {v: [] for v in variants}
    for _ in range(rounds):
        for v in variants:
            ops.set_tunable(tunable, v)
            times[v].append(t_ms(fn, it=it, warm=1))
    ops.set_tunable(tunable, 0)
    return {v: (sorted(ts)[len(ts) // 2], min(ts)) for v, ts in times.items()}


S, d, H, D, F = 32760, 1536, 12, 128, 8960
res = {}


def gemm_case(name, M, N, K, epi=0, residual=False):
    a = torch.randn(M, K, device=dev).bfloat16()
    w = (torch.randn(N, K, device=dev) * K**-0.5).bfloat16()
    b = torch.randn(N, device=dev).bfloat16()
    r = torch.randn(M, N, device=dev).bfloat16() if residual else None
    g = torch.randn(1, N, device=dev) if residual else None
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    fl = 2.0 * M * N * K
    r_ = ab(lambda: ops.gemm(a, w, b, epilogue=epi, residual=r, gate=g, out=out), "gemm_impl", [0, 1])
    ms = t_ms(lambda: torch.nn.functional.linear(a, w, b), it=5)
    res[name] = {"pp_ms": r_[0][0], "pp_tflops": fl / r_[0][0] / 1e9, "pp_best_tflops": fl / r_[0][1] / 1e9,
                 "v1_tflops": fl / r_[1][0] / 1e9, "hipblaslt_ms": ms, "hipblaslt_tflops": fl / ms / 1e9}
    print(name, json.dumps(res[name]), flush=True)


gemm_case("gemm_qkv_fused[S,4608,1536]", S, 3 * d, d)
gemm_case("gemm_out[S,1536,1536]", S, d, d)
gemm_case("gemm_ffn_in_gelu[S,8960,1536]", S, F, d, ops.EPI_GELU_TANH)
gemm_case("gemm_ffn_out_resgate[S,1536,8960]", S, d, F, ops.EPI_RESIDUAL_GATE, residual=True)
if not QUICK:
    gemm_case("gemm_ffn_in_noepi[S,8960,1536]", S, F, d)
    gemm_case("gemm_8k^3", 8192, 8192, 8192)
    gemm_case("gemm_4k^3", 4096, 4096, 4096)

q, k, v = (torch.randn(1, S, H, D, device=dev).bfloat16() for _ in range(3))
vt = ops.v_transpose(v)
o = torch.empty_like(q)
fl = 4.0 * S * S * H * D
r_ = ab(lambda: ops.attn_dense(q, k, vt=vt, out=o), "attn_impl", [0, 1, 2], rounds=3, it=3)
res["attn_dense[S=32760,H=12]"] = {f"impl{v_}_tflops": fl / m[0] / 1e9 for v_, m in r_.items()} | {f"impl{v_}_ms": m[0] for v_, m in r_.items()}
print("attn_dense", json.dumps(res["attn_dense[S=32760,H=12]"]), flush=True)
# cross-attention shape: 512 text keys
kc, vc = (torch.randn(1, 512, H, D, device=dev).bfloat16() for _ in range(2))
vtc = ops.v_transpose(vc)
flc = 4.0 * S * 512 * H * D
r_ = ab(lambda: ops.attn_dense(q, kc, vt=vtc, out=o), "attn_impl", [0, 1], rounds=3, it=5)
res["attn_cross[S x 512]"] = {f"impl{v_}_tflops": flc / m[0] / 1e9 for v_, m in r_.items()} | {f"impl{v_}_ms": m[0] for v_, m in r_.items()}
print("attn_cross", json.dumps(res["attn_cross[S x 512]"]), flush=True)
if not QUICK:
    try:
        ms = t_ms(lambda: torch.nn.functional.scaled_dot_product_attention(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2)), it=2, warm=1)
        res["torch_sdpa"] = dict(ms=ms, tflops=fl / ms / 1e9)
    except Exception as ex:
        res["torch_sdpa"] = dict(error=str(ex)[:200])
    print("torch_sdpa", json.dumps(res["torch_sdpa"]), flush=True)

x = torch.randn(S, d, device=dev).bfloat16()
ms = t_ms(lambda: ops.v_transpose(v)); res["v_transpose"] = dict(ms=ms, gbs=2 * S * d * 2 / ms / 1e6)
mul = torch.randn(1, d, device=dev); add = torch.randn(1, d, device=dev)
ms = t_ms(lambda: ops.ln_modulate(x, mul=mul, add=add)); res["ln_modulate"] = dict(ms=ms, gbs=2 * S * d * 2 / ms / 1e6)
ms = t_ms(lambda: ops.ln_modulate(x, residual=x, gate=mul, ln_w=mul[0], ln_b=add[0], want_residual=True)); res["res_gate_ln"] = dict(ms=ms, gbs=4 * S * d * 2 / ms / 1e6)
qkv = torch.randn(S, 3 * d, device=dev).bfloat16(); w1 = torch.ones(d, device=dev).bfloat16()
cos = torch.randn(S, D, device=dev); sin = torch.randn(S, D, device=dev)
ms = t_ms(lambda: ops.rmsnorm_rope([qkv[:, :d], qkv[:, d:2 * d]], [w1, w1], cos, sin, seq_len=S)); res["qk_rmsnorm_rope"] = dict(ms=ms, gbs=(4 * S * d * 2 + 2 * S * D * 8) / ms / 1e6)
for k_ in ("v_transpose", "ln_modulate", "res_gate_ln", "qk_rmsnorm_rope"):
    print(k_, json.dumps(res[k_]))
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/microbench.json", "w"), indent=1)
