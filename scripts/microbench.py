"""Per-kernel timings at the Wan2.1-1.3B 81f x 480p (cfg2) shapes.  Not the contract bench (see bench.py)."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fastvideo_amd import ops

dev = "cuda"
def t_ms(fn, it=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it

S, d, H, D, F = 32760, 1536, 12, 128, 8960
res = {}
x = torch.randn(S, d, device=dev).bfloat16()
def gemm_case(name, M, N, K, epi=0):
    a = torch.randn(M, K, device=dev).bfloat16(); w = (torch.randn(N, K, device=dev) * K**-0.5).bfloat16(); b = torch.randn(N, device=dev).bfloat16()
    ms = t_ms(lambda: ops.gemm(a, w, b, epilogue=epi))
    res[name] = dict(ms=ms, tflops=2.0 * M * N * K / ms / 1e9)
gemm_case("gemm_qkv_fused[S,4608,1536]", S, 3 * d, d)
gemm_case("gemm_out[S,1536,1536]", S, d, d)
gemm_case("gemm_ffn_in_gelu[S,8960,1536]", S, F, d, ops.EPI_GELU_TANH)
gemm_case("gemm_ffn_out[S,1536,8960]", S, d, F)
gemm_case("gemm_8k^3", 8192, 8192, 8192)
# torch (hipBLASLt) reference points for the same shapes
for name, (M, N, K) in {"torch_mm_qkv": (S, 3 * d, d), "torch_mm_ffn_in": (S, F, d), "torch_mm_8k^3": (8192, 8192, 8192)}.items():
    a = torch.randn(M, K, device=dev).bfloat16(); w = torch.randn(N, K, device=dev).bfloat16()
    ms = t_ms(lambda: torch.nn.functional.linear(a, w)); res[name] = dict(ms=ms, tflops=2.0 * M * N * K / ms / 1e9)

q = torch.randn(1, S, H, D, device=dev).bfloat16(); k = torch.randn(1, S, H, D, device=dev).bfloat16(); v = torch.randn(1, S, H, D, device=dev).bfloat16()
vt = ops.v_transpose(v)
ms = t_ms(lambda: ops.attn_dense(q, k, vt=vt), it=5, warm=2)
res["attn_dense[S=32760,H=12]"] = dict(ms=ms, tflops=4.0 * S * S * H * D / ms / 1e9)
ms = t_ms(lambda: ops.v_transpose(v)); res["v_transpose"] = dict(ms=ms, gbs=2 * S * d * 2 / ms / 1e6)
try:
    ms = t_ms(lambda: torch.nn.functional.scaled_dot_product_attention(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2)), it=3, warm=1)
    res["torch_sdpa"] = dict(ms=ms, tflops=4.0 * S * S * H * D / ms / 1e9)
except Exception as ex:
    res["torch_sdpa"] = dict(error=str(ex)[:200])
mul = torch.randn(1, d, device=dev); add = torch.randn(1, d, device=dev)
ms = t_ms(lambda: ops.ln_modulate(x, mul=mul, add=add)); res["ln_modulate"] = dict(ms=ms, gbs=2 * S * d * 2 / ms / 1e6)
ms = t_ms(lambda: ops.ln_modulate(x, residual=x, gate=mul, ln_w=mul[0], ln_b=add[0], want_residual=True)); res["res_gate_ln"] = dict(ms=ms, gbs=4 * S * d * 2 / ms / 1e6)
qkv = torch.randn(S, 3 * d, device=dev).bfloat16(); w = torch.ones(d, device=dev).bfloat16()
cos = torch.randn(S, D, device=dev); sin = torch.randn(S, D, device=dev)
ms = t_ms(lambda: ops.rmsnorm_rope([qkv[:, :d], qkv[:, d:2 * d]], [w, w], cos, sin, seq_len=S)); res["qk_rmsnorm_rope"] = dict(ms=ms, gbs=(4 * S * d * 2 + 2 * S * D * 8) / ms / 1e6)
for k_, v_ in res.items(): print(k_, json.dumps(v_))
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/microbench.json", "w"), indent=1)
