"""CPU restatement of the reference's fp8 (e4m3fn) linear path.  TEST INFRASTRUCTURE ONLY (same import rule as wan_oracle.py).

Restates fastvideo/layers/quantization/fp8_config.py:
  :24-26   FP8_DTYPE = float8_e4m3fn, FP8_MAX = 448, FP8_MIN_SCALE = 1/(448*512)
  :55-68   _quantize_tensorwise / _quantize_rowwise   (absmax -> scale -> divide IN THE INPUT DTYPE -> clamp -> cast)
  :119-158 FP8QuantizeMethod.apply                     (torch._scaled_mm(x_fp8, w_fp8.t(), scale_a, scale_b, out_dtype=bf16) + bias)
  :211-245 convert_model_to_fp8                        (the weight side of the same arithmetic; per-tensor or per-output-channel)
torch._scaled_mm has no CPU kernel for every layout, so the product is restated as its definition: fp32 accumulation of the exact
e4m3 x e4m3 products, times the two scales, rounded once to bf16.
Pinned: tests/test_fp8_oracle.py compares the quantisers bit-for-bit with the reference's own functions imported from /root/reference
(when present) and with tests/golden/fp8_quant.pt."""
from __future__ import annotations

import torch

FP8_DTYPE = torch.float8_e4m3fn
FP8_MAX = float(torch.finfo(FP8_DTYPE).max)
FP8_MIN_SCALE = 1.0 / (FP8_MAX * 512.0)


def quantize_tensorwise(x_2d):
    x_absmax = x_2d.abs().amax().float()
    x_scale = (x_absmax / FP8_MAX).clamp(min=FP8_MIN_SCALE)
    x_fp8 = (x_2d / x_scale.to(x_2d.dtype)).clamp(-FP8_MAX, FP8_MAX).to(FP8_DTYPE)
    return x_fp8, x_scale.view(1)


def quantize_rowwise(x_2d):
    x_absmax = x_2d.abs().amax(dim=-1, keepdim=True).float()
    x_scale = (x_absmax / FP8_MAX).clamp(min=FP8_MIN_SCALE)
    x_fp8 = (x_2d / x_scale.to(x_2d.dtype)).clamp(-FP8_MAX, FP8_MAX).to(FP8_DTYPE)
    return x_fp8, x_scale


def quantize_weight(w, granularity="tensor"):
    """convert_model_to_fp8 (fp8_config.py:225-236): returns (w_fp8 [N,K], w_scale fp32 [1] or [N])."""
    if granularity == "channel":
        w_absmax = w.detach().abs().amax(dim=1).nan_to_num().float()
        w_scale = (w_absmax / FP8_MAX).clamp(min=FP8_MIN_SCALE)
        w_fp8 = (w / w_scale.to(w.dtype).unsqueeze(1)).clamp(-FP8_MAX, FP8_MAX).to(FP8_DTYPE)
        return w_fp8, w_scale
    w_absmax = w.detach().abs().amax().nan_to_num().to(torch.float32)
    w_scale = (w_absmax / FP8_MAX).clamp(min=FP8_MIN_SCALE).view(1)
    w_fp8 = (w / w_scale.to(w.dtype)).clamp(-FP8_MAX, FP8_MAX).to(FP8_DTYPE)
    return w_fp8, w_scale


def scaled_mm_bias(x_fp8, x_scale, w_fp8, w_scale, bias=None):
    """FP8QuantizeMethod.apply (fp8_config.py:141-152): bf16(_scaled_mm) then `out + bias` in bf16."""
    acc = x_fp8.float() @ w_fp8.float().t()
    sb = w_scale.view(1, -1) if w_scale.numel() != 1 else w_scale
    out = (acc * x_scale.float() * sb.float()).to(torch.bfloat16)
    if bias is not None:
        out = out + bias
    return out


def fp8_linear(x, w_fp8, w_scale, bias=None, granularity="tensor"):
    x2 = x.reshape(-1, x.shape[-1])
    xq, xs = quantize_rowwise(x2) if granularity == "channel" else quantize_tensorwise(x2)
    return scaled_mm_bias(xq, xs, w_fp8, w_scale, bias).view(*x.shape[:-1], w_fp8.shape[0])
