"""Layer-op bindings: the reference's ``fastvideo.layers`` op API with MI355X HIP kernels behind ``forward_cuda`` / ``apply``.

The reference exposes two extension points for per-layer kernels (SURVEY.md §8b "Layer-op API"):

* ``CustomOp`` (fastvideo/layers/custom_op.py:14-97): an ``nn.Module`` whose ``forward`` dispatches to ``forward_cuda`` /
  ``forward_native``; ops are registered by name (``rms_norm`` layernorm.py:12, ``rotary_embedding`` rotary_embedding.py:153).
  The reference's ``dispatch_forward`` is hard-wired to ``forward_native`` (custom_op.py:54-58, a FIXME there) — the classes
  below override it, so constructing them IS selecting the HIP kernel.
* ``QuantizeMethodBase`` (fastvideo/layers/quantization/base_config.py:18-47; linear flavour ``LinearMethodBase``
  fastvideo/layers/linear.py:80-118): ``create_weights(layer, in, [outs], in, out, dtype, **attrs)`` + ``apply(layer, x, bias)``,
  selected per layer by ``QuantizationConfig.get_quant_method(layer, prefix)`` and registered by name with
  ``@register_quantization_config`` (fastvideo/layers/quantization/__init__.py:13-45).

When ``fastvideo`` is importable these classes subclass the reference's own base classes (so ``isinstance`` checks inside the
reference hold and ``ReplicatedLinear(quant_config=Mi355xBf16Config())`` just works); otherwise structural twins are used (the
GPU box has no reference package installed).  ``install()`` performs the registrations a maintainer would add.

No eager fallback: non-ROCm tensors, unsupported dtypes or modes raise (the reference's refusal convention,
fastvideo/platforms/cuda.py:149-154).  ``forward_native`` stays what the reference defines — it is the oracle the tests compare to.
"""
from __future__ import annotations

import torch
import torch.nn as nn
from torch.nn.parameter import Parameter

from . import ops

BF16 = torch.bfloat16
FP8_SUFFIXES = ("ffn.fc_in", "ffn.fc_out", "to_q", "to_k", "to_v", "to_out")  # ref: fp8_config.py:31-44 (the Wan names)

try:  # the reference's own bases when it is installed
    from fastvideo.layers.custom_op import CustomOp as _CustomOp  # type: ignore
    from fastvideo.layers.layernorm import RMSNorm as _RefRMSNorm  # type: ignore
    from fastvideo.layers.linear import LinearBase as _LinearBase  # type: ignore
    from fastvideo.layers.linear import LinearMethodBase as _LinearMethodBase  # type: ignore
    from fastvideo.layers.quantization.base_config import QuantizationConfig as _QuantizationConfig  # type: ignore
    from fastvideo.layers.quantization.base_config import QuantizeMethodBase as _QuantizeMethodBase  # type: ignore
    from fastvideo.models.utils import set_weight_attrs as _set_weight_attrs  # type: ignore
    HAVE_REFERENCE = True
except Exception:  # noqa: BLE001 - no reference package: structural twins with the same method names
    HAVE_REFERENCE = False

    class _CustomOp(nn.Module):  # ref: custom_op.py:14-97
        op_registry: dict = {}

        def __init__(self) -> None:
            super().__init__()
            self._forward_method = self.dispatch_forward()

        def forward(self, *args, **kwargs):
            return self._forward_method(*args, **kwargs)

        def forward_native(self, *args, **kwargs):
            raise NotImplementedError

        def forward_cuda(self, *args, **kwargs):
            raise NotImplementedError

        def dispatch_forward(self):
            return self.forward_native

        @classmethod
        def register(cls, name: str):

            def decorator(op_cls):
                op_cls.name = name
                cls.op_registry[name] = op_cls
                return op_cls

            return decorator

    class _RefRMSNorm(_CustomOp):  # ref: layernorm.py:12-83 (constructor + forward_native restated for the twin)

        def __init__(self, hidden_size: int, eps: float = 1e-6, dtype: torch.dtype = torch.float32,
                     var_hidden_size: int | None = None, has_weight: bool = True) -> None:
            super().__init__()
            self.hidden_size, self.variance_epsilon, self.has_weight = hidden_size, eps, has_weight
            self.variance_size_override = None if var_hidden_size == hidden_size else var_hidden_size
            self.weight = torch.ones(hidden_size)
            if has_weight:
                self.weight = nn.Parameter(self.weight)

        def forward_native(self, x, residual=None):
            orig = x.dtype
            x = x.to(torch.float32)
            if residual is not None:
                x = x + residual.to(torch.float32)
                residual = x.to(orig)
            x = (x * torch.rsqrt(x.pow(2).mean(dim=-1, keepdim=True) + self.variance_epsilon)).to(orig)
            if self.has_weight:
                x = x * self.weight
            return x if residual is None else (x, residual)

    class _QuantizeMethodBase:  # ref: base_config.py:18-47

        def create_weights(self, layer, *weight_args, **extra_weight_attrs):
            raise NotImplementedError

        def apply(self, layer, *args, **kwargs):
            raise NotImplementedError

        def process_weights_after_loading(self, layer) -> None:
            return

    _LinearMethodBase = _QuantizeMethodBase

    class _QuantizationConfig:  # ref: base_config.py:62-140

        def __init__(self) -> None:
            self.packed_modules_mapping: dict = {}

    _LinearBase = nn.Module

    def _set_weight_attrs(weight, attrs):  # ref: fastvideo/models/utils.py set_weight_attrs
        for k, v in (attrs or {}).items():
            setattr(weight, k, v)


def _require_rocm_bf16(t: torch.Tensor, what: str) -> None:
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError(f"fastvideo_amd.layers: {what} must be a ROCm device tensor (the HIP path has no CPU fallback)")
    if t.dtype != BF16:
        raise RuntimeError(f"fastvideo_amd.layers: {what} must be bf16 (the reference runs these ops under bf16 autocast), got {t.dtype}")


# ------------------------------------------------------------------ rms_norm (ref: layernorm.py:12-83)
class HipRMSNorm(_RefRMSNorm):
    """``RMSNorm`` whose ``forward`` is the gfx950 kernel ``fvk_rmsnorm_rope_bf16`` (norm only: no rotary tables).

    Arithmetic = ``forward_native`` (layernorm.py:48-83): fp32 normalise -> round to bf16 -> multiply by the weight -> round.
    A bf16 weight (what FSDP mixed precision / ``model.to(bf16)`` gives the reference) is multiplied inside the kernel and the
    output is bf16.  An fp32 weight (the reference's default-constructed ``torch.ones`` parameter) follows the reference's type
    promotion instead: ``x.to(bf16) * weight_fp32`` is an UNROUNDED fp32 product, so the kernel runs without a weight and the
    bf16 normalised rows are promoted and multiplied by the fp32 weight — the result is fp32, bit-identical to ``forward_native``'s
    promotion (one extra elementwise pass; not the Wan hot path, whose modules are cast to bf16).  Other weight dtypes are refused.
    ``residual`` and ``var_hidden_size`` (not on the Wan path: wanvideo.py:316-319) are refused."""

    def dispatch_forward(self):
        return self.forward_cuda

    def _weight_bf16(self, device) -> torch.Tensor:
        w = self.weight
        cached = getattr(self, "_hip_w", None)
        if cached is not None and cached[0] == (w.data_ptr(), w._version, w.device, w.dtype) and cached[1].device == device:
            return cached[1]
        if w.dtype != BF16:
            raise RuntimeError(f"HipRMSNorm: the in-kernel weight must be bf16, got {w.dtype}")
        wb = w.detach().to(device=device).contiguous()
        self._hip_w = ((w.data_ptr(), w._version, w.device, w.dtype), wb)
        return wb

    def forward_cuda(self, x: torch.Tensor, residual: torch.Tensor | None = None):
        if residual is not None:
            raise NotImplementedError("HipRMSNorm: the fused-residual form is not on the Wan path")
        if self.variance_size_override is not None:
            raise NotImplementedError("HipRMSNorm: var_hidden_size is not supported")
        _require_rocm_bf16(x, "x")
        if x.shape[-1] != self.hidden_size:
            raise ValueError(f"Expected hidden_size to be {self.hidden_size}, but found: {x.shape[-1]}")
        x2 = x.reshape(-1, self.hidden_size)
        if x2.stride(1) != 1:
            x2 = x2.contiguous()
        w = self.weight if self.has_weight else None
        if w is not None and w.dtype not in (BF16, torch.float32):
            raise RuntimeError(f"HipRMSNorm: weight must be bf16 or fp32, got {w.dtype}")
        in_kernel = w is not None and w.dtype == BF16
        out = ops.rmsnorm_rope([x2], [self._weight_bf16(x.device)] if in_kernel else None, None, None, head_dim=self.hidden_size,
                               seq_len=x2.shape[0], eps=self.variance_epsilon)[0]
        out = out.view(x.shape)
        if w is not None and not in_kernel:  # forward_native: `x.to(orig_dtype) * self.weight` promotes to fp32, product unrounded
            out = out * w.detach().to(x.device)
        return out


# ------------------------------------------------------------------ rotary (ref: rotary_embedding.py:105-150, 153-236)
def _full_tables(cos: torch.Tensor, sin: torch.Tensor, head_size: int):
    """[S, D/2] pair tables -> the kernel's [S, D] layout (each pair's value repeated), fp32 contiguous."""
    if cos.shape[-1] * 2 == head_size:
        cos, sin = cos.repeat_interleave(2, dim=-1), sin.repeat_interleave(2, dim=-1)
    elif cos.shape[-1] != head_size:
        raise ValueError(f"rotary tables of width {cos.shape[-1]} do not fit head_size {head_size}")
    return cos.float().contiguous(), sin.float().contiguous()


def apply_rotary_emb(x: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor, is_neox_style: bool = False) -> torch.Tensor:
    """Drop-in for ``_apply_rotary_emb`` (rotary_embedding.py:105-150) as the Wan path calls it (attention/layer.py:130-132,
    wanvideo.py:679-687): x [..., S, H, D] bf16 and FULL-WIDTH fp32 tables cos/sin [S, D] (each pair's value repeated) — the
    reference's ``rope_dim == head_size`` branch, ``x.float()*cos + rotate(x).float()*sin`` rounded once; ``is_neox_style`` is
    ignored there, as in the reference.  Half-width tables [S, D/2] are served for the interleaved (GPT-J) style, which is the
    same rotation; the Neox half-split style is refused."""
    _require_rocm_bf16(x, "x")
    S, H, D = x.shape[-3], x.shape[-2], x.shape[-1]
    if cos.shape[-1] != D and is_neox_style:
        raise NotImplementedError("apply_rotary_emb: the gfx950 kernel rotates interleaved pairs (the Wan DiT's form), not Neox halves")
    if cos.shape[0] != S:
        raise ValueError(f"rotary tables hold {cos.shape[0]} positions, x has {S} tokens")
    c, s = _full_tables(cos.to(x.device), sin.to(x.device), D)
    x3 = x.reshape(-1, S, H * D)  # [batch, S, H*D]: the position of row m of the flattened [batch*S, H*D] view is m % S
    if x3.stride(2) != 1 or (x3.shape[0] > 1 and x3.stride(0) != S * x3.stride(1)):
        x3 = x3.contiguous()
    x2 = x3.as_strided((x3.shape[0] * S, H * D), (x3.stride(1), 1), x3.storage_offset())  # batches are stride(1)*S apart: one row stride
    out = torch.empty((x3.shape[0] * S, H * D), dtype=BF16, device=x.device)  # explicitly contiguous
    ops.rmsnorm_rope([x2], None, c, s, head_dim=D, seq_len=S, outs=[out])  # ONE launch for every batch element
    return out.view(x.shape)


class HipRotaryEmbedding(_CustomOp):
    """``RotaryEmbedding`` (rotary_embedding.py:153-236) with ``forward_cuda`` on the gfx950 kernel: positions index a
    ``cos_sin_cache`` built exactly as the reference builds it; GPT-J style with ``rotary_dim == head_size`` (the kernel rotates
    whole heads).  Other configurations are refused."""

    def __init__(self, head_size: int, rotary_dim: int, max_position_embeddings: int, base, is_neox_style: bool,
                 dtype: torch.dtype) -> None:
        super().__init__()
        if is_neox_style or rotary_dim != head_size:
            raise ValueError("HipRotaryEmbedding: interleaved (GPT-J) style with rotary_dim == head_size only")
        self.head_size, self.rotary_dim, self.max_position_embeddings = head_size, rotary_dim, max_position_embeddings
        self.base, self.is_neox_style, self.dtype = base, is_neox_style, dtype
        inv_freq = 1.0 / (base**(torch.arange(0, rotary_dim, 2, dtype=torch.float) / rotary_dim))  # ref :181-188
        freqs = torch.einsum("i,j -> ij", torch.arange(max_position_embeddings, dtype=torch.float), inv_freq)
        self.register_buffer("cos_sin_cache", torch.cat((freqs.cos(), freqs.sin()), dim=-1).to(dtype), persistent=False)

    def dispatch_forward(self):
        return self.forward_cuda

    def forward_native(self, positions, query, key, offsets=None):
        """ref: rotary_embedding.py:201-229 restated for this configuration (the test oracle)."""
        if offsets is not None:
            positions = positions + offsets
        positions = positions.flatten()
        cos, sin = self.cos_sin_cache.index_select(0, positions).chunk(2, dim=-1)

        def rot(t):
            shape = t.shape
            t = t.view(positions.shape[0], -1, self.head_size)
            x1, x2 = t[..., ::2], t[..., 1::2]
            c, s = cos.unsqueeze(-2), sin.unsqueeze(-2)
            o1 = (x1.float() * c - x2.float() * s).type_as(t)
            o2 = (x2.float() * c + x1.float() * s).type_as(t)
            return torch.stack((o1, o2), dim=-1).flatten(-2).reshape(shape)

        return rot(query), rot(key)

    def forward_cuda(self, positions, query, key, offsets=None):
        _require_rocm_bf16(query, "query"), _require_rocm_bf16(key, "key")
        if offsets is not None:
            positions = positions + offsets
        positions = positions.flatten().to(query.device)
        n = positions.shape[0]
        cos, sin = self.cos_sin_cache.to(query.device).index_select(0, positions).chunk(2, dim=-1)
        c, s = _full_tables(cos, sin, self.head_size)
        outs = []
        for t in (query, key):
            t2 = t.reshape(n, -1)
            if t2.stride(1) != 1:
                t2 = t2.contiguous()
            outs.append(ops.rmsnorm_rope([t2], None, c, s, head_dim=self.head_size, seq_len=n)[0].view(t.shape))
        return outs[0], outs[1]


# ------------------------------------------------------------------ linear methods (ref: linear.py:80-156, fp8_config.py:71-173)
class HipLinearMethod(_LinearMethodBase):
    """``UnquantizedLinearMethod`` (linear.py:121-156) on the hand-written bf16 MFMA GEMM (``fvk_gemm_bf16``)."""

    def create_weights(self, layer, input_size_per_partition: int, output_partition_sizes: list, input_size: int, output_size: int,
                       params_dtype: torch.dtype, **extra_weight_attrs) -> None:
        weight = Parameter(torch.empty(sum(output_partition_sizes), input_size_per_partition, dtype=params_dtype), requires_grad=False)
        _set_weight_attrs(weight, {"input_dim": 1, "output_dim": 0})
        layer.register_parameter("weight", weight)
        _set_weight_attrs(weight, extra_weight_attrs)

    def apply(self, layer, x: torch.Tensor, bias: torch.Tensor | None = None) -> torch.Tensor:
        _require_rocm_bf16(x, "x"), _require_rocm_bf16(layer.weight, "layer.weight")
        if bias is not None:
            _require_rocm_bf16(bias, "bias")
        x2 = x.reshape(-1, x.shape[-1])
        return ops.gemm(x2, layer.weight, bias).view(*x.shape[:-1], layer.weight.shape[0])


class HipFP8LinearMethod(_QuantizeMethodBase):
    """``FP8QuantizeMethod`` (fp8_config.py:71-173): dynamic activation quantisation (``fvk_fp8_quantize_bf16``, bytes and scales
    bit-identical to ``_quantize_tensorwise/_rowwise``) + e4m3fn MFMA GEMM with the dequantisation scales, bias and bf16 rounding of
    ``torch._scaled_mm(...) + bias`` in the epilogue (``fvk_gemm_fp8``).  Weights: ``convert_model_to_fp8`` below."""

    def __init__(self, granularity: str = "tensor"):
        if granularity not in ("tensor", "channel"):
            raise ValueError(f"granularity must be 'tensor' or 'channel', got {granularity!r}")
        self.granularity = granularity

    create_weights = HipLinearMethod.create_weights

    def quantize_input(self, x: torch.Tensor):
        """ref :105-114 — shared by the q / k / v projections of one block."""
        assert x.dtype in (torch.bfloat16, ), f"only allow bf16 inputs to fp8 linear, got {x.dtype}"
        _require_rocm_bf16(x, "x")
        q, s = ops.fp8_quantize(x.reshape(-1, x.shape[-1]), rowwise=(self.granularity == "channel"))
        return q, s, None

    def wants_prequantized_input(self) -> bool:
        return True

    def apply(self, layer, x: torch.Tensor, bias: torch.Tensor | None = None, pre_quantized=None) -> torch.Tensor:
        if pre_quantized is not None:
            x_fp8, x_scale, _ = pre_quantized
            x_fp8 = x_fp8.reshape(-1, x_fp8.shape[-1])
        else:
            x_fp8, x_scale, _ = self.quantize_input(x)
        if bias is not None:
            _require_rocm_bf16(bias, "bias")
        out = ops.gemm_fp8(x_fp8, x_scale, layer._fp8_weight, layer._fp8_weight_scale, bias)
        return out.view(*x.shape[:-1], layer._fp8_weight.shape[0])


def convert_model_to_fp8(model: nn.Module) -> None:
    """ref: fp8_config.py:211-245 — every layer whose ``quant_method`` is a ``HipFP8LinearMethod`` gets ``_fp8_weight`` /
    ``_fp8_weight_scale`` buffers (quantised on the device by the same kernel as the activations) and loses ``weight``."""
    with torch.no_grad():
        for mod in model.modules():
            qm = getattr(mod, "quant_method", None)
            if not isinstance(qm, HipFP8LinearMethod) or getattr(mod, "weight", None) is None:
                continue
            w = mod.weight.detach()
            _require_rocm_bf16(w, "weight")
            w_fp8, w_scale = ops.fp8_quantize(w.contiguous(), rowwise=(qm.granularity == "channel"))
            mod.register_buffer("_fp8_weight", w_fp8.contiguous(), persistent=False)
            mod.register_buffer("_fp8_weight_scale", w_scale.reshape(-1).to(torch.float32), persistent=False)
            mod._parameters.pop("weight", None)


class Mi355xBf16Config(_QuantizationConfig):
    """A ``QuantizationConfig`` that routes EVERY linear layer to the hand-written bf16 GEMM (no quantisation): the way to put
    ``fvk_gemm_bf16`` behind ``ReplicatedLinear`` without editing it (linear.py:196-206 asks the config per layer)."""

    def get_name(self) -> str:
        return "MI355X_BF16"

    def get_supported_act_dtypes(self) -> list:
        return [torch.bfloat16]

    @classmethod
    def get_min_capability(cls) -> int:
        return 0  # capability numbers are a CUDA notion; gfx950 is checked when libfvk_amd.so is loaded

    @staticmethod
    def get_config_filenames() -> list:
        return []

    @classmethod
    def from_config(cls, config: dict):
        return cls()

    def get_quant_method(self, layer, prefix: str):
        return HipLinearMethod() if isinstance(layer, _LinearBase) else None


class Mi355xFp8Config(Mi355xBf16Config):
    """``FP8Config`` (fp8_config.py:176-208): fp8 for the layers the reference tags by suffix, the bf16 HIP GEMM for the rest."""

    def __init__(self, granularity: str = "tensor"):
        super().__init__()
        if granularity not in ("tensor", "channel"):
            raise ValueError(f"granularity must be 'tensor' or 'channel', got {granularity!r}")
        self.granularity = granularity

    def get_name(self) -> str:
        return "MI355X_FP8"

    @classmethod
    def from_config(cls, config: dict):
        return cls(granularity=config.get("granularity", "tensor"))

    def get_quant_method(self, layer, prefix: str):
        if not isinstance(layer, _LinearBase):
            return None
        if any(s in prefix for s in FP8_SUFFIXES):
            return HipFP8LinearMethod(granularity=self.granularity)
        return HipLinearMethod()


_installed = False


def install() -> dict:
    """The registrations a maintainer adds.  With the reference importable:
    ``CustomOp.op_registry['rms_norm'|'rotary_embedding']`` -> the Hip classes, and the two configs under
    ``get_quantization_config('MI355X_BF16'|'MI355X_FP8')``.  Idempotent; returns what was registered."""
    global _installed
    _CustomOp.op_registry["rms_norm"] = HipRMSNorm
    _CustomOp.op_registry["rotary_embedding"] = HipRotaryEmbedding
    HipRMSNorm.name, HipRotaryEmbedding.name = "rms_norm", "rotary_embedding"
    reg = {"rms_norm": HipRMSNorm, "rotary_embedding": HipRotaryEmbedding}
    if HAVE_REFERENCE and not _installed:
        from fastvideo.layers.quantization import QUANTIZATION_METHODS, register_quantization_config  # type: ignore
        for name, cls in (("MI355X_BF16", Mi355xBf16Config), ("MI355X_FP8", Mi355xFp8Config)):
            if name not in QUANTIZATION_METHODS:
                register_quantization_config(name)(cls)
    _installed = True
    reg.update({"MI355X_BF16": Mi355xBf16Config, "MI355X_FP8": Mi355xFp8Config})
    return reg


__all__ = ["HipRMSNorm", "HipRotaryEmbedding", "apply_rotary_emb", "HipLinearMethod", "HipFP8LinearMethod", "convert_model_to_fp8",
           "Mi355xBf16Config", "Mi355xFp8Config", "install", "HAVE_REFERENCE"]
