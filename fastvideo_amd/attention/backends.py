from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Any

import torch

from .. import kernel_api, ops

try:  # subclass the reference's ABCs when it is installed (drop-in registration), else use local twins
    from fastvideo.attention.backends.abstract import (AttentionBackend, AttentionImpl, AttentionMetadata,  # type: ignore
                                                       AttentionMetadataBuilder)
except Exception:  # noqa: BLE001 - the GPU box / CI has no reference package

    class AttentionBackend:  # ref: abstract.py:31-66
        accept_output_buffer: bool = False

    @dataclass
    class AttentionMetadata:  # ref: abstract.py:69-86
        current_timestep: int
        VSA_sparsity: float = field(default=0.0, kw_only=True)

    class AttentionMetadataBuilder:  # ref: abstract.py:92-113
        pass

    class AttentionImpl:  # ref: abstract.py:133-194

        def preprocess_qkv(self, qkv, attn_metadata):
            return qkv

        def postprocess_output(self, output, attn_metadata):
            return output


VSA_TILE_SIZE = (4, 4, 4)


def _require_bf16_cuda(*ts):
    for t in ts:
        if not t.is_cuda or t.dtype != torch.bfloat16:
            raise RuntimeError("fastvideo_amd attention backends take bf16 ROCm tensors [B,S,H,128] "
                               f"(got {t.dtype} on {t.device}); there is no eager fallback")


def _i32(t: torch.Tensor, device) -> torch.Tensor:
    """Index tensors of the reference's metadata are int64 (vsa_utils.py); the kernels take int32 on the device."""
    return t if (t.dtype == torch.int32 and t.device == device) else t.to(device=device, dtype=torch.int32)


# ------------------------------------------------------------------ dense (replaces SDPA / flash-attn on ROCm)
class HipDenseAttentionImpl(AttentionImpl):
    """ref: SDPAImpl (fastvideo/attention/backends/sdpa.py:108-147) / FlashAttentionImpl (flash_attn.py:247-345)."""

    def __init__(self, num_heads: int, head_size: int, causal: bool = False, softmax_scale: float | None = None,
                 num_kv_heads: int | None = None, prefix: str = "", **extra_impl_args) -> None:
        if causal:
            raise ValueError("HipDenseAttentionImpl: causal attention is not on the Wan T2V path")
        if head_size != 128:
            raise ValueError(f"HipDenseAttentionImpl: head_size {head_size} unsupported (128 only)")
        if num_kv_heads not in (None, num_heads):
            raise ValueError("HipDenseAttentionImpl: grouped-query attention is not supported")
        self.softmax_scale = head_size**-0.5 if softmax_scale is None else softmax_scale

    @staticmethod
    def _key_padding_mask(attn_mask, key) -> torch.Tensor:
        """The reference's mask normalisation (``_normalize_attn_mask_for_sdpa``, sdpa.py:70-103) restricted to what a KEY-PADDING mask is:
        bool / integer [B, Skv] (or [B, 1, 1, Skv] / [B, 1, Skv]), True / non-zero = attend; a floating mask must hold only 0 and -inf.
        Shorter masks are front-padded with "attend" exactly as the reference does (:88-92).  Returns bool [B, Skv].  Masks that differ per
        query row or per head are refused (the Wan / FastWan pipelines only ever pass tokenizer padding masks)."""
        B, Skv = key.shape[0], key.shape[1]
        m = attn_mask.to(device=key.device)
        if m.dim() == 4:
            if m.shape[1] != 1 or m.shape[2] != 1:
                raise NotImplementedError(f"HipDenseAttentionImpl: only key-padding masks [B, 1, 1, Skv] are supported, got {tuple(m.shape)}")
            m = m[:, 0, 0]
        elif m.dim() == 3:
            if m.shape[1] != 1:
                raise NotImplementedError(f"HipDenseAttentionImpl: only key-padding masks [B, 1, Skv] are supported, got {tuple(m.shape)}")
            m = m[:, 0]
        elif m.dim() != 2:
            raise ValueError(f"Unsupported attention mask shape for SDPA: {tuple(attn_mask.shape)}")
        if m.dtype.is_floating_point:
            # additive masks: 0 = attend; -inf, or the "large negative" HF pipelines write instead (finfo.min of the mask dtype, anything
            # <= finfo.min / 2), = masked — after softmax both are exactly zero weight.  Any other value is a bias, not a padding mask
            masked = m <= torch.finfo(m.dtype).min / 2
            if not bool(((m == 0) | masked).all()):
                raise NotImplementedError("HipDenseAttentionImpl: additive masks other than 0 / -inf (or finfo.min) are not supported")
            m = ~masked
        elif m.dtype != torch.bool:
            m = m != 0
        if m.shape[-1] > Skv:
            raise ValueError(f"Invalid attention mask length for SDPA: expected at most {Skv}, got {m.shape[-1]}")
        if m.shape[-1] < Skv:
            m = torch.nn.functional.pad(m, (Skv - m.shape[-1], 0), value=True)
        if m.shape[0] == 1 and B > 1:
            m = m.expand(B, Skv)
        if m.shape[0] != B:
            raise ValueError(f"attention mask batch {m.shape[0]} != {B}")
        return m

    def _forward_key_padding(self, query, key, value, mask):
        """Key-padding mask [B, Skv] (sdpa.py:134-147; flash_attn.py:279-330's varlen branch): every batch element attends to its valid keys
        only.  The kernels take a key COUNT, so a sample whose valid keys are a prefix (padding at the end: the tokenizer case) runs in place
        on ``key[b, :n]``; a mask with holes compacts that sample's K / V rows first (one gather pass).  One launch per sample; the counts
        come to the host once per call (one synchronisation — this is not the Wan T2V hot path, which passes no mask).  A sample with no valid
        key yields NaN rows in the reference (softmax over an empty set); refused here.  Self-attention (Sq == Skv): padded QUERY rows come
        back as zeros, as from the reference's flash_attn_no_pad; cross-attention (every query valid): all rows computed."""
        B = query.shape[0]
        counts = mask.sum(dim=1).tolist()
        prefix = (mask.to(torch.int8).diff(dim=1) <= 0).all(dim=1).tolist()   # no False -> True transition: valid keys first
        out = torch.empty_like(query)
        for b in range(B):
            n = int(counts[b])
            if n == 0:
                raise ValueError(f"HipDenseAttentionImpl: sample {b} has no valid key (the reference's softmax would return NaN rows)")
            if prefix[b]:
                kb, vb = key[b:b + 1, :n], value[b:b + 1, :n]
            else:
                idx = torch.nonzero(mask[b], as_tuple=False).flatten().to(torch.int32)
                kb = ops.gather_rows(key[b:b + 1], n, src_index=idx)
                vb = ops.gather_rows(value[b:b + 1], n, src_index=idx)
            ops.attn_dense(query[b:b + 1], kb, vb, scale=self.softmax_scale, layout="bshd", out=out[b:b + 1])
        if query.shape[1] == key.shape[1]:
            # self-attention: flash_attn_no_pad unpads q with the SAME mask and pad_input returns ZERO rows for the padded queries
            # (flash_attn.py:322-330); torch SDPA would compute them.  This backend answers to the name FLASH_ATTN: zeros.
            out.masked_fill_(~mask[:, :, None, None], 0)
        return out

    def forward(self, query, key, value, attn_metadata=None):
        attn_mask = getattr(attn_metadata, "attn_mask", None) if attn_metadata is not None else None
        # ref: FlashAttentionImpl.forward (flash_attn.py:255-266): non-half activations that leak into attention are cast through
        # bf16 for the kernel and restored on output (the reference's own tests run fp32 tensors through this backend)
        orig_dtype = query.dtype
        if orig_dtype != torch.bfloat16:
            if orig_dtype != torch.float32:  # fp16 would silently lose mantissa bits in a bf16 kernel: refused
                raise RuntimeError(f"HipDenseAttentionImpl: unsupported dtype {orig_dtype} (bf16, or fp32 cast through bf16)")
            query, key, value = query.to(torch.bfloat16), key.to(torch.bfloat16), value.to(torch.bfloat16)
        _require_bf16_cuda(query, key, value)
        if attn_mask is not None:
            out = self._forward_key_padding(query, key, value, self._key_padding_mask(attn_mask, key))
        else:
            out = ops.attn_dense(query, key, value, scale=self.softmax_scale, layout="bshd")
        return out if out.dtype == orig_dtype else out.to(orig_dtype)


class HipDenseAttentionBackend(AttentionBackend):
    accept_output_buffer: bool = True

    @staticmethod
    def get_supported_head_sizes() -> list[int]:
        return [128]

    @staticmethod
    def get_name() -> str:
        return "FLASH_ATTN"  # an existing AttentionBackendEnum member (platforms/interface.py:13-27): no enum change needed

    @staticmethod
    def get_impl_cls():
        return HipDenseAttentionImpl

    @staticmethod
    def get_metadata_cls():
        return AttentionMetadata

    @staticmethod
    def get_builder_cls():
        return None


# ------------------------------------------------------------------ VSA (ref: backends/video_sparse_attn.py)
def compute_topk(sparsity: float, num_blocks: int) -> int:
    """ref: video_sparse_attn.py:161-163."""
    return max(1, min(math.ceil((1 - sparsity) * num_blocks), num_blocks))


@dataclass
class VideoSparseAttentionMetadata(AttentionMetadata):  # ref: video_sparse_attn.py:139-158
    current_timestep: int
    dit_seq_shape: tuple
    num_tiles: tuple
    total_seq_length: int
    tile_partition_indices: torch.Tensor
    reverse_tile_partition_indices: torch.Tensor
    variable_block_sizes: torch.Tensor
    non_pad_index: torch.Tensor
    untile_combined_index: torch.Tensor
    tile_buf: torch.Tensor | None = None
    cache_tile_buf: bool = True


class VideoSparseAttentionMetadataBuilder(AttentionMetadataBuilder):  # ref: video_sparse_attn.py:192-235

    def __init__(self) -> None:
        pass

    def prepare(self) -> None:
        pass

    def build(self, current_timestep: int, raw_latent_shape, patch_size, VSA_sparsity: float, device,
              cache_tile_buf: bool = True, **kwargs: Any) -> VideoSparseAttentionMetadata:
        shape = tuple(r // p for r, p in zip(raw_latent_shape, patch_size))
        m = ops.vsa_build_metadata_host(shape, VSA_TILE_SIZE)  # pure-integer C ABI call, bit-exact vs the reference
        to = lambda t: t.to(device)
        return VideoSparseAttentionMetadata(
            current_timestep=current_timestep, dit_seq_shape=shape, VSA_sparsity=VSA_sparsity, num_tiles=m["num_tiles"],
            total_seq_length=math.prod(shape), tile_partition_indices=to(m["tile_partition_indices"]),
            reverse_tile_partition_indices=to(m["reverse_tile_partition_indices"]),
            variable_block_sizes=to(m["variable_block_sizes"]), non_pad_index=to(m["non_pad_index"]),
            untile_combined_index=to(m["untile_combined_index"]), cache_tile_buf=cache_tile_buf)


class HipVideoSparseAttentionImpl(AttentionImpl):
    """ref: VideoSparseAttentionImpl (video_sparse_attn.py:238-342): preprocess_qkv = tile, forward = video_sparse_attn,
    postprocess_output = untile.  Index tensors are int32 on device (the reference uses int64)."""

    def __init__(self, num_heads: int, head_size: int, causal: bool = False, softmax_scale: float | None = None,
                 num_kv_heads: int | None = None, prefix: str = "", **extra_impl_args) -> None:
        if head_size != 128:
            raise ValueError(f"HipVideoSparseAttentionImpl: head_size {head_size} unsupported (128 only)")
        self.prefix = prefix

    def tile(self, x, md: VideoSparseAttentionMetadata):
        """ref: VideoSparseAttentionImpl.tile (video_sparse_attn.py:254-281): zero-padded tile-major copy of [B,S,H,D]; with
        ``cache_tile_buf`` the buffer lives on the per-step metadata (pad rows are zeroed once, never written afterwards)."""
        s_pad = math.prod(md.num_tiles) * math.prod(VSA_TILE_SIZE)
        perm, npi = _i32(md.tile_partition_indices, x.device), _i32(md.non_pad_index, x.device)
        if not getattr(md, "cache_tile_buf", False):
            return ops.gather_rows(x, s_pad, perm, npi, zero_init=True)
        buf = getattr(md, "tile_buf", None)
        shape = (x.shape[0], s_pad, *x.shape[2:])
        if buf is None or tuple(buf.shape) != shape or buf.dtype != x.dtype or buf.device != x.device:
            buf = torch.zeros(shape, dtype=x.dtype, device=x.device)
            md.tile_buf = buf
        return ops.gather_rows(x, s_pad, perm, npi, out=buf)

    def untile(self, x, md: VideoSparseAttentionMetadata):
        total = getattr(md, "total_seq_length", None) or md.untile_combined_index.numel()
        return ops.gather_rows(x, total, _i32(md.untile_combined_index, x.device), None)

    def preprocess_qkv(self, qkv, attn_metadata):
        return self.tile(qkv, attn_metadata)

    def postprocess_output(self, output, attn_metadata):
        return self.untile(output, attn_metadata)

    def forward(self, query, key, value, gate_compress, attn_metadata):
        _require_bf16_cuda(query, key, value)
        md = attn_metadata
        topk = compute_topk(md.VSA_sparsity, md.variable_block_sizes.numel())
        # the kernels take strides: the [B, S_pad, H, D] tensors go in as they are — the reference's four
        # ``transpose(1, 2).contiguous()`` copies per layer (video_sparse_attn.py:296-303) have no counterpart
        if gate_compress is not None:
            _require_bf16_cuda(gate_compress)
        return kernel_api.video_sparse_attn_bshd(query, key, value, md.variable_block_sizes, md.variable_block_sizes, topk,
                                                 block_size=VSA_TILE_SIZE, compress_attn_weight=gate_compress)


class HipVideoSparseAttentionBackend(AttentionBackend):
    accept_output_buffer: bool = True

    @staticmethod
    def get_supported_head_sizes() -> list[int]:
        return [128]

    @staticmethod
    def get_name() -> str:
        return "VIDEO_SPARSE_ATTN"

    @staticmethod
    def get_impl_cls():
        return HipVideoSparseAttentionImpl

    @staticmethod
    def get_metadata_cls():
        return VideoSparseAttentionMetadata

    @staticmethod
    def get_builder_cls():
        return VideoSparseAttentionMetadataBuilder


# ------------------------------------------------------------------ STA (the archived SLIDING_TILE_ATTN backend)
class HipSlidingTileAttentionImpl(AttentionImpl):
    """Tokens are permuted raster -> tile-major (tile (6,8,8) by default) in preprocess_qkv and back in postprocess_output;
    the canvas must be divisible by the tile (the reference kernels hard-code three such canvases)."""

    def __init__(self, num_heads: int, head_size: int, causal: bool = False, softmax_scale: float | None = None,
                 num_kv_heads: int | None = None, prefix: str = "", *, canvas_thw=None, tile_thw=(6, 8, 8),
                 window_size=None, **extra_impl_args) -> None:
        if head_size != 128:
            raise ValueError(f"HipSlidingTileAttentionImpl: head_size {head_size} unsupported (128 only)")
        if canvas_thw is None or any(c % t for c, t in zip(canvas_thw, tile_thw)):
            raise ValueError(f"HipSlidingTileAttentionImpl: canvas {canvas_thw} must be divisible by tile {tile_thw}")
        self.canvas, self.tile_thw = tuple(canvas_thw), tuple(tile_thw)
        self.windows = list(window_size) if window_size is not None else [(3, 3, 3)] * num_heads
        if len(self.windows) != num_heads:
            raise ValueError("HipSlidingTileAttentionImpl: window_size must list one (t,h,w) per head")
        m = ops.vsa_build_metadata_host(self.canvas, self.tile_thw)  # same raster->tile permutation as VSA, other tile
        self._perm, self._rev = m["tile_partition_indices"], m["reverse_tile_partition_indices"]

    def _idx(self, name, device):
        t = getattr(self, name)
        if t.device != device:
            t = t.to(device)
            setattr(self, name, t)
        return t

    def preprocess_qkv(self, qkv, attn_metadata=None):
        return ops.gather_rows(qkv, qkv.shape[1], self._idx("_perm", qkv.device), None)

    def postprocess_output(self, output, attn_metadata=None):
        return ops.gather_rows(output, output.shape[1], self._idx("_rev", output.device), None)

    def forward(self, query, key, value, attn_metadata=None):
        _require_bf16_cuda(query, key, value)
        tiles = tuple(c // t for c, t in zip(self.canvas, self.tile_thw))
        return kernel_api.sliding_tile_attention_canvas(query, key, value, tiles, math.prod(self.tile_thw), self.windows, layout="bshd")


class HipSlidingTileAttentionBackend(AttentionBackend):
    accept_output_buffer: bool = True

    @staticmethod
    def get_supported_head_sizes() -> list[int]:
        return [128]

    @staticmethod
    def get_name() -> str:
        return "SLIDING_TILE_ATTN"

    @staticmethod
    def get_impl_cls():
        return HipSlidingTileAttentionImpl

    @staticmethod
    def get_metadata_cls():
        return AttentionMetadata

    @staticmethod
    def get_builder_cls():
        return None
