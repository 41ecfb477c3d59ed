"""Attention backends with the reference's backend protocol (fastvideo/attention/backends/abstract.py:31-194):
``AttentionBackend.{get_name,get_impl_cls,get_metadata_cls,get_builder_cls}`` and
``AttentionImpl.{__init__(num_heads, head_size, softmax_scale, causal, num_kv_heads, prefix, **extra), preprocess_qkv,
postprocess_output, forward(q, k, v, attn_metadata)}`` over ``[B, S, H, D]`` tensors — so they slot in behind
``fastvideo.attention.selector.get_attn_backend`` via a Platform's ``get_attn_backend_cls`` (see platform.py and
INTEGRATION.md).  When the reference package is importable the classes subclass its ABCs; otherwise structural twins
defined here are used (the GPU box has no reference checkout)."""
from .backends import (HipDenseAttentionBackend, HipDenseAttentionImpl, HipSlidingTileAttentionBackend,
                       HipSlidingTileAttentionImpl, HipVideoSparseAttentionBackend, HipVideoSparseAttentionImpl,
                       VideoSparseAttentionMetadata, VideoSparseAttentionMetadataBuilder, compute_topk)

__all__ = [
    "HipDenseAttentionBackend", "HipDenseAttentionImpl", "HipVideoSparseAttentionBackend", "HipVideoSparseAttentionImpl",
    "HipSlidingTileAttentionBackend", "HipSlidingTileAttentionImpl", "VideoSparseAttentionMetadata",
    "VideoSparseAttentionMetadataBuilder", "compute_topk"
]
