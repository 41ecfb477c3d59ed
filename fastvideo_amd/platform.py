"""Platform hook: the reference resolves an attention backend through
``current_platform.get_attn_backend_cls(selected_backend, head_size, dtype) -> "dotted.path.Class"``
(fastvideo/platforms/interface.py:121-125, selector.py:289-292).  Its ROCm platform only knows TORCH_SDPA / FLASH_ATTN and
raises for everything else (fastvideo/platforms/rocm.py:63-112).  ``Mi355xPlatformMixin`` is the override a
maintainer adds: every name below is a class in fastvideo_amd.attention."""
from __future__ import annotations

_BACKENDS = {
    "FLASH_ATTN": "fastvideo_amd.attention.HipDenseAttentionBackend",
    "TORCH_SDPA": "fastvideo_amd.attention.HipDenseAttentionBackend",
    "VIDEO_SPARSE_ATTN": "fastvideo_amd.attention.HipVideoSparseAttentionBackend",
    "SLIDING_TILE_ATTN": "fastvideo_amd.attention.HipSlidingTileAttentionBackend",
}


def get_attn_backend_cls(selected_backend, head_size: int, dtype) -> str:
    """Same contract as ``Platform.get_attn_backend_cls``: returns a qualname, raises ValueError to refuse
    (silent fallback is a bug in the reference's convention, fastvideo/platforms/cuda.py:149-154,182-186)."""
    import torch
    name = getattr(selected_backend, "name", selected_backend) or "FLASH_ATTN"
    if name not in _BACKENDS:
        raise ValueError(f"Invalid attention backend for MI355X: {name}")
    if head_size != 128:
        raise ValueError(f"MI355X HIP attention kernels support head_size 128 only (got {head_size})")
    # The layer asks with its COMPUTE dtype (attention/layer.py:61-62: get_compute_dtype() = torch.get_default_dtype() unless a mixed
    # precision policy is set), i.e. usually fp32 while the activations arrive as bf16 under autocast.  So fp32 is accepted at
    # SELECTION time; at call time the dense Impl casts fp32 tensors through bf16 like FlashAttentionImpl (flash_attn.py:255-266), the
    # sparse Impls raise on anything but bf16 (the reference's sparse kernels are bf16-only, block_sparse_h100.cu:700-717); fp16 is
    # refused outright (a bf16 kernel would silently drop three mantissa bits).
    if dtype not in (torch.bfloat16, torch.float32, None):
        raise ValueError(f"MI355X HIP attention kernels compute in bf16 (got {dtype})")
    return _BACKENDS[name]


class Mi355xPlatformMixin:
    """``class Mi355xPlatform(Mi355xPlatformMixin, RocmPlatform)`` — see INTEGRATION.md."""

    @classmethod
    def get_attn_backend_cls(cls, selected_backend, head_size: int, dtype) -> str:
        return get_attn_backend_cls(selected_backend, head_size, dtype)

    @classmethod
    def get_device_communicator_cls(cls) -> str:
        """ref: Platform.get_device_communicator_cls (fastvideo/platforms/interface.py:240-245, rocm.py:114-116)."""
        return "fastvideo_amd.distributed.Mi355xCommunicator"
