"""The drop-in boundary executed against the REAL reference (imported on CPU through oracle/ref_loader.py — the live checkout
in the build container, the copy staged by oracle/stage_ref.py elsewhere): INTEGRATION.md §2-§4 as tests.

  * the three ``Hip*Backend`` classes resolve through ``fastvideo.attention.selector.get_attn_backend`` on a platform carrying
    ``Mi355xPlatformMixin`` (fastvideo/attention/selector.py:177-292, fastvideo/platforms/interface.py:121-125) — and refuse, by
    raising, what they cannot serve;
  * the reference's UNMODIFIED ``fastvideo/attention/backends/video_sparse_attn.py`` imports ``fastvideo_amd.kernel_api`` as its
    ``fastvideo_kernel`` (``:8-15``): it binds our functions, our signatures match the wheel's, and its metadata builder produces
    exactly the index tensors ours does;
  * the layer-op classes are what the reference's extension points accept: ``CustomOp`` registry, ``QuantizeMethodBase`` /
    ``QuantizationConfig`` through ``ReplicatedLinear`` and ``get_quantization_config``.

No kernel is launched here (no GPU): compute parity of the same classes is tests/test_gpu_boundary.py."""
import importlib
import inspect
import sys

import pytest
import torch

from oracle import ref_loader as R

pytestmark = pytest.mark.skipif(not R.available(), reason="no reference checkout (live or staged)")


@pytest.fixture(scope="module")
def ref():
    R.init_distributed()
    return R


@pytest.fixture()
def mi355x_platform(ref):
    """``class Mi355xPlatform(Mi355xPlatformMixin, RocmPlatform)`` installed as the reference's current platform."""
    import fastvideo.platforms as P
    from fastvideo.attention import selector
    from fastvideo.platforms.rocm import RocmPlatform
    from fastvideo_amd.platform import Mi355xPlatformMixin

    class Mi355xPlatform(Mi355xPlatformMixin, RocmPlatform):
        pass

    old = P._current_platform
    P._current_platform = Mi355xPlatform()
    selector._cached_get_attn_backend.cache_clear()
    yield P._current_platform
    P._current_platform = old
    selector._cached_get_attn_backend.cache_clear()


def test_selector_resolves_hip_backends(mi355x_platform):
    from fastvideo.attention.backends.abstract import AttentionBackend, AttentionImpl
    from fastvideo.attention.selector import get_attn_backend
    from fastvideo.platforms import AttentionBackendEnum as E
    import fastvideo_amd.attention as A
    importlib.reload(importlib.import_module("fastvideo_amd.attention.backends"))  # (re)bind to the reference's ABCs
    importlib.reload(A)
    sup = (E.FLASH_ATTN, E.TORCH_SDPA, E.VIDEO_SPARSE_ATTN)
    want = {E.FLASH_ATTN: "HipDenseAttentionBackend", E.TORCH_SDPA: "HipDenseAttentionBackend",
            E.VIDEO_SPARSE_ATTN: "HipVideoSparseAttentionBackend", None: "HipDenseAttentionBackend"}
    for req, cls_name in want.items():
        for dtype in (torch.bfloat16, torch.float32):  # layers ask with the compute dtype: fp32 unless a policy is set (layer.py:61-62)
            be = get_attn_backend(128, dtype, supported_attention_backends=sup, requested=req)
            assert be.__name__ == cls_name and be.__module__.startswith("fastvideo_amd.attention")
            assert issubclass(be, AttentionBackend) and issubclass(be.get_impl_cls(), AttentionImpl)
            assert be.get_name() in E.__members__  # layer.py:79 maps the name back to the enum
    # the impl constructor signature the layers use (attention/layer.py:62-70)
    impl = get_attn_backend(128, torch.bfloat16, supported_attention_backends=sup, requested=E.FLASH_ATTN).get_impl_cls()(
        num_heads=12, head_size=128, causal=False, softmax_scale=128**-0.5, num_kv_heads=12, prefix="blocks.0.attn1.impl")
    assert impl.softmax_scale == 128**-0.5
    # refusals are exceptions, never a silent fallback (fastvideo/platforms/cuda.py:149-154)
    with pytest.raises(ValueError):
        get_attn_backend(64, torch.bfloat16, supported_attention_backends=sup, requested=E.FLASH_ATTN)
    with pytest.raises(ValueError):
        get_attn_backend(128, torch.float16, supported_attention_backends=sup, requested=E.FLASH_ATTN)
    with pytest.raises(ValueError):
        get_attn_backend(128, torch.bfloat16, supported_attention_backends=(E.SAGE_ATTN, ), requested=E.SAGE_ATTN)
    with pytest.raises(ValueError):
        impl.__class__(num_heads=12, head_size=128, causal=True, softmax_scale=1.0)
    assert mi355x_platform.get_device_communicator_cls() == "fastvideo_amd.distributed.Mi355xCommunicator"


def test_distributed_attention_layer_constructs_with_hip_backend(mi355x_platform):
    """The reference's own ``DistributedAttention`` / ``LocalAttention`` (attention/layer.py:29-80, 248-286) pick up the HIP impl."""
    from fastvideo.attention.layer import DistributedAttention, LocalAttention
    from fastvideo.platforms import AttentionBackendEnum as E
    sup = (E.FLASH_ATTN, E.TORCH_SDPA)
    for cls in (DistributedAttention, LocalAttention):
        layer = cls(num_heads=12, head_size=128, supported_attention_backends=sup)
        assert type(layer.attn_impl).__name__ == "HipDenseAttentionImpl"
        assert layer.backend == E.FLASH_ATTN


def test_reference_vsa_backend_binds_kernel_api(ref):
    """``sys.modules['fastvideo_kernel'] = fastvideo_amd.kernel_api``: zero reference edits."""
    from fastvideo_amd import kernel_api
    old = sys.modules.get("fastvideo_kernel")
    sys.modules["fastvideo_kernel"] = kernel_api
    try:
        mod = importlib.import_module("fastvideo.attention.backends.video_sparse_attn")
        mod = importlib.reload(mod)
        assert mod.video_sparse_attn is kernel_api.video_sparse_attn
        assert mod.video_sparse_attn_bshd is kernel_api.video_sparse_attn_bshd
        # our signatures carry every parameter of the wheel's, same order and defaults
        wheel = R.load_kernel_module("python/fastvideo_kernel/vsa_utils.py", "_ref_vsa_utils")
        for name in ("get_tile_partition_indices", "get_reverse_tile_partition_indices", "construct_variable_block_sizes",
                     "get_non_pad_index", "build_vsa_metadata"):
            ref_params = list(inspect.signature(getattr(wheel, name)).parameters)
            ours = list(inspect.signature(getattr(kernel_api, name)).parameters)
            assert ours[:len(ref_params)] == ref_params or set(ref_params) <= set(ours), (name, ref_params, ours)
        want_vsa = ["q", "k", "v", "variable_block_sizes", "q_variable_block_sizes", "topk", "block_size", "compress_attn_weight"]
        assert list(inspect.signature(kernel_api.video_sparse_attn).parameters)[:8] == want_vsa
        assert list(inspect.signature(kernel_api.video_sparse_attn_bshd).parameters)[:8] == want_vsa
        assert list(inspect.signature(kernel_api.sliding_tile_attention).parameters)[:7] == [
            "q", "k", "v", "window_size", "text_length", "has_text", "seq_shape"]
        # the reference's metadata builder (unmodified) == ours, tensor for tensor
        from fastvideo_amd.attention import VideoSparseAttentionMetadataBuilder as Ours
        for raw in ((21, 60, 104), (9, 64, 64), (5, 14, 6), (33, 90, 160)):
            a = mod.VideoSparseAttentionMetadataBuilder().build(current_timestep=3, raw_latent_shape=raw, patch_size=(1, 2, 2),
                                                                VSA_sparsity=0.8, device=torch.device("cpu"))
            b = Ours().build(current_timestep=3, raw_latent_shape=raw, patch_size=(1, 2, 2), VSA_sparsity=0.8,
                             device=torch.device("cpu"))
            assert a.dit_seq_shape == b.dit_seq_shape and tuple(a.num_tiles) == tuple(b.num_tiles)
            assert a.total_seq_length == b.total_seq_length and a.VSA_sparsity == b.VSA_sparsity
            for f in ("tile_partition_indices", "reverse_tile_partition_indices", "variable_block_sizes", "non_pad_index",
                      "untile_combined_index"):
                assert torch.equal(getattr(a, f).long(), getattr(b, f).long()), (raw, f)
            # the top-k rule of the reference impl (video_sparse_attn.py:161-163) == ours
            from fastvideo_amd.attention import compute_topk
            assert mod._compute_cur_topk(a) == compute_topk(b.VSA_sparsity, b.variable_block_sizes.numel())
        # the reference's Impl.forward refuses to run without the functions; with them bound it reaches OUR argument checks
        impl = mod.VideoSparseAttentionImpl(num_heads=2, head_size=128, causal=False, softmax_scale=128**-0.5)
        md = mod.VideoSparseAttentionMetadataBuilder().build(current_timestep=0, raw_latent_shape=(4, 8, 8), patch_size=(1, 2, 2),
                                                             VSA_sparsity=0.5, device=torch.device("cpu"))
        x = torch.zeros((1, md.total_seq_length, 2, 128), dtype=torch.bfloat16)
        t = impl.preprocess_qkv(x, md)
        with pytest.raises(RuntimeError, match="ROCm|device"):  # CPU tensors: the HIP path has no fallback
            impl.forward(t, t, t, t, md)
    finally:
        if old is None:
            sys.modules.pop("fastvideo_kernel", None)
        else:
            sys.modules["fastvideo_kernel"] = old
        importlib.reload(importlib.import_module("fastvideo.attention.backends.video_sparse_attn"))


def test_layer_op_classes_plug_into_reference_extension_points(ref):
    import fastvideo_amd.layers as L
    L = importlib.reload(L)  # bind to the reference's bases now that it is importable
    assert L.HAVE_REFERENCE
    from fastvideo.layers.custom_op import CustomOp
    from fastvideo.layers.layernorm import RMSNorm
    from fastvideo.layers.linear import LinearMethodBase, ReplicatedLinear
    from fastvideo.layers.quantization import get_quantization_config
    from fastvideo.layers.quantization.base_config import QuantizationConfig, QuantizeMethodBase
    saved = dict(CustomOp.op_registry)
    try:
        reg = L.install()
        assert CustomOp.op_registry["rms_norm"] is L.HipRMSNorm and issubclass(L.HipRMSNorm, RMSNorm)
        assert CustomOp.op_registry["rotary_embedding"] is L.HipRotaryEmbedding and issubclass(L.HipRotaryEmbedding, CustomOp)
        n = L.HipRMSNorm(1536, eps=1e-6)
        assert n._forward_method == n.forward_cuda  # constructing the class selects the HIP kernel (custom_op.py:21-24)
        assert n.weight.shape == (1536, ) and isinstance(n.weight, torch.nn.Parameter)
        with pytest.raises(RuntimeError, match="ROCm"):
            n(torch.zeros(2, 3, 1536, dtype=torch.bfloat16))  # no CPU fallback
        # forward_native is still the reference's: the oracle stays callable on the same module
        assert torch.equal(n.forward_native(torch.ones(1, 2, 1536)), RMSNorm(1536).forward_native(torch.ones(1, 2, 1536)))
        assert issubclass(L.HipLinearMethod, LinearMethodBase) and issubclass(L.HipFP8LinearMethod, QuantizeMethodBase)
        assert get_quantization_config("MI355X_BF16") is L.Mi355xBf16Config is reg["MI355X_BF16"]
        assert get_quantization_config("MI355X_FP8") is L.Mi355xFp8Config
        assert issubclass(L.Mi355xFp8Config, QuantizationConfig)
        # ReplicatedLinear asks the config per layer (linear.py:196-206) and lets the method create its weights (:264-272)
        lin = ReplicatedLinear(1536, 4608, bias=True, params_dtype=torch.bfloat16, quant_config=L.Mi355xBf16Config(),
                               prefix="blocks.0.to_q")
        assert isinstance(lin.quant_method, L.HipLinearMethod)
        assert lin.weight.shape == (4608, 1536) and lin.weight.dtype == torch.bfloat16 and lin.bias.shape == (4608, )
        assert lin.weight.weight_loader == lin.weight_loader  # extra_weight_attrs honoured (loader hooks keep working)
        cfg = L.Mi355xFp8Config(granularity="channel")
        tagged = ReplicatedLinear(1536, 1536, params_dtype=torch.bfloat16, quant_config=cfg, prefix="blocks.3.ffn.fc_in")
        plain = ReplicatedLinear(1536, 1536, params_dtype=torch.bfloat16, quant_config=cfg, prefix="blocks.3.to_gate_compress")
        assert isinstance(tagged.quant_method, L.HipFP8LinearMethod) and tagged.quant_method.granularity == "channel"
        assert tagged.quant_method.wants_prequantized_input()
        assert isinstance(plain.quant_method, L.HipLinearMethod)  # untagged layers: bf16 HIP GEMM (fp8_config.py:203-208 returns None)
        with pytest.raises(RuntimeError, match="ROCm"):
            lin(torch.zeros(4, 1536, dtype=torch.bfloat16))
        with pytest.raises(ValueError):
            L.Mi355xFp8Config(granularity="block")
    finally:
        CustomOp.op_registry.clear()
        CustomOp.op_registry.update(saved)
