"""Compute parity of the drop-in boundary classes on the MI355X — the layer the reference actually calls
(SURVEY.md §8b): every ``Hip*Impl`` through ``preprocess_qkv -> forward -> postprocess_output`` on [B,S,H,D] tensors that are
strided chunks of a stacked buffer (as ``DistributedAttention.forward`` hands them over, attention/layer.py:117-158, 214-240), the
``fastvideo_kernel``-shaped functions of ``kernel_api`` that no other test calls, and the layer-op classes of
``fastvideo_amd.layers`` against their own ``forward_native`` (the reference's eager arithmetic).

Tolerances: attention max |err| < 4e-2 (fastvideo-kernel/tests/test_sta.py:88-91) and a mean bound; fused elementwise ops
atol = rtol = 1e-2 (fastvideo-kernel/tests/test_turbodiffusion.py:143); fp8 bytes / scales bit-exact."""
import math

import numpy as np
import pytest
import torch

from oracle import vsa_oracle as V
from oracle import wan_oracle as W

pytestmark = pytest.mark.gpu
DEV = "cuda"


def g(seed):
    return torch.Generator().manual_seed(seed)


def rnd(shape, seed, scale=1.0, dtype=torch.bfloat16):
    return (torch.randn(shape, generator=g(seed)) * scale).to(dtype)


def _attn_check(out, ref, what, mean_tol=None):
    """max |err| < 4e-2 (test_sta.py:88-91); mean |err| < 3e-3 x mean |ref| + 2e-5 — 1.4x the bf16 rounding of the output, which is what
    the kernels measure (2.1e-3 x mean |ref|); ``mean_tol`` widens the relative factor for composites with two bf16 outputs summed."""
    err = (out.float().cpu() - ref.float()).abs()
    assert torch.isfinite(out.float()).all(), f"{what}: non-finite output"
    bound = (mean_tol or 3e-3) * ref.float().abs().mean().item() + 2e-5
    assert err.max().item() < 4e-2 and err.mean().item() < bound, f"{what}: max {err.max().item():.4g} mean {err.mean().item():.4g} (bound {bound:.4g})"


def _same_rounding(got, ref, what):
    """Ops with the reference's rounding points: atol = rtol = 1e-2 everywhere, and all but a sliver of the elements bit-identical
    (an fp32 summation-order / fma difference can move a value across a bf16 rounding boundary)."""
    got, ref = got.float().cpu(), ref.float().cpu()
    err = (got - ref).abs()
    assert (err <= 1e-2 + 1e-2 * ref.abs()).all(), f"{what}: max err {err.max().item():.4g}"
    frac = (got != ref).float().mean().item()
    assert frac < 5e-3, f"{what}: {frac:.3%} of the elements differ from the eager result"


# ------------------------------------------------------------------ attention backends (fastvideo_amd/attention/backends.py)
def test_dense_impl_on_strided_chunks_and_fp32_inputs():
    """qkv = cat([q,k,v]) -> preprocess_qkv -> chunk(3) (layer.py:117-145): the Impl sees NON-contiguous-in-batch chunks of one
    buffer; fp32 tensors are cast through bf16 and restored (flash_attn.py:255-266)."""
    from fastvideo_amd.attention import HipDenseAttentionBackend
    B, S, H, D = 2, 333, 3, 128
    impl = HipDenseAttentionBackend.get_impl_cls()(num_heads=H, head_size=D, causal=False, softmax_scale=D**-0.5, num_kv_heads=H,
                                                  prefix="blocks.0.attn1.impl")
    qkv = rnd((3 * B, S, H, D), 1).to(DEV)
    qkv = impl.preprocess_qkv(qkv, None)
    q, k, v = qkv.chunk(3, dim=0)
    out = impl.postprocess_output(impl.forward(q, k, v, None), None)
    ref = W.attention_fp32_ref(q.cpu().transpose(1, 2), k.cpu().transpose(1, 2), v.cpu().transpose(1, 2), D**-0.5).transpose(1, 2)
    assert out.shape == q.shape and out.dtype == torch.bfloat16
    _attn_check(out, ref, "dense impl (bf16 chunks)")
    # a fused [B,S,3,H,D] projection viewed per tensor: row stride 3*H*D
    fused = rnd((B, S, 3, H, D), 2).to(DEV)
    out2 = impl.forward(fused[:, :, 0], fused[:, :, 1], fused[:, :, 2], None)
    f = fused.cpu()
    ref2 = W.attention_fp32_ref(f[:, :, 0].transpose(1, 2), f[:, :, 1].transpose(1, 2), f[:, :, 2].transpose(1, 2), D**-0.5).transpose(1, 2)
    _attn_check(out2, ref2, "dense impl (strided views of a fused buffer)")
    # fp32 in -> fp32 out, numerically the bf16 kernel's result
    out32 = impl.forward(q.float(), k.float(), v.float(), None)
    assert out32.dtype == torch.float32 and torch.equal(out32, impl.forward(q, k, v, None).float())
    with pytest.raises(RuntimeError):
        impl.forward(q.half(), k.half(), v.half(), None)
    with pytest.raises(RuntimeError):
        impl.forward(q.cpu(), k.cpu(), v.cpu(), None)


def test_dense_impl_key_padding_masks():
    """Key-padding masks through ``attn_metadata.attn_mask`` (SDPAImpl.forward + ``_normalize_attn_mask_for_sdpa``, sdpa.py:70-147; the varlen
    branch of flash_attn.py:279-330): [B, Skv] bool / int64 tokenizer masks, the [B, 1, 1, Skv] form, a 0 / -inf additive mask, a SHORTER
    mask (front-padded with "attend", :88-92), padding at the end (in place, a key count) and a mask with holes (compacted rows) — against
    torch SDPA in fp32 with the same mask.  Per-query / per-head masks are refused with the reason."""
    import types
    import torch.nn.functional as F
    from fastvideo_amd.attention import HipDenseAttentionBackend
    B, Sq, Skv, H, D = 2, 300, 700, 3, 128
    impl = HipDenseAttentionBackend.get_impl_cls()(num_heads=H, head_size=D, causal=False, softmax_scale=D**-0.5, num_kv_heads=H)
    q, k, v = rnd((B, Sq, H, D), 1).to(DEV), rnd((B, Skv, H, D), 2).to(DEV), rnd((B, Skv, H, D), 3).to(DEV)
    tail = torch.ones((B, Skv), dtype=torch.bool)
    tail[1, 413:] = False                                  # sample 1: padding at the end (sample 0: nothing masked)
    holes = tail.clone()
    holes[0, 5:77] = False
    holes[0, 600] = False
    holes[1, ::3] = False

    def ref(mask_bool):
        m4 = mask_bool[:, None, None, :]
        return F.scaled_dot_product_attention(q.float().cpu().transpose(1, 2), k.float().cpu().transpose(1, 2), v.float().cpu().transpose(1, 2),
                                              attn_mask=m4, scale=D**-0.5).transpose(1, 2)

    md = lambda m: types.SimpleNamespace(attn_mask=m)
    for name, mask in (("tail", tail), ("holes", holes)):
        r = ref(mask)
        for form, m in (("bool [B,Skv]", mask), ("int64", mask.long()), ("[B,1,1,Skv]", mask[:, None, None, :].to(DEV)),
                        ("additive 0/-inf", torch.zeros((B, Skv)).masked_fill(~mask, float("-inf")))):
            _attn_check(impl.forward(q, k, v, md(m)), r, f"key-padding mask {name}, {form}")
        # the "large negative" additive masks HF pipelines write instead of -inf (finfo.min of the mask dtype): masked all the same
        for dt in (torch.float32, torch.bfloat16):
            big = torch.zeros((B, Skv), dtype=dt).masked_fill(~mask, torch.finfo(dt).min)
            assert torch.equal(impl.forward(q, k, v, md(big)), impl.forward(q, k, v, md(mask))), (name, dt)
    with pytest.raises(NotImplementedError, match="additive"):
        impl.forward(q, k, v, md(torch.full((B, Skv), -1.5)))                 # a bias is not a padding mask
    # SELF-attention (Sq == Skv): the reference's flash_attn_no_pad unpads the queries with the same mask and pads the output back with
    # ZERO rows (flash_attn.py:322-330) — valid rows as SDPA, padded rows exactly zero
    qs = rnd((B, Skv, H, D), 9).to(DEV)
    o_self = impl.forward(qs, k, v, md(tail))
    r_self = F.scaled_dot_product_attention(qs.float().cpu().transpose(1, 2), k.float().cpu().transpose(1, 2), v.float().cpu().transpose(1, 2),
                                            attn_mask=tail[:, None, None, :], scale=D**-0.5).transpose(1, 2)
    _attn_check(o_self[1:, :413], r_self[1:, :413], "self-attention key padding, valid rows")
    _attn_check(o_self[:1], r_self[:1], "self-attention key padding, unmasked sample")
    assert (o_self[1, 413:] == 0).all(), "padded query rows must be zero (flash_attn_no_pad / pad_input)"
    # a shorter mask covers the LAST keys; the keys in front of it are attended (front-pad, sdpa.py:88-92)
    short = torch.ones((B, 500), dtype=torch.bool)
    short[0, 450:] = False
    full = torch.cat([torch.ones((B, 200), dtype=torch.bool), short], 1)
    _attn_check(impl.forward(q, k, v, md(short)), ref(full), "short mask, front-padded")
    assert torch.equal(impl.forward(q, k, v, md(torch.ones((B, Skv), dtype=torch.bool))), impl.forward(q, k, v, None))   # all-true == no mask
    with pytest.raises(NotImplementedError, match="key-padding"):
        impl.forward(q, k, v, md(torch.ones((B, 1, Sq, Skv), dtype=torch.bool)))
    with pytest.raises(ValueError, match="no valid key"):
        impl.forward(q, k, v, md(torch.zeros((B, Skv), dtype=torch.bool)))
    with pytest.raises(ValueError, match="expected at most"):
        impl.forward(q, k, v, md(torch.ones((B, Skv + 1), dtype=torch.bool)))


def test_vsa_impl_tile_forward_untile_matches_oracle():
    """``HipVideoSparseAttentionImpl`` driven exactly as ``DistributedAttention_VSA.forward`` drives the reference impl
    (layer.py:214-240): qkvg stacked on the batch axis -> preprocess_qkv (tile) -> chunk(4) -> forward -> postprocess_output."""
    from fastvideo_amd import kernel_api as KA
    from fastvideo_amd.attention import HipVideoSparseAttentionBackend, compute_topk
    raw = (8, 20, 14)  # dit grid (8,10,7): ragged tiles (2,3,2) = 12 blocks
    B, H, D = 1, 2, 128
    be = HipVideoSparseAttentionBackend
    md = be.get_builder_cls()().build(current_timestep=0, raw_latent_shape=raw, patch_size=(1, 2, 2), VSA_sparsity=0.5, device=DEV)
    S = md.total_seq_length
    impl = be.get_impl_cls()(num_heads=H, head_size=D, causal=False, softmax_scale=D**-0.5, num_kv_heads=H, prefix="blocks.0.attn1.impl")
    qkvg = torch.cat([rnd((B, S, H, D), s) for s in (1, 2, 3, 4)], 0).to(DEV)
    tiled = impl.preprocess_qkv(qkvg, md)
    assert md.tile_buf is tiled  # cached on the per-step metadata like the reference's tile_buf (video_sparse_attn.py:254-281)
    tq, tk, tv, tg = tiled.chunk(4, dim=0)
    out_t = impl.forward(tq, tk, tv, tg, md)
    out = impl.postprocess_output(out_t, md)
    assert out.shape == (B, S, H, D)
    # oracle: the integer metadata and the tile permutation are bit-exact ...
    om = V.build_metadata(raw)
    assert np.array_equal(md.tile_partition_indices.cpu().numpy(), om["tile_partition_indices"])
    q, k, v, gate = (t.cpu() for t in qkvg.chunk(4, dim=0))
    assert torch.equal(tq.cpu(), V.tile(q, om)) and torch.equal(tg.cpu(), V.tile(gate, om))
    # ... the block selection is the exact top-k of OUR coarse scores (ties -> lowest index), checked unconditionally by feeding
    # the GPU's mask to the oracle composite: the two can then differ only by kernel arithmetic
    vbs = om["variable_block_sizes"]
    topk = compute_topk(0.5, len(vbs))
    same, inter = KA._vsa_forward(tq, tk, tv, md.variable_block_sizes.int(), md.variable_block_sizes.int(), topk, tg, "bshd", True)
    assert torch.equal(same, out_t)  # deterministic: same kernels, same inputs
    mask = inter["mask"].cpu().numpy()
    assert np.array_equal(mask, V.topk_mask_bisect(inter["scores"].float().cpu().numpy(), topk))
    assert (mask.sum(-1) == topk).all()
    t = lambda z: V.tile(z, om).transpose(1, 2).contiguous()
    ref_t, _ = V.video_sparse_attn(t(q), t(k), t(v), vbs, vbs, topk, 64, t(gate), mask_override=mask)
    ref = V.untile(ref_t.transpose(1, 2), om)
    _attn_check(out, ref, "vsa impl composite (GPU mask)", mean_tol=5e-3)
    # the reference-layout entry of the kernel package gives the same numbers as the strided one
    o_bhsd = KA.video_sparse_attn(tq.transpose(1, 2).contiguous(), tk.transpose(1, 2).contiguous(), tv.transpose(1, 2).contiguous(),
                                  md.variable_block_sizes, md.variable_block_sizes, topk, (4, 4, 4), tg.transpose(1, 2).contiguous())
    assert torch.equal(o_bhsd.transpose(1, 2), out_t)


def test_sta_impl_permute_forward_unpermute_matches_masked_oracle():
    from fastvideo_amd.attention import HipSlidingTileAttentionBackend
    canvas, tile, wins = (12, 16, 24), (6, 8, 8), [(3, 3, 3), (1, 3, 1), (3, 1, 3)]
    B, H, D, S = 1, 3, 128, math.prod(canvas)
    impl = HipSlidingTileAttentionBackend.get_impl_cls()(num_heads=H, head_size=D, causal=False, softmax_scale=D**-0.5, num_kv_heads=H,
                                                        canvas_thw=canvas, tile_thw=tile, window_size=wins)
    qkv = torch.cat([rnd((B, S, H, D), s) for s in (1, 2, 3)], 0).to(DEV)   # raster token order, as the model holds it
    p = impl.preprocess_qkv(qkv)
    q, k, v = p.chunk(3, dim=0)
    out = impl.postprocess_output(impl.forward(q, k, v))
    perm = torch.from_numpy(V.tile_partition_indices(canvas, tile).astype(np.int64))
    assert torch.equal(q.cpu(), qkv[:B].cpu()[:, perm])
    qc, kc, vc = (t.cpu().transpose(1, 2) for t in (q, k, v))  # tile-major [B,H,S,D]
    rev = torch.argsort(perm)
    for h, w in enumerate(wins):
        mask = V.sta_mask(canvas, w, tile)
        ref_t = W.attention_fp32_ref(qc[:, h:h + 1], kc[:, h:h + 1], vc[:, h:h + 1], D**-0.5, mask)  # tile-major rows
        _attn_check(out[:, :, h].cpu(), ref_t[0, 0][rev].unsqueeze(0), f"sta impl head {h} window {w}")
    with pytest.raises(ValueError):
        HipSlidingTileAttentionBackend.get_impl_cls()(num_heads=H, head_size=D, canvas_thw=(21, 30, 52), tile_thw=tile)


# ------------------------------------------------------------------ kernel_api entry points (fastvideo_kernel surface)
def test_kernel_api_block_sparse_attn_bool_map():
    """``block_sparse_attn(q, k, v, block_map_bool, variable_block_sizes) -> (o, lse)`` (fastvideo_kernel/block_sparse_attn.py:384-393)."""
    from fastvideo_amd import kernel_api as KA
    B, H, nq, nk = 1, 2, 6, 6
    q, k, v = (rnd((B, H, nq * 64, 128), s) for s in (1, 2, 3))
    rng = np.random.default_rng(5)
    bm = rng.random((B, H, nq, nk)) < 0.4
    bm[..., 2] = True
    vbs = np.array([64, 17, 64, 64, 1, 40], dtype=np.int32)
    o, lse = KA.block_sparse_attn(q.to(DEV), k.to(DEV), v.to(DEV), torch.from_numpy(bm).to(DEV), torch.from_numpy(vbs).to(DEV))
    _attn_check(o, V.block_sparse_attn(q, k, v, bm, vbs), "kernel_api.block_sparse_attn")
    assert lse.shape == (B, H, nq * 64) and lse.dtype == torch.float32 and torch.isfinite(lse).all()
    with pytest.raises(RuntimeError):
        KA.block_sparse_attn(q.float().to(DEV), k.to(DEV), v.to(DEV), torch.from_numpy(bm).to(DEV), torch.from_numpy(vbs).to(DEV))


def _sta_distribution(shape, seed):
    """fastvideo-kernel/tests/test_sta.py:23-29: unit directions x N(mean 0.1, std 10) magnitudes, bf16 throughout."""
    gen = torch.Generator(device=DEV).manual_seed(seed)
    t = torch.randn(shape, dtype=torch.bfloat16, device=DEV, generator=gen)
    mag = torch.norm(t, dim=-1, keepdim=True)
    return (t * (torch.randn(mag.shape, dtype=torch.bfloat16, device=DEV, generator=gen) * 10 + 0.1) / mag).contiguous()


def test_kernel_api_sliding_tile_attention_on_the_reference_canvas():
    """The reference's pinned STA case (fastvideo-kernel/tests/test_sta.py:17-19, 80-91): ``seq_shape='18x48x80'`` =
    69 120 tokens, 24 heads, its input distribution, its kernel sizes (3,3,5) / (3,1,10) plus Wan's (3,3,3) — through
    ``kernel_api.sliding_tile_attention`` — against the masked fp32 formulation (the semantics of the reference's flex-attention
    mask, support_flex_sta.py:29-59) on sampled query rows.  The reference's own 'TK vs flex' thresholds compare two bf16-P
    flash kernels; against exact fp32 softmax the bound on the mean is the bf16 rounding of P and O (measured ~4e-5 on |O| ~ 0.07)."""
    from fastvideo_amd import kernel_api as KA
    canvas, tile = (18, 48, 80), (6, 8, 8)
    B, H, S, D = 1, 24, 69120, 128
    wins = [(3, 3, 5), (3, 1, 10), (3, 3, 3)] * 8
    q, k, v = (_sta_distribution((B, H, S, D), s) for s in (0, 1, 2))
    o = KA.sliding_tile_attention(q, k, v, wins, 0, False, "18x48x80")
    assert o.shape == q.shape and o.dtype == torch.bfloat16 and torch.isfinite(o.float()).all()
    nt = tuple(c // t for c, t in zip(canvas, tile))
    tile_of = torch.arange(S, device=DEV) // 384
    tt, th, tw = tile_of // (nt[1] * nt[2]), (tile_of // nt[2]) % nt[1], tile_of % nt[2]
    rows = torch.randperm(S, generator=g(7))[:96].sort().values.to(DEV)
    worst_max, tot, cnt = 0.0, 0.0, 0
    for h in (0, 1, 2, 10, 23):
        w = wins[h]
        ok = torch.ones((rows.numel(), S), dtype=torch.bool, device=DEV)
        for qc, kc, n, kk in ((tt[rows], tt, nt[0], w[0]), (th[rows], th, nt[1], w[1]), (tw[rows], tw, nt[2], w[2])):
            centre = qc.clamp(kk // 2, (n - 1) - kk // 2)
            ok &= (centre[:, None] - kc[None, :]).abs() <= kk // 2
        s = (q[0, h, rows].float() @ k[0, h].float().T) * D**-0.5
        ref = torch.softmax(s.masked_fill(~ok, float("-inf")), -1) @ v[0, h].float()
        err = (o[0, h, rows].float() - ref).abs()
        worst_max, tot, cnt = max(worst_max, err.max().item()), tot + err.sum().item(), cnt + err.numel()
    print(f"STA 18x48x80 vs masked fp32: max|err|={worst_max:.4g} mean|err|={tot / cnt:.4g}")
    assert worst_max < 4e-2 and tot / cnt < 2e-4, (worst_max, tot / cnt)
    # has_text=True says rows past the canvas are text: 10 valid text tokens in a sequence that is exactly the canvas is an argument error
    with pytest.raises(ValueError, match="text tokens"):
        KA.sliding_tile_attention(q, k, v, wins, 10, True, "18x48x80")
    with pytest.raises(ValueError):
        KA.sliding_tile_attention(q, k, v, wins[:3], 0, False, "18x48x80")
    # has_text=True with no text rows at all is the image-only mask (support_flex_sta.py:52-55 with text_length 0)
    assert torch.equal(KA.sliding_tile_attention(q[:, :3], k[:, :3], v[:, :3], wins[:3], 0, True, "18x48x80"), o[:, :3])


@pytest.mark.parametrize("text_rows,text_length", [(256, 77), (384, 0), (100, 100)])
def test_kernel_api_sliding_tile_attention_with_text_tokens(text_rows, text_length):
    """The text-token form of ``sliding_tile_attention`` (fastvideo_kernel/ops.py:36-60; HunyuanVideo / StepVideo callers): image
    queries attend their window + the valid text keys, text queries attend every image key + the valid text keys
    (support_flex_sta.py:52-55) — vs exact fp32 attention under the oracle's ``sta_mask`` (the flex-attention mask restated), on a
    4-tile canvas with a per-head window, text rows that do / do not fill whole 64-row blocks and 384-row tiles."""
    from fastvideo_amd import kernel_api as KA
    from oracle import vsa_oracle as V
    from oracle import wan_oracle as W
    canvas, tile = (6, 16, 16), (6, 8, 8)
    img = 6 * 16 * 16
    S, H, D = img + text_rows, 3, 128
    wins = [(1, 1, 1), (1, 3, 1), (1, 1, 3)]
    q, k, v = (_sta_distribution((1, H, S, D), s) for s in (3, 4, 5))
    o = KA.sliding_tile_attention(q, k, v, wins, text_length, True, "6x16x16")
    assert o.shape == q.shape and o.dtype == torch.bfloat16
    qc, kc, vc = q.cpu(), k.cpu(), v.cpu()
    for h, w in enumerate(wins):
        m = V.sta_mask(canvas, w, tile, text_length=text_length, total_len=S)
        ref = W.attention_fp32_ref(qc[:, h:h + 1], kc[:, h:h + 1], vc[:, h:h + 1], D**-0.5, m)
        rows = slice(0, img + text_length)        # padded text QUERY rows attend real keys too (txt2all); compare every row
        err = (o[:, h:h + 1].float().cpu() - ref).abs()
        print(f"STA + text ({text_rows} rows, {text_length} valid), window {w}: max|err|={err.max().item():.4g} mean|err|={err.mean().item():.4g}")
        assert err[:, :, rows].max().item() < 4e-2 and err.mean().item() < 2e-4, (w, err.max().item(), err.mean().item())
        assert err.max().item() < 4e-2


def test_kernel_api_ops_under_torch_compile_on_the_device():
    """The three torch.library ops (fastvideo_kernel::*_gfx950) on the MI355X inside torch.compile(fullgraph=True): one graph, no break at
    the ctypes calls, results bit-identical to the eager calls (the reference registers its kernels the same way so that
    ``enable_torch_compile`` models trace: block_sparse_attn.py:103-145, component_loader.py:1127).  backend="aot_eager": traces through
    AOTAutograd with the fake kernels, runs the real ops, and needs no code generator (the product has no Triton)."""
    from fastvideo_amd import kernel_api as KA
    raw = (8, 16, 16)   # token grid (8,8,8): 2 x 2 x 2 tiles of (4,4,4) = 8 blocks of 64, 512 tokens
    md = KA.build_vsa_metadata((8, 8, 8), device=DEV)
    vbs = md["variable_block_sizes"].int()
    q, k, v, gate = (rnd((1, 2, 512, 128), s).to(DEV) for s in (1, 2, 3, 4))
    wins = [(1, 1, 1), (1, 1, 1)]

    def caller(q, k, v, gate, vbs):
        a = KA.video_sparse_attn(q, k, v, vbs, vbs, 3, (4, 4, 4), compress_attn_weight=gate)
        b = KA.sliding_tile_attention(torch.cat([a, a, a], 2)[:, :, :768], torch.cat([k, k], 2)[:, :, :768], torch.cat([v, v], 2)[:, :, :768], wins, 0, False, "6x8x16")
        idx = torch.arange(8, device=q.device, dtype=torch.int32).expand(1, 2, 8, 8).contiguous()
        num = torch.full((1, 2, 8), 5, device=q.device, dtype=torch.int32)
        o, lse = KA.block_sparse_attn_from_indices(a, k, v, idx, num, vbs)
        return a, b, o, lse

    eager = caller(q, k, v, gate, vbs)
    torch._dynamo.reset()
    compiled = torch.compile(caller, backend="aot_eager", fullgraph=True)(q, k, v, gate, vbs)
    for name, x, y in zip(("video_sparse_attn", "sliding_tile_attention", "block_sparse o", "block_sparse lse"), eager, compiled):
        assert torch.equal(x, y), name
    assert torch.isfinite(eager[1].float()).all()


# ------------------------------------------------------------------ layer ops (fastvideo_amd/layers.py)
def test_hip_rmsnorm_matches_forward_native():
    from fastvideo_amd.layers import HipRMSNorm
    torch.manual_seed(0)
    n = HipRMSNorm(1536, eps=1e-6).to(DEV).to(torch.bfloat16)
    with torch.no_grad():
        n.weight.copy_((1 + 0.1 * torch.randn(1536)).to(torch.bfloat16))
    x = rnd((2, 77, 1536), 1, 2.0).to(DEV)
    got = n(x)  # CustomOp.forward -> forward_cuda
    ref = n.forward_native(x)
    assert got.dtype == torch.bfloat16 and got.shape == x.shape
    _same_rounding(got, ref, "rms_norm")  # same rounding points; only the fp32 sum order of the variance differs
    # a strided view (q columns of a fused QKV buffer)
    buf = rnd((154, 3 * 1536), 2).to(DEV)
    _same_rounding(n(buf[:, 1536:3072]), n.forward_native(buf[:, 1536:3072]), "rms_norm strided")
    with pytest.raises(NotImplementedError):
        n(x, x)
    with pytest.raises(ValueError):
        n(x[..., :512])


def test_hip_rotary_ops_match_reference_formula():
    from fastvideo_amd.layers import HipRotaryEmbedding, apply_rotary_emb
    grid, H, D = (3, 5, 7), 4, 128
    S = math.prod(grid)
    cos, sin = W.rope_tables(grid, D)  # [S, D] fp32 full-width tables, what the reference passes (wanvideo.py:679-687)
    x = rnd((2, S, H, D), 3)
    ref = W.apply_rotary_emb(x, cos, sin)
    got = apply_rotary_emb(x.to(DEV), cos.to(DEV), sin.to(DEV), is_neox_style=False)
    _same_rounding(got, ref, "apply_rotary_emb (full-width tables)")
    half = apply_rotary_emb(x.to(DEV), cos[:, ::2].contiguous().to(DEV), sin[:, ::2].contiguous().to(DEV), is_neox_style=False)
    assert torch.equal(half, got)  # [S, D/2] GPT-J tables = the same rotation
    with pytest.raises(NotImplementedError):
        apply_rotary_emb(x.to(DEV), cos[:, ::2].contiguous().to(DEV), sin[:, ::2].contiguous().to(DEV), is_neox_style=True)
    rot = HipRotaryEmbedding(head_size=D, rotary_dim=D, max_position_embeddings=64, base=10000, is_neox_style=False, dtype=torch.float32).to(DEV)
    pos = torch.tensor([5, 0, 63, 7, 7, 31], device=DEV)
    qq, kk = rnd((6, H * D), 4).to(DEV), rnd((6, 2 * D), 5).to(DEV)
    gq, gk = rot(pos, qq, kk)
    rq, rk = rot.forward_native(pos, qq, kk)
    _same_rounding(gq, rq, "rotary_embedding q"), _same_rounding(gk, rk, "rotary_embedding k")


def test_hip_linear_methods_match_eager_linear():
    """``QuantizeMethodBase.apply`` classes on a minimal layer object: bf16 GEMM vs ``F.linear`` (atol=rtol=1e-2), fp8 method
    vs the oracle's fp8 linear (bytes/scales bit-exact, GEMM to bf16 tolerance)."""
    import torch.nn as nn
    from fastvideo_amd import layers as L
    from oracle import fp8_oracle as F8

    class Layer(nn.Module):
        pass

    for method, prefix in ((L.HipLinearMethod(), "x"), (L.HipFP8LinearMethod("tensor"), "ffn.fc_in"), (L.HipFP8LinearMethod("channel"), "to_q")):
        lay = Layer()
        method.create_weights(lay, 1536, [512, 256], 1536, 768, torch.bfloat16)
        assert lay.weight.shape == (768, 1536) and not lay.weight.requires_grad
        lay.quant_method = method
        w, b = rnd((768, 1536), 1, 0.05), rnd((768, ), 2, 0.1)
        lay.weight.data.copy_(w)
        lay = lay.to(DEV)
        x = rnd((2, 130, 1536), 3).to(DEV)
        if isinstance(method, L.HipFP8LinearMethod):
            L.convert_model_to_fp8(lay)
            assert not hasattr(lay, "weight") or "weight" not in lay._parameters
            gran = method.granularity
            wq, ws = F8.quantize_weight(w, gran)
            assert torch.equal(lay._fp8_weight.cpu().view(torch.uint8), wq.view(torch.uint8))
            assert torch.equal(lay._fp8_weight_scale.cpu().view(-1), ws.view(-1))
            pre = method.quantize_input(x)
            ref = F8.fp8_linear(x.cpu(), wq, ws, b, gran)
            for out in (method.apply(lay, x, b.to(DEV)), method.apply(lay, x, b.to(DEV), pre_quantized=pre)):
                assert out.shape == (2, 130, 768)
                err = (out.float().cpu() - ref.float()).abs()
                assert (err <= 2e-2 + 2e-2 * ref.float().abs()).all(), err.max().item()
        else:
            out = method.apply(lay, x, b.to(DEV))
            ref = torch.nn.functional.linear(x.cpu().float(), w.float(), b.float())
            err = (out.float().cpu() - ref).abs()
            assert out.shape == (2, 130, 768) and (err <= 1e-2 + 1e-2 * ref.abs()).all(), err.max().item()
    cfg = L.Mi355xFp8Config("channel")
    assert cfg.get_name() == "MI355X_FP8" and L.Mi355xBf16Config().get_name() == "MI355X_BF16"


# ------------------------------------------------------------------ sequence-parallel exchange packing (fvk_qkv_norm_rope_pack_bf16)
@pytest.mark.parametrize("H,G,U", [(12, 4, 2), (12, 4, 1), (12, 2, 1), (40, 8, 1), (2, 2, 2)])
def test_qkv_norm_rope_pack_equals_norm_rope_then_torch_pack(H, G, U):
    """The fused kernel = ``rmsnorm_rope`` on q, k (+ v untouched) followed by the layout ``SequenceParallel.pack_rows`` builds with
    plain torch — bit for bit (same arithmetic, only the store addresses differ), from strided column blocks of a fused QKV buffer,
    with a shard position offset (RoPE uses GLOBAL token positions)."""
    from fastvideo_amd import ops
    from fastvideo_amd.distributed import SequenceParallel, SPLayout
    D, Sl, S, pos0 = 128, 37, 150, 74
    d = H * D
    qkv = rnd((Sl, 3 * d), 1, 1.5).to(DEV)
    wq, wk = (1 + 0.1 * torch.randn(d, generator=g(2))).bfloat16().to(DEV), (1 + 0.1 * torch.randn(d, generator=g(3))).bfloat16().to(DEV)
    cos, sin = W.rope_tables((5, 5, 6), D)
    cos, sin = cos.to(DEV), sin.to(DEV)
    q, k, v = qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:]
    send = ops.qkv_norm_rope_pack(q, k, v, wq, wk, cos, sin, G, U, head_dim=D, seq_len=S, eps=1e-6, pos_offset=pos0)
    qn, kn = ops.rmsnorm_rope([q, k], [wq, wk], cos, sin, head_dim=D, seq_len=S, eps=1e-6, pos_offset=pos0)
    sp = SequenceParallel(H)
    sp.lay = SPLayout(P=G * U, rank=0, H=H, G=G, U=U)
    want = sp.pack_rows(qn.view(Sl, H, D), kn.view(Sl, H, D), v.reshape(Sl, H, D))
    assert send.shape == want.shape == (G * U, Sl, 3, d // G)
    assert torch.equal(send, want)
    with pytest.raises(RuntimeError):
        ops.qkv_norm_rope_pack(q, k, v, wq, wk, cos, sin, 7, 1, head_dim=D, seq_len=S)  # 12 / 40 / 2 heads do not split 7 ways


@pytest.mark.parametrize("block", [(4, 4, 8), (4, 8, 8)])
def test_video_sparse_attn_128_and_256_token_blocks(block):
    """``video_sparse_attn`` / ``video_sparse_attn_bshd`` with 128- and 256-token blocks (fastvideo_kernel/ops.py:78-81, 125-128, 136-200:
    the reference's Blackwell CuTe paths) — the 128-row list kernel over lists expanded to 64-key sub-blocks — vs the oracle composite
    with the device's block selection (unconditional, as for the 64-token block)."""
    from fastvideo_amd import kernel_api as KA
    be = math.prod(block)
    B, H, D, nb = 1, 2, 128, 9
    vbs = np.array([be, be, be - 37, be, 1, be // 2 + 3, be, 70, be], dtype=np.int32)
    S = nb * be
    q, k, v, gate = (rnd((B, H, S, D), s_) for s_ in (11, 12, 13, 14))
    for t in (q, k, v, gate):
        for b in range(nb):
            t[:, :, b * be + int(vbs[b]):(b + 1) * be] = 0  # a block's real tokens first, zero padding after
    topk = 4
    tv = torch.from_numpy(vbs).to(DEV)
    out, inter = KA.video_sparse_attn(q.to(DEV), k.to(DEV), v.to(DEV), tv, tv, topk, block, gate.to(DEV), return_intermediates=True)
    mask = inter["mask"].cpu().numpy()
    assert np.array_equal(mask, V.topk_mask_bisect(inter["scores"].float().cpu().numpy(), topk)) and (mask.sum(-1) == topk).all()
    ref, _ = V.video_sparse_attn(q, k, v, vbs, vbs, topk, be, gate, mask_override=mask)
    rows = torch.cat([torch.arange(b * be, b * be + int(vbs[b])) for b in range(nb)])  # pad query rows are dropped by untile
    _attn_check(out[:, :, rows], ref.float()[:, :, rows], f"vsa composite, {be}-token blocks", mean_tol=5e-3)
    o2 = KA.video_sparse_attn_bshd(q.transpose(1, 2).contiguous().to(DEV), k.transpose(1, 2).contiguous().to(DEV),
                                   v.transpose(1, 2).contiguous().to(DEV), tv, tv, topk, block, gate.transpose(1, 2).contiguous().to(DEV))
    assert torch.equal(o2.transpose(1, 2), out)
    with pytest.raises(ValueError):
        KA.video_sparse_attn(q.to(DEV), k.to(DEV), v.to(DEV), tv, tv, topk, (4, 4, 2), gate.to(DEV))  # 32-token blocks do not exist
