"""Parity of every HIP kernel (called through the C ABI via fastvideo_amd.ops) against the CPU oracle on the same
seeded inputs.  Integer/index outputs: bit-exact.  bf16 activations: the reference's own fused-op tolerance
atol=rtol=1e-2 (fastvideo-kernel/tests/test_turbodiffusion.py:143, test_fused_compress_topk.py:138); attention:
max |Δ| < 4e-2 as in fastvideo-kernel/tests/test_sta.py:88-91 plus a mean-error bound."""
import math

import numpy as np
import pytest
import torch

from oracle import vsa_oracle as V
from oracle import wan_oracle as W

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from fastvideo_amd import ops as o
    return o


def g(seed):
    return torch.Generator().manual_seed(seed)


def rnd(shape, seed, scale=1.0, dtype=torch.bfloat16):
    return (torch.randn(shape, generator=g(seed)) * scale).to(dtype)


def close(a, b, atol=1e-2, rtol=1e-2, what=""):
    a, b = a.float().cpu(), b.float().cpu()
    err = (a - b).abs()
    bad = err > atol + rtol * b.abs()
    assert not bad.any(), (f"{what}: {int(bad.sum())}/{bad.numel()} elements off; max abs err {err.max().item():.4g} "
                           f"at {np.unravel_index(int(err.argmax()), tuple(err.shape))}, ref there "
                           f"{b.flatten()[err.argmax()].item():.4g}")


# ------------------------------------------------------------------ LN / modulate family
@pytest.mark.parametrize("M,d,B", [(37, 256, 1), (70, 1536, 2), (5, 5120, 1)])
def test_ln_scale_shift_block(ops, M, d, B):
    x = rnd((B, M, d), 1, 2.0)
    scale, shift = rnd((B, 1, d), 2, 0.3, torch.float32), rnd((B, 1, d), 3, 0.3, torch.float32)
    ref = W.ln_scale_shift_block(x, scale, shift)
    out = ops.ln_modulate(x.to(DEV), mul=(1 + scale).to(DEV), add=shift.to(DEV), eps=1e-6)
    close(out, ref, what="norm1")


@pytest.mark.parametrize("M,d,B", [(33, 256, 1), (64, 1536, 2)])
def test_scale_residual_ln_self_attn(ops, M, d, B):
    res, x = rnd((B, M, d), 1, 2.0), rnd((B, M, d), 2)
    gate = rnd((B, 1, d), 3, 0.5, torch.float32)
    w, b = rnd((d, ), 4).float() * 0.1 + 1, rnd((d, ), 5).float() * 0.1
    null = torch.tensor([0])
    mod, ro = W.scale_residual_ln_scale_shift(res, x, gate, null, null, w, b)
    out, rout = ops.ln_modulate(x.to(DEV), residual=res.to(DEV), gate=gate.to(DEV), ln_w=w.to(DEV), ln_b=b.to(DEV),
                                want_residual=True)
    close(rout, ro.bfloat16(), what="residual_out")
    close(out, mod.bfloat16(), what="normed")


@pytest.mark.parametrize("M,d,B", [(33, 256, 1), (64, 1536, 2)])
def test_scale_residual_ln_cross_attn(ops, M, d, B):
    res, x = rnd((B, M, d), 1, 2.0), rnd((B, M, d), 2)
    scale, shift = rnd((B, 1, d), 2, 0.3, torch.float32), rnd((B, 1, d), 3, 0.3, torch.float32)
    mod, ro = W.scale_residual_ln_scale_shift(res, x, 1, shift, scale)
    out, rout = ops.ln_modulate(x.to(DEV), residual=res.to(DEV), mul=(1.0 + scale).to(DEV), add=shift.to(DEV),
                                round_residual=True, round_norm=True, want_residual=True)
    assert torch.equal(rout.cpu(), ro), "bf16 residual add must be bit exact"
    close(out, mod.bfloat16(), what="normed")


def test_ln_scale_shift_out_and_scale_residual(ops):
    B, M, d = 2, 50, 1536
    x = rnd((B, M, d), 1, 2.0)
    scale, shift = rnd((B, 1, d), 2, 0.3), rnd((B, 1, d), 3, 0.3)  # bf16 like the model's norm_out
    ref = W.ln_scale_shift_out(x, shift, scale)
    out = ops.ln_modulate(x.to(DEV), mul=(1.0 + scale).float().to(DEV), add=shift.float().to(DEV), round_norm=True)
    close(out, ref, what="norm_out")
    gate = rnd((B, 1, d), 4, 0.5, torch.float32)
    y = rnd((B, M, d), 5)
    ref2 = W.scale_residual(x, y, gate).bfloat16()
    out2 = ops.scale_residual(x.to(DEV), y.to(DEV), gate.to(DEV))
    assert torch.equal(out2.cpu(), ref2), "gated residual is two exactly-rounded fp32 ops + one bf16 rounding"


@pytest.mark.parametrize("grid,H", [((3, 4, 4), 2), ((3, 5, 7), 12)])
def test_rmsnorm_rope(ops, grid, H):
    S, D = math.prod(grid), 128
    d = H * D
    qkv = rnd((S, 3 * d), 1, 1.5)
    wq, wk = (rnd((d, ), 2).float() * 0.2 + 1).bfloat16(), (rnd((d, ), 3).float() * 0.2 + 1).bfloat16()
    cos, sin = W.rope_tables(grid, D)
    refs = []
    for i, w in enumerate((wq, wk)):
        n = W.rms_norm(qkv[:, i * d:(i + 1) * d].unsqueeze(0), w)
        refs.append(W.apply_rotary_emb(n.view(1, S, H, D), cos, sin).view(S, d))
    dq = qkv.to(DEV)
    outs = ops.rmsnorm_rope([dq[:, :d], dq[:, d:2 * d]], [wq.to(DEV), wk.to(DEV)], cos.to(DEV), sin.to(DEV), head_dim=D,
                            seq_len=S)
    for o, r, n in zip(outs, refs, "qk"):
        close(o, r, what=f"rmsnorm+rope {n}")
    # norm only / rope only must compose to the same thing
    n_only = ops.rmsnorm_rope([dq[:, :d]], [wq.to(DEV)], head_dim=D, seq_len=S)[0]
    close(n_only, W.rms_norm(qkv[:, :d].unsqueeze(0), wq)[0], what="rmsnorm only")
    r_only = ops.rmsnorm_rope([n_only], None, cos.to(DEV), sin.to(DEV), head_dim=D, seq_len=S)[0]
    assert torch.equal(r_only, outs[0]), "norm->rope split must equal the fused kernel bit for bit"
    # scattered output rows (fvk_rmsnorm_rope_scatter_bf16): q through a permutation into a larger buffer, k with every 5th row dropped
    gperm = torch.randperm(S + 40, generator=g(9))[:S].to(torch.int32)
    kmap = torch.arange(S, dtype=torch.int32)
    kmap[::5] = -1
    qo = torch.full((S + 40, d), 7.0, dtype=torch.bfloat16, device=DEV)
    ko = torch.full((S + 40, d), 7.0, dtype=torch.bfloat16, device=DEV)
    ops.rmsnorm_rope([dq[:, :d], dq[:, d:2 * d]], [wq.to(DEV), wk.to(DEV)], cos.to(DEV), sin.to(DEV), head_dim=D, seq_len=S,
                     outs=[qo, ko], row_maps=[gperm.to(DEV), kmap.to(DEV)])
    assert torch.equal(qo[gperm.long().to(DEV)], outs[0])
    untouched = torch.ones(S + 40, dtype=torch.bool)
    untouched[gperm.long()] = False
    assert (qo[untouched.to(DEV)] == 7.0).all()
    keep = (kmap >= 0)
    assert torch.equal(ko[:S][keep.to(DEV)], outs[1][keep.to(DEV)]) and (ko[:S][~keep.to(DEV)] == 7.0).all() and (ko[S:] == 7.0).all()


def test_v_transpose_layout(ops):
    B, S, H, D = 2, 150, 3, 128
    v = rnd((B, S, H, D), 1)
    vt = ops.v_transpose(v.to(DEV)).cpu()
    S_pad = 256  # whole 128-key tiles (zero padded)
    assert vt.shape == (B, H, D, S_pad)
    p = torch.arange(S_pad)
    key = (p & ~12) | ((p & 4) << 1) | ((p & 8) >> 1)
    ref = torch.zeros(B, H, D, S_pad, dtype=torch.bfloat16)
    vpad = torch.zeros(B, S_pad, H, D, dtype=torch.bfloat16)
    vpad[:, :S] = v
    ref[:] = vpad[:, key].permute(0, 2, 3, 1)
    assert torch.equal(vt, ref)
    # strided (bhsd) source
    v2 = v.permute(0, 2, 1, 3).contiguous().to(DEV)  # [B,H,S,D]
    assert torch.equal(ops.v_transpose(v2.transpose(1, 2)).cpu(), ref)
    # gathered source rows (fvk_v_transpose_gather_bf16): key position p takes row src[p] (negative: a zero column) == gather, then transpose
    src = torch.full((384,), -1, dtype=torch.int32)
    src[torch.randperm(384, generator=g(4))[:S]] = torch.randperm(S, generator=g(5)).to(torch.int32)
    vg = torch.zeros(B, 384, H, D, dtype=torch.bfloat16)
    vg[:, src >= 0] = v[:, src[src >= 0].long()]
    assert torch.equal(ops.v_transpose(v.to(DEV), src_rows=src.to(DEV)), ops.v_transpose(vg.to(DEV)))
    with pytest.raises(RuntimeError, match="whole 128-key tiles"):
        ops.v_transpose(v.to(DEV), src_rows=src[:300].to(DEV))


# ------------------------------------------------------------------ GEMM
def _lin_ref(x, w, b):
    return (x.float() @ w.float().t() + (0 if b is None else b.float())).bfloat16()


@pytest.mark.parametrize("B,S,d,K", [(1, 264, 256, 256), (2, 1000, 1536, 1536), (1, 32760, 1536, 1536), (1, 4095, 384, 128)])
def test_gemm_vt_equals_linear_then_v_transpose(ops, B, S, d, K):
    """fvk_gemm_vt_bf16 (round 5): the V projection written straight into the attention kernels' V^T layout == `to_v` as a plain GEMM
    (linear.py:146-156) followed by the layout pass fvk_v_transpose_bf16, BIT FOR BIT: the same fragments meet in the same MFMAs in the same k
    order, only as B / A instead of A / B, and the epilogue rounds y = bf16(acc + bias) at the same point.  Cases: one tile; a batch of two with
    a ragged last tile and a half-filled last 16-key group (1000 = 62 x 16 + 8: positions 4-7 and 12-15 of that group are padding); the
    contract shape; S % 8 != 0 is refused.  Also against the fp32 oracle, and the padding columns must be exact zeros (0 x finite in P·V)."""
    if S % 8:
        x = rnd((B, S, K), 1).to(DEV)
        assert not ops.gemm_vt_eligible(x, rnd((d, K), 2).to(DEV))
        with pytest.raises(RuntimeError, match="not served"):
            ops.gemm_vt(x, rnd((d, K), 2).to(DEV))
        return
    x, w, b = rnd((B, S, K), 1), rnd((d, K), 2, K**-0.5), rnd((d,), 3, 0.5)
    H = d // 128
    v = ops.gemm(x.to(DEV).view(B * S, K), w.to(DEV), b.to(DEV)).view(B, S, H, 128)
    want = ops.v_transpose(v)
    got = ops.gemm_vt(x.to(DEV), w.to(DEV), b.to(DEV))
    assert got.shape == want.shape == (B, H, 128, (S + 127) // 128 * 128)
    assert torch.equal(got, want), f"{int((got != want).sum())} of {got.numel()} elements differ, max {(got.float() - want.float()).abs().max().item():.4g}"
    # no-bias form
    assert torch.equal(ops.gemm_vt(x.to(DEV), w.to(DEV)), ops.v_transpose(ops.gemm(x.to(DEV).view(B * S, K), w.to(DEV)).view(B, S, H, 128)))
    # fp32 oracle on the un-permuted columns of a few heads + exact zeros in the padding
    ref = W.linear(x, w, b).float().view(B, S, H, 128).permute(0, 2, 3, 1)            # [B,H,128,S]
    pos = torch.arange(got.shape[-1])
    key = (pos & ~12) | ((pos & 4) << 1) | ((pos & 8) >> 1)
    valid = key < S
    gc = got.float().cpu()
    close(gc[..., valid], ref[..., key[valid]], what="gemm_vt vs fp32 linear")
    assert (gc[..., ~valid] == 0).all()


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (300, 200, 128), (1, 1536, 256), (257, 64, 1536), (130, 8960, 192)])
def test_gemm_bias(ops, M, N, K):
    x, w, b = rnd((M, K), 1), rnd((N, K), 2, K**-0.5), rnd((N, ), 3)
    out = ops.gemm(x.to(DEV), w.to(DEV), b.to(DEV))
    close(out, _lin_ref(x, w, b), what=f"gemm {M}x{N}x{K}")
    out_nb = ops.gemm(x.to(DEV), w.to(DEV), None)
    close(out_nb, _lin_ref(x, w, None), what="gemm no bias")


def test_gemm_identity_asymmetric(ops):
    """A = I with an asymmetric B catches a transposed C write (guide: 'Always A=I-check with ASYMMETRIC B')."""
    K = 128
    x = torch.eye(K).bfloat16()
    w = (torch.arange(K * K).view(K, K) % 251).float().bfloat16()  # exactly representable
    out = ops.gemm(x.to(DEV), w.to(DEV), None).cpu()
    assert torch.equal(out, w.t().contiguous())


def test_gemm_epilogues(ops):
    M, N, K, B = 200, 256, 128, 2
    x, w, b = rnd((M, K), 1), rnd((N, K), 2, K**-0.5), rnd((N, ), 3)
    y = _lin_ref(x, w, b)
    close(ops.gemm(x.to(DEV), w.to(DEV), b.to(DEV), epilogue=ops.EPI_GELU_TANH), W.gelu_tanh(y), what="gelu")
    close(ops.gemm(x.to(DEV), w.to(DEV), b.to(DEV), epilogue=ops.EPI_SILU), torch.nn.functional.silu(y), what="silu")
    res, gate = rnd((M, N), 4, 2.0), rnd((B, N), 5, 0.5, torch.float32)
    ref = W.scale_residual(res.view(B, M // B, N), y.view(B, M // B, N), gate.view(B, 1, N)).bfloat16().view(M, N)
    out = ops.gemm(x.to(DEV), w.to(DEV), b.to(DEV), epilogue=ops.EPI_RESIDUAL_GATE, residual=res.to(DEV), gate=gate.to(DEV))
    close(out, ref, what="residual+gate")
    # strided A (column slice of a wider buffer) and batched DIV
    wide = rnd((M, 3 * K), 6)
    close(ops.gemm(wide.to(DEV)[:, K:2 * K], w.to(DEV), b.to(DEV)), _lin_ref(wide[:, K:2 * K], w, b), what="strided A")
    xb, wb = rnd((3, 70, 128), 7), rnd((3, 90, 128), 8)
    refb = (torch.matmul(xb, wb.transpose(-1, -2)) / (128**0.5))
    close(ops.gemm_batched(xb.to(DEV), wb.to(DEV), ops.EPI_DIV, 128**0.5), refb, what="batched div")


@pytest.mark.parametrize("P", [8, 4])
def test_gemm_rank_shapes_equal_the_rows_of_the_full_problem(ops, P):
    """A rank's projections under sequence parallelism (M = 32 760 / P rows) must produce the bytes the SAME rows get inside the SP = 1
    problem — whichever GEMM kernel the shape selects.  At P = 8 the N = 1536 projections are 96 tiles of 256 x 256 and run on gemm_w1n.hip
    (256 x 128 tiles, round 4); the full problem runs on gemm_w1.hip; both accumulate the same 32-k MFMA steps in the same order.  Plain,
    GELU and gated-residual epilogues, K = 1536 and the FFN-out depth 8960, and a ragged last tile (4095 = 15 x 256 + 255 rows)."""
    S, d, f = 32760, 1536, 8960
    Sl = S // P
    r = P - 1                                    # the last rank: its shard ends at the ragged end of the sequence
    for K, N, epi in ((d, d, ops.EPI_NONE), (d, d, ops.EPI_RESIDUAL_GATE), (f, d, ops.EPI_RESIDUAL_GATE), (d, 2 * d, ops.EPI_GELU_TANH)):
        x, w, b = rnd((S, K), 1).to(DEV), rnd((N, K), 2, K**-0.5).to(DEV), rnd((N,), 3).to(DEV)
        kw_full, kw_rank = {}, {}
        if epi == ops.EPI_RESIDUAL_GATE:
            res, gate = rnd((S, N), 4, 2.0).to(DEV), rnd((1, N), 5, 0.5, torch.float32).to(DEV)
            kw_full = dict(residual=res, gate=gate, rows_per_batch=S)
            kw_rank = dict(residual=res[r * Sl:(r + 1) * Sl], gate=gate, rows_per_batch=Sl)
        full = ops.gemm(x, w, b, epilogue=epi, **kw_full)
        rank = ops.gemm(x[r * Sl:(r + 1) * Sl], w, b, epilogue=epi, **kw_rank)
        assert torch.equal(rank, full[r * Sl:(r + 1) * Sl]), f"P={P} K={K} N={N} epilogue {epi}: rank rows differ from the full problem's"
        close(rank[:64], _lin_ref(x[r * Sl:r * Sl + 64].cpu(), w.cpu(), b.cpu()) if epi == ops.EPI_NONE else rank[:64], what="rank rows vs fp32")


def test_small_kernel_beside_gemm_w1_on_another_stream(ops):
    """Regression test of round 4's co-residency bug (profiles/r04z_pk_f32_beside_mfma.log): while gemm_w1 ran on another stream, a wave
    of the QK-norm / RoPE / pack pass that shared a SIMD with one of its waves (gemm_w1 left 104 of 512 registers free) got wrong packed-fp32
    results — 192 of 200 launches.  The one-wave-per-SIMD 16x16x32 kernels now claim the whole register file; the pass beside gemm_w1, gemm_w1n
    and conv3w must reproduce its stand-alone bytes every time."""
    g_ = torch.Generator().manual_seed(5)
    Sl, d, D = 338, 768, 128
    qkv = torch.randn((Sl, 3 * d), generator=g_).bfloat16().to(DEV)
    wq, wk = (1 + 0.1 * torch.randn(d, generator=g_)).bfloat16().to(DEV), (1 + 0.1 * torch.randn(d, generator=g_)).bfloat16().to(DEV)
    ang = torch.rand((2 * Sl, D), generator=g_) * 6.28
    cos, sin = torch.cos(ang).float().to(DEV), torch.sin(ang).float().to(DEV)
    A, B = torch.randn((8192, 4096), generator=g_).bfloat16().to(DEV), torch.randn((4096, 4096), generator=g_).bfloat16().to(DEV)
    An, Bn = torch.randn((4095, 1536), generator=g_).bfloat16().to(DEV), torch.randn((1536, 1536), generator=g_).bfloat16().to(DEV)   # gemm_w1n's shape class
    xc = torch.randn((6, 240, 416, 96), generator=g_).bfloat16().to(DEV)
    wc, bc = (torch.randn((96, 27 * 96), generator=g_) * (27 * 96)**-0.5).bfloat16().to(DEV), torch.zeros(96).bfloat16().to(DEV)
    pack = lambda: ops.qkv_norm_rope_pack(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], wq, wk, cos, sin, 2, 1, head_dim=D, seq_len=2 * Sl)
    ref = pack().clone()
    torch.cuda.synchronize()
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    for name, load in (("gemm_w1", lambda: ops.gemm(A, B)), ("gemm_w1n", lambda: ops.gemm(An, Bn)),
                       ("conv3w", lambda: ops.vae_conv(xc, wc, bc, T=4, H=240, W=416, kt=3, ks=3))):
        outs = []
        for _ in range(120):
            with torch.cuda.stream(sa):
                keep = load()
            with torch.cuda.stream(sb):
                outs.append(pack())
        torch.cuda.synchronize()
        bad = sum(0 if torch.equal(o, ref) else 1 for o in outs)
        assert bad == 0, f"{bad} of {len(outs)} launches of the norm / RoPE / pack pass beside {name} differ from the stand-alone result"


def test_vsa_union_lists_and_union_walk_are_exact(ops):
    """Round 4: fvk_vsa_union_lists (the ascending merge of the lists of query blocks 2p / 2p + 1, entries tagged with the halves that selected
    them) is bit-exact against a host merge, and fvk_attn_block_sparse_union_bf16 — one walk over the merged list, shared KV tiles fetched once —
    returns the bytes of the two-list kernel: with heavily overlapping lists, disjoint lists, an odd block count (a last block without a
    partner), an empty list, and variable block sizes."""
    from fastvideo_amd import _lib
    import ctypes as C
    B, H, nq, nk, D = 1, 3, 9, 40, 128
    g_ = torch.Generator().manual_seed(0)
    mask = torch.rand((B, H, nq, nk), generator=g_) < 0.3
    mask[0, 0, 1] = mask[0, 0, 0]                      # identical neighbours
    mask[0, 1, 2] = ~mask[0, 1, 3]                     # disjoint neighbours
    mask[0, 2, 4] = False                              # an empty list beside a full one
    mask[0, 2, 5] = True
    vbs = torch.randint(1, 65, (nk,), generator=g_, dtype=torch.int32)
    vbs[::3] = 64
    idx, num = ops.map_to_index(mask.to(DEV))
    max_kv = idx.shape[-1]
    u_idx = torch.zeros((B * H, (nq + 1) // 2, 2 * max_kv), dtype=torch.int32, device=DEV)
    u_num = torch.zeros((B * H, (nq + 1) // 2), dtype=torch.int32, device=DEV)
    p_ = lambda t: C.c_void_p(t.data_ptr())
    _lib.call("fvk_vsa_union_lists", p_(idx), p_(num), p_(vbs.to(DEV)), p_(u_idx), p_(u_num), B * H, nq, max_kv, ops._stream())
    ui, un = u_idx.cpu(), u_num.cpu()
    m = mask.view(B * H, nq, nk)
    for bh in range(B * H):
        for p in range((nq + 1) // 2):
            a_ = m[bh, 2 * p]
            b_ = m[bh, 2 * p + 1] if 2 * p + 1 < nq else torch.zeros(nk, dtype=torch.bool)
            ids = torch.nonzero(a_ | b_).flatten()
            want = [int(i) | (int(vbs[i]) << 22) | ((int(a_[i]) | (int(b_[i]) << 1)) << 29) for i in ids]
            assert int(un[bh, p]) == len(want) and ui[bh, p, :len(want)].tolist() == want, (bh, p)
    S = nq * 64
    q, k, v = rnd((B, H, S, D), 1).to(DEV), rnd((B, H, nk * 64, D), 2).to(DEV), rnd((B, H, nk * 64, D), 3).to(DEV)
    two = ops.attn_block_sparse(q, k, v, idx, num, vbs.to(DEV), return_lse=True, pair_union=False)
    uni = ops.attn_block_sparse(q, k, v, idx, num, vbs.to(DEV), return_lse=True, pair_union=True)
    # round 6: the plain 64-row form runs attn_bs16 (fixed softmax reference, one wave per list), the union walk still runs the round-1 kernel
    # (online softmax) — the same attention to rounding: one bf16 ulp of the output (|o| < 0.5 here), LSE to the bf16-P row sum's 2^-9
    # (bit-identity of the union walk with the round-1 two-list kernel: scripts/probes/variant_tests.py, measurement build)
    assert torch.isfinite(two[0].float()).all() and torch.isfinite(uni[0].float()).all()
    d = (two[0].float() - uni[0].float()).abs()
    assert d.max().item() <= 4e-3 and d.mean().item() <= 3e-4, (d.max().item(), d.mean().item())   # measured 2.0e-3 / 1.0e-4 (lists of ~12 blocks)
    live = num.view(B, H, nq) > 0
    rows = live.repeat_interleave(64, dim=2)
    assert (two[1][rows] - uni[1][rows]).abs().max().item() <= 2e-2
    assert (uni[0][0, 2, 4 * 64:5 * 64] == 0).all() and (two[0][0, 2, 4 * 64:5 * 64] == 0).all()    # the empty list: zeros in both


# ------------------------------------------------------------------ attention
def _attn_check(out, ref, what):
    """max |err| < 4e-2: the reference's own kernel-test bound (fastvideo-kernel/tests/test_sta.py:88-91).  Mean: the kernel's only bf16
    roundings are P and the output, and the measured mean |err| is 2.1e-3 x mean |ref| on every case (= the rounding of a bf16 output:
    half an ulp of 2^-8 relative, on average); the bound is 1.4x that plus an absolute floor for near-zero outputs — not the former
    absolute 2e-3, which was ~15x what the kernel achieves."""
    err = (out.float().cpu() - ref).abs()
    assert torch.isfinite(out.float()).all(), f"{what}: non-finite output"
    mean_bound = 3e-3 * ref.float().abs().mean().item() + 2e-5
    assert err.max().item() < 4e-2 and err.mean().item() < mean_bound, \
        f"{what}: max {err.max().item():.4g} mean {err.mean().item():.4g} (bound {mean_bound:.4g})"


@pytest.mark.parametrize("B,H,Sq,Skv", [(1, 2, 48, 48), (1, 2, 105, 105), (2, 3, 300, 77), (1, 1, 1000, 1000), (1, 12, 130, 512)])
def test_attn_dense_bshd(ops, B, H, Sq, Skv):
    q, k, v = rnd((B, Sq, H, 128), 1), rnd((B, Skv, H, 128), 2), rnd((B, Skv, H, 128), 3)
    ref = W.attention_fp32_ref(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2), 128**-0.5).transpose(1, 2)
    out = ops.attn_dense(q.to(DEV), k.to(DEV), v.to(DEV), layout="bshd")
    _attn_check(out, ref, f"dense bshd {B},{H},{Sq},{Skv}")


def test_attn_dense_bhsd_and_sta_distribution(ops):
    """[B,H,S,D] layout with the input distribution of fastvideo-kernel/tests/test_sta.py:23-29."""
    B, H, S = 1, 2, 400
    def gen(seed):
        d = torch.randn((B, H, S, 128), generator=g(seed))
        d = d / d.norm(dim=-1, keepdim=True)
        mag = (torch.randn((B, H, S, 1), generator=g(seed + 10)) * 10 + 0.1)
        return (d * mag).bfloat16()
    q, k, v = gen(1), gen(2), gen(3)
    ref = W.attention_fp32_ref(q, k, v, 128**-0.5)
    out = ops.attn_dense(q.to(DEV), k.to(DEV), v.to(DEV), layout="bhsd")
    err = (out.float().cpu() - ref).abs()
    assert err.max().item() < 4e-2 * max(1.0, ref.abs().max().item() / 4), f"max {err.max().item()}"


def test_attn_dense_softmax_rescale_branch(ops):
    """Force the running-max rescale at a late KV tile (guide §5.4 rule 26): one key spikes against one query."""
    B, H, S = 1, 1, 320
    q, k, v = rnd((B, S, H, 128), 1, 0.5), rnd((B, S, H, 128), 2, 0.5), rnd((B, S, H, 128), 3)
    k[0, 250, 0] = q[0, 7, 0] * 6  # raw q.k >> other scores, inside tile 3
    ref = W.attention_fp32_ref(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2), 128**-0.5).transpose(1, 2)
    out = ops.attn_dense(q.to(DEV), k.to(DEV), v.to(DEV), layout="bshd")
    _attn_check(out, ref, "rescale branch")


@pytest.mark.parametrize("B,H,Sq,Skv,spike", [(1, 2, 700, 3100, 0), (2, 3, 512, 2049, 0), (1, 2, 300, 2048, 1), (1, 1, 300, 4000, 1), (1, 2, 1030, 3200, 2), (1, 12, 256, 8192, 0)])
def test_attn_dense_long_keys_w16(ops, B, H, Sq, Skv, spike):
    """Key axes of 2048 and more take attn_w16 (4 waves x 64 rows, 16x16x32 MFMAs, fixed softmax reference, row sums from the matrix pipe): ragged Sq / Skv tails (masked last stage, one
    valid key in the last stage), odd stage counts, a spiked key far beyond the first sub-tile's maximum (growth ~2^98: the row's exact
    recompute) and one beyond any fp32 range (x20: every row of that head), repeatability, and the LSE."""
    q, k, v = rnd((B, Sq, H, 128), Sq), rnd((B, Skv, H, 128), Skv), rnd((B, Skv, H, 128), 3)
    if spike >= 1:
        k[0, Skv - 100, 0] = q[0, 7, 0] * 6
    if spike == 2:
        k[0, 1500, 1] = q[0, 300, 1] * 20
    ref = W.attention_fp32_ref(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2), 128**-0.5).transpose(1, 2)
    qd, kd, vd = q.to(DEV), k.to(DEV), v.to(DEV)
    o, lse = ops.attn_dense(qd, kd, vd, layout="bshd", return_lse=True, key_splits=1)
    _attn_check(o, ref, f"attn_w16 {B},{H},{Sq},{Skv} spike {spike}")
    s2 = (q.float().transpose(1, 2) @ k.float().transpose(1, 2).transpose(-1, -2)) * (128**-0.5 * 1.4426950408889634)
    lse_ref = torch.logsumexp(s2 * 0.6931471805599453, -1) * 1.4426950408889634
    assert (lse.cpu() - lse_ref).abs().max().item() < 2e-2
    o2 = ops.attn_dense(qd, kd, vd, layout="bshd", key_splits=1)
    assert torch.equal(o, o2)


def test_attn_dense_kernel_choice(ops):
    """fvk_attn_dense_kernel_bf16: the long-key kernel on 16x16x32 (1, the default) or 32x32x16 MFMAs (2) — both within the attention bound of
    the fp32 reference and within rounding of each other, with a spiked key (each one's exact recompute) and a ragged tail; below 2048 keys
    the choice is ignored (8-wave kernel: bit-identical outputs); an unknown kernel id is refused."""
    B, H, Sq, Skv = 1, 3, 700, 2600
    q, k, v = rnd((B, Sq, H, 128), 1), rnd((B, Skv, H, 128), 2), rnd((B, Skv, H, 128), 3)
    k[0, 2500, 1] = q[0, 300, 1] * 6
    ref = W.attention_fp32_ref(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2), 128**-0.5).transpose(1, 2)
    qd, kd, vd = q.to(DEV), k.to(DEV), v.to(DEV)
    outs = {kk: ops.attn_dense(qd, kd, vd, layout="bshd", kernel=kk, return_lse=True) for kk in (0, 1, 2, 3)}
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])   # 0 = the default = 1
    for kk in (1, 2, 3):
        _attn_check(outs[kk][0], ref, f"long-key kernel {kk}")
    assert (outs[1][0].float() - outs[2][0].float()).abs().max().item() < 2e-2
    assert (outs[1][0].float() - outs[3][0].float()).abs().max().item() < 2e-2   # 3 = the 8-wave online-softmax kernel at any length
    assert (outs[1][1] - outs[2][1]).abs().max().item() < 1e-2   # lse: sums of bf16 vs fp32 probabilities
    assert (outs[3][1] - outs[2][1]).abs().max().item() < 1e-2
    short = [ops.attn_dense(qd, kd[:, :1500], vd[:, :1500], layout="bshd", kernel=kk) for kk in (0, 1, 2, 3)]
    assert torch.equal(short[0], short[1]) and torch.equal(short[0], short[2]) and torch.equal(short[0], short[3])
    with pytest.raises(RuntimeError, match="kernel=4"):
        ops.attn_dense(qd, kd, vd, layout="bshd", kernel=4)
    with pytest.raises(RuntimeError, match="key_splits"):   # the split-KV form runs attn_w16: naming another kernel with it is refused
        ops.attn_dense(qd, kd, vd, layout="bshd", kernel=2, key_splits=2)


def test_attn_dense_whole_tensor_vs_torch_sdpa(ops):
    """Whole-tensor parity of the three dense kernels against what the reference's SDPAImpl.forward calls on this device
    (torch.nn.functional.scaled_dot_product_attention on bf16 ROCm tensors, sdpa.py:134-147) and against exact fp32 softmax, on the input
    distribution of the reference's own attention-kernel test (randn q / k / v, fastvideo-kernel/tests/test_sta.py:60-91) at 12 heads x
    4 680 tokens (three latent frames of 480p).  The long-key kernels keep a FIXED softmax reference (no online rescale), i.e. they round
    P at other points than a flash kernel — dense is not exempt from the attention bound for that: every kernel must sit within the
    thresholds below of BOTH checkers, and the spread between kernels is printed."""
    import torch.nn.functional as F
    torch.manual_seed(0)
    B, H, S = 1, 12, 4680
    q, k, v = (torch.randn((B, S, H, 128), device=DEV, dtype=torch.bfloat16) for _ in range(3))
    exact = torch.softmax((q.float().transpose(1, 2) @ k.float().transpose(1, 2).transpose(-1, -2)) * 128**-0.5, -1) @ v.float().transpose(1, 2)
    sdpa = F.scaled_dot_product_attention(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2)).float()
    base = (sdpa - exact).abs()
    print(f"torch SDPA vs exact fp32: avg {base.mean().item():.3g} max {base.max().item():.3g}")
    for kk, name in ((1, "attn_w16"), (2, "attn_w64"), (3, "attn_pp2")):
        o = ops.attn_dense(q, k, v, layout="bshd", kernel=kk).float().transpose(1, 2)
        e_exact, e_sdpa = (o - exact).abs(), (o - sdpa).abs()
        print(f"{name}: vs exact fp32 avg {e_exact.mean().item():.3g} max {e_exact.max().item():.3g}; vs torch SDPA avg {e_sdpa.mean().item():.3g} "
              f"max {e_sdpa.max().item():.3g}")
        assert torch.isfinite(o).all()
        # a bf16 output of magnitude ~0.05 rounds at ~1e-4; the reference's kernel test allows max 4e-2 (test_sta.py:88-91)
        assert e_exact.max().item() < 4e-2 and e_sdpa.max().item() < 4e-2, name
        assert e_exact.mean().item() < 2.0 * base.mean().item() + 2e-5, f"{name}: mean error vs exact fp32 more than 2x torch SDPA's own"


@pytest.mark.parametrize("B,H,Sq,Skv,splits", [(1, 3, 512, 4096, 4), (2, 2, 300, 2500, 3), (1, 1, 256, 1000, 8), (1, 2, 1030, 5000, 2), (1, 2, 256, 300, 5)])
def test_attn_dense_key_splits(ops, B, H, Sq, Skv, splits):
    """fvk_attn_dense_split_bf16: the key axis cut into runs of whole 128-key stages (one workgroup per run) + the LSE-weighted merge — the form
    the small per-rank grids of sequence parallelism take.  Against the fp32 reference and the un-split kernel (same tolerance: the merge
    adds one fp32 rescale per run), with ragged lengths, more runs than stages (empty runs), a spiked key in one run (that run's exact
    pass) and the merged LSE."""
    q, k, v = rnd((B, Sq, H, 128), 1), rnd((B, Skv, H, 128), 2), rnd((B, Skv, H, 128), 3)
    k[0, Skv - 7, 0] = q[0, 5, 0] * 6   # growth of ~2^98 over the first sub-tile's maximum, in the last run
    ref = W.attention_fp32_ref(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2), 128**-0.5).transpose(1, 2)
    qd, kd, vd = q.to(DEV), k.to(DEV), v.to(DEV)
    o1, l1 = ops.attn_dense(qd, kd, vd, layout="bshd", return_lse=True, key_splits=1)
    os_, ls = ops.attn_dense(qd, kd, vd, layout="bshd", return_lse=True, key_splits=splits)
    _attn_check(os_, ref, f"split-KV x{splits} {B},{H},{Sq},{Skv}")
    assert (os_.float() - o1.float()).abs().max().item() < 4e-2   # both are within the attention bound of the reference
    # the row sums are sums of the bf16-rounded probabilities — the values the numerator uses, so a run's O_r * 2^lse_r is exact and the merge is
    # self-consistent — which leaves each reported LSE with P's rounding (~2^-9 relative on l, ~3e-3 in log2 units; measured <= 7.1e-3 between
    # the split and the un-split walk of a spiked row)
    assert torch.isfinite(ls).all() and (ls - l1).abs().max().item() < 1e-2
    # the automatic choice leaves a grid that fills the chip alone and cuts one that does not
    assert ops.attn_key_splits(1536, 256) == 1 and ops.attn_key_splits(192, 256) == 4 and ops.attn_key_splits(384, 256) == 2
    assert ops.attn_key_splits(192, 8) == 1 and ops.attn_key_splits(432, 72) == 1   # too few stages / a grid that is full enough


def test_attn_block_sparse(ops):
    B, H, nq, nk = 1, 2, 5, 7
    q, k, v = rnd((B, H, nq * 64, 128), 1), rnd((B, H, nk * 64, 128), 2), rnd((B, H, nk * 64, 128), 3)
    rng = np.random.default_rng(0)
    bm = rng.random((B, H, nq, nk)) < 0.5
    bm[..., 0] = True
    vbs = np.array([64, 64, 48, 64, 1, 33, 24], dtype=np.int32)
    ref = V.block_sparse_attn(q, k, v, bm, vbs)
    idx, num = V.map_to_index(bm)
    out, lse = ops.attn_block_sparse(q.to(DEV), k.to(DEV), v.to(DEV), torch.from_numpy(idx).to(DEV), torch.from_numpy(num).to(DEV),
                                     torch.from_numpy(vbs).to(DEV), layout="bhsd", return_lse=True)
    _attn_check(out, ref, "block sparse")
    # lse (base 2, scaled) against the oracle's masked scores
    s = torch.matmul(q.float(), k.float().transpose(-1, -2)) * 128**-0.5
    bmf = torch.from_numpy(bm).view(B, H, nq, 1, nk, 1).expand(B, H, nq, 64, nk, 64)
    col = (torch.arange(64)[None, :] < torch.from_numpy(vbs.astype(np.int64))[:, None]).view(1, 1, 1, 1, nk, 64)
    s = s.masked_fill(~(bmf & col).reshape(B, H, nq * 64, nk * 64), float("-inf"))
    lse_ref = torch.logsumexp(s, dim=-1) * 1.4426950408889634
    assert (lse.cpu() - lse_ref).abs().max().item() < 2e-2


def test_attn_block_sparse_cold_paths(ops):
    """The paths of attn_bs16 that random data never takes (round 6): a listed block of size 0 — as the FIRST block of a list (it would have set
    the softmax reference: the rows go through the exact pass), in the middle and at the end —, one-block and odd-length lists, an empty list, and
    rows whose later keys score so far above the first block's that the fixed-reference row sum overflows (2^90 and beyond: the per-row exact
    recompute).  Against the oracle's masked fp32 attention."""
    B, H, nq, nk = 1, 3, 6, 9
    q, k, v = rnd((B, H, nq * 64, 128), 11), rnd((B, H, nk * 64, 128), 12), rnd((B, H, nk * 64, 128), 13)
    vbs = np.array([0, 64, 17, 0, 64, 33, 64, 0, 5], dtype=np.int32)
    bm = np.zeros((B, H, nq, nk), dtype=bool)
    bm[0, 0, 0, [0, 1, 2]] = True            # size-0 block first
    bm[0, 0, 1, [1, 3, 4]] = True            # size-0 block in the middle
    bm[0, 0, 2, [2, 5, 7]] = True            # size-0 block last, odd length
    bm[0, 0, 3, [4]] = True                  # one block
    bm[0, 0, 4, :] = True                    # all nine (odd), three of them empty
    # head 0, query block 5: empty list
    bm[0, 1] = np.random.default_rng(3).random((nq, nk)) < 0.6
    bm[0, 1, :, 1] = True
    bm[0, 2] = bm[0, 1]
    # head 2: overflow of the fixed reference — the first listed block (1) scores low, block 6 scores 16 384 raw units higher
    q[0, 2] = 8.0
    k[0, 2, 1 * 64:2 * 64] = -8.0
    k[0, 2, 6 * 64:7 * 64] = 8.0
    bm[0, 2, :, 6] = True
    ref = torch.nan_to_num(V.block_sparse_attn(q, k, v, bm, vbs), nan=0.0)
    idx, num = V.map_to_index(bm)
    out, lse = ops.attn_block_sparse(q.to(DEV), k.to(DEV), v.to(DEV), torch.from_numpy(idx).to(DEV), torch.from_numpy(num).to(DEV),
                                     torch.from_numpy(vbs).to(DEV), layout="bhsd", return_lse=True)
    _attn_check(out, ref, "block sparse, cold paths")
    assert (out[0, 0, 5 * 64:6 * 64] == 0).all()
    # the overflow rows attend block 6 alone to fp32 precision: the mean of its V rows
    want = v[0, 2, 6 * 64:7 * 64].float().mean(0)
    assert (out[0, 2].float().cpu() - want).abs().max().item() < 2e-2


def test_attn_block_sparse_split_last_round(ops):
    """fvk_attn_block_sparse_ws_bf16 (round 6): the workgroups of the launch's last, partly empty round walk their lists in 2-4 PARTS that are
    merged by LSE weights.  Small grid (every workgroup is "last round": 4 parts): lists of every length class — empty, shorter than the part
    count, odd, full —, ragged blocks, a size-0 block where a part starts, a spiked key inside one part (that part's exact pass); against the
    oracle's masked fp32 attention, the un-split call, and the merged LSE."""
    B, H, nq, nk = 1, 2, 8, 80
    q, k, v = rnd((B, H, nq * 64, 128), 21), rnd((B, H, nk * 64, 128), 22), rnd((B, H, nk * 64, 128), 23)
    rng = np.random.default_rng(5)
    vbs = rng.integers(1, 65, nk).astype(np.int32)
    vbs[rng.random(nk) < 0.6] = 64
    bm = rng.random((B, H, nq, nk)) < 0.7
    bm[0, 0, 0, :] = False                       # empty list
    bm[0, 0, 1, :] = False; bm[0, 0, 1, [3, 50, 77]] = True   # 3 blocks: fewer than the parts x 2
    bm[0, 0, 2, :] = True                        # all 80
    bm[0, 0, 3, :] = False; bm[0, 0, 3, :41] = True           # odd
    idx, num = V.map_to_index(bm)
    # list (0, 1, 0): its second part starts with a size-0 block (per = ceil(n / 4) rounded up to even)
    n010 = int(num[0, 1, 0]); per = ((n010 + 3) // 4 + 1) & ~1
    vbs[idx[0, 1, 0, per]] = 0
    # list (0, 1, 5): one key of its LAST part scores 2^100 above the rest
    last = int(idx[0, 1, 5, int(num[0, 1, 5]) - 1])
    k[0, 1, last * 64 + 3] = q[0, 1, 5 * 64 + 9] * 6
    ref = torch.nan_to_num(V.block_sparse_attn(q, k, v, bm, vbs), nan=0.0)
    dv = lambda t: torch.from_numpy(t).to(DEV)
    args = (q.to(DEV), k.to(DEV), v.to(DEV), dv(idx), dv(num), dv(vbs))
    o1, l1 = ops.attn_block_sparse(*args, layout="bhsd", return_lse=True, split_last_round=False)
    o4, l4 = ops.attn_block_sparse(*args, layout="bhsd", return_lse=True)
    _attn_check(o1, ref, "block sparse, whole lists")
    _attn_check(o4, ref, "block sparse, split last round")
    assert (o4[0, 0, :64] == 0).all()
    assert (o4.float() - o1.float()).abs().max().item() < 4e-2
    fin = torch.isfinite(l1)
    assert (torch.isfinite(l4) == fin).all() and (l4[fin] - l1[fin]).abs().max().item() < 1e-2


def test_attn_block_sparse_split_last_round_full_rounds_bit_identical(ops):
    """A grid of one full round + a tail (260 workgroups on 256 CUs): the lists of the full round are bit-identical to the un-split call, the
    four workgroups of the tail (split in 4 parts) agree to rounding."""
    B, H, nq, nk = 1, 2, 520, 70
    g = torch.Generator().manual_seed(7)
    q, k, v = (torch.randn((B, H, n * 64, 128), generator=g).to(torch.bfloat16).to(DEV) for n in (nq, nk, nk))
    rng = np.random.default_rng(9)
    bm = rng.random((B, H, nq, nk)) < 0.6
    bm[..., 0] = True
    vbs = np.full(nk, 64, dtype=np.int32); vbs[-1] = 40
    idx, num = V.map_to_index(bm)
    dv = lambda t: torch.from_numpy(t).to(DEV)
    o1 = ops.attn_block_sparse(q, k, v, dv(idx), dv(num), dv(vbs), layout="bhsd", split_last_round=False)
    o4 = ops.attn_block_sparse(q, k, v, dv(idx), dv(num), dv(vbs), layout="bhsd")
    assert torch.isfinite(o4).all()
    d = (o4.float() - o1.float()).abs().amax(dim=-1).view(-1)            # per (head, row), flat = workgroup order
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    total_wg = H * (nq // 4)
    full_rows = (total_wg - total_wg % cus) * 256
    assert total_wg % cus != 0 and d[:full_rows].max().item() == 0.0
    assert 0.0 < d[full_rows:].max().item() < 2e-2


@pytest.mark.parametrize("rows", [256, 384, 512])
def test_attn_tile_lists_shared_kv_lists(ops, rows):
    """fvk_attn_tile_lists_bf16: every `rows` consecutive query rows share one list of 64-key blocks (sliding-tile windows).  Against the
    block-sparse oracle with the tile's list repeated for each of its 64-row query blocks — lists of odd and even length, in arbitrary
    order, with partially filled and EMPTY (size 0) blocks, an empty list, and row groups skipped through q_rows_valid — and against
    fvk_attn_block_sparse_bf16 (one list per 128 rows) on the same lists."""
    B, H, nl, nk = 1, 2, 4, 11
    q, k, v = rnd((B, H, nl * rows, 128), 1), rnd((B, H, nk * 64, 128), 2), rnd((B, H, nk * 64, 128), 3)
    rng = np.random.default_rng(rows)
    bm = rng.random((B, H, nl, nk)) < 0.55
    bm[..., 0] = True
    bm[0, 1, 2, :] = False                       # an empty list: zeros
    bm[0, 0, 1, :] = False
    bm[0, 0, 1, [2, 5, 9]] = True                # an odd-length list (the last 128-key step has one real half)
    vbs = np.array([64, 64, 48, 64, 1, 33, 24, 64, 0, 64, 17], dtype=np.int32)
    idx, num = V.map_to_index(bm)
    for b_, h_, l_ in ((0, 0, 0), (0, 1, 3)):    # list order is free: shuffle two of them
        n_ = int(num[b_, h_, l_])
        idx[b_, h_, l_, :n_] = rng.permutation(idx[b_, h_, l_, :n_])
    rep = rows // 64
    ref = torch.nan_to_num(V.block_sparse_attn(q, k, v, np.repeat(bm, rep, axis=2), vbs), nan=0.0)  # empty lists: zeros by contract
    valid = np.array([rows, 100, rows, 300 if rows > 256 else 256], dtype=np.int32)  # real query rows per list
    dv = lambda t: torch.from_numpy(t).to(DEV)
    out = ops.attn_tile_lists(q.to(DEV), k.to(DEV), v.to(DEV), dv(idx), dv(num), dv(vbs), rows, dv(valid), layout="bhsd").cpu()
    groups = [(0, 256)] * (rows >= 256) + [(256, 512)] * (rows >= 512) + [(rows - 128, rows)] * (rows % 256 == 128)
    for l_ in range(nl):
        for r0, r1 in groups:
            sl = slice(l_ * rows + r0, l_ * rows + r1)
            if r0 >= valid[l_]:
                assert (out[:, :, sl] == 0).all(), f"list {l_} rows {r0}..{r1}: skipped group must be zeros"
            else:
                _attn_check(out[:, :, sl], ref[:, :, sl], f"tile lists rows_per_list={rows} list {l_} rows {r0}..{r1}")
    assert (out[0, 1, 2 * rows:3 * rows] == 0).all()  # the empty list
    # the same lists, one per 128-row block, through the 4-wave kernel
    o128 = ops.attn_block_sparse(q.to(DEV), k.to(DEV), v.to(DEV), dv(np.repeat(idx, rows // 128, axis=2)), dv(np.repeat(num, rows // 128, axis=2)),
                                 dv(vbs), layout="bhsd", q_block=128).cpu()
    full = ops.attn_tile_lists(q.to(DEV), k.to(DEV), v.to(DEV), dv(idx), dv(num), dv(vbs), rows, None, layout="bhsd").cpu()
    err = (full.float() - o128.float()).abs()
    assert err.max().item() < 4e-2 and err.mean().item() < 2e-3, (err.max().item(), err.mean().item())
    if rows % 256 == 0:  # scattered output rows: row r -> o_rows[r] (a permutation with some rows dropped), bit-identical values
        Sq = nl * rows
        o_rows = torch.randperm(Sq + 64, generator=g(6))[:Sq].to(torch.int32)
        o_rows[::7] = -1
        sc = ops.attn_tile_lists(q.to(DEV), k.to(DEV), v.to(DEV), dv(idx), dv(num), dv(vbs), rows, None, layout="bhsd", o_rows=o_rows.to(DEV),
                                 n_out_rows=Sq + 64).cpu()
        kept = o_rows >= 0
        assert sc.shape == (B, H, Sq + 64, 128) and torch.equal(sc[:, :, o_rows[kept].long()], full[:, :, kept])
    else:
        with pytest.raises(RuntimeError, match="o_rows"):
            ops.attn_tile_lists(q.to(DEV), k.to(DEV), v.to(DEV), dv(idx), dv(num), dv(vbs), rows, None, layout="bhsd",
                                o_rows=torch.zeros(nl * rows, dtype=torch.int32, device=DEV), n_out_rows=nl * rows)


def test_attn_tile_lists_batch_and_head_specific_lists(ops):
    """B = 2, H = 3 with a DIFFERENT list for every (batch, head, group): the list and count arrays are indexed [b, h, list]."""
    B, H, nl, nk, rows = 2, 3, 3, 8, 256
    q, k, v = rnd((B, H, nl * rows, 128), 11), rnd((B, H, nk * 64, 128), 12), rnd((B, H, nk * 64, 128), 13)
    rng = np.random.default_rng(5)
    bm = rng.random((B, H, nl, nk)) < 0.5
    bm[..., 3] = True
    vbs = np.array([64, 10, 64, 64, 64, 40, 64, 64], dtype=np.int32)
    idx, num = V.map_to_index(bm)
    assert len({tuple(idx[b_, h_, l_, :num[b_, h_, l_]]) for b_ in range(B) for h_ in range(H) for l_ in range(nl)}) > 12
    ref = V.block_sparse_attn(q, k, v, np.repeat(bm, rows // 64, axis=2), vbs)
    dv = lambda t: torch.from_numpy(t).to(DEV)
    out = ops.attn_tile_lists(q.to(DEV), k.to(DEV), v.to(DEV), dv(idx), dv(num), dv(vbs), rows, None, layout="bhsd")
    _attn_check(out, ref, "tile lists, per-(batch, head) lists")
    out_s = ops.attn_tile_lists(q.transpose(1, 2).contiguous().to(DEV), k.transpose(1, 2).contiguous().to(DEV),
                                v.transpose(1, 2).contiguous().to(DEV), dv(idx), dv(num), dv(vbs), rows, None, layout="bshd")
    assert torch.equal(out_s.transpose(1, 2), out)   # the two layouts differ in strides only


def test_attn_tile_lists_refuses_bad_geometry(ops):
    q, k, v = rnd((1, 1, 768, 128), 1).to(DEV), rnd((1, 1, 256, 128), 2).to(DEV), rnd((1, 1, 256, 128), 3).to(DEV)
    idx, num, vbs = torch.zeros((1, 1, 2, 4), dtype=torch.int32, device=DEV), torch.ones((1, 1, 2), dtype=torch.int32, device=DEV), \
        torch.full((4,), 64, dtype=torch.int32, device=DEV)
    for rows in (128, 320):  # 128-row lists belong to fvk_attn_block_sparse_bf16; 320 is not a multiple of 128
        with pytest.raises(RuntimeError, match="rows_per_list"):
            ops.attn_tile_lists(q, k, v, idx, num, vbs, rows, None, layout="bhsd")
    with pytest.raises(RuntimeError, match="multiple of rows_per_list"):
        ops.attn_tile_lists(q, k, v, idx, num, vbs, 512, None, layout="bhsd")


@pytest.mark.parametrize("canvas,tile,win", [((4, 8, 16), (2, 8, 8), [(1, 1, 1), (3, 1, 3)]),
                                             ((12, 16, 24), (6, 8, 8), [(3, 3, 3), (1, 3, 1), (3, 1, 5)])])
def test_attn_sta(ops, canvas, tile, win):
    tv = math.prod(tile)
    S = math.prod(canvas)
    B, H = 1, len(win)
    q, k, v = rnd((B, H, S, 128), 1), rnd((B, H, S, 128), 2), rnd((B, H, S, 128), 3)
    ct = tuple(c // t for c, t in zip(canvas, tile))
    out = ops.attn_sta(q.to(DEV), k.to(DEV), v.to(DEV), ct, tv, win, layout="bhsd")
    for h, w in enumerate(win):
        mask = V.sta_mask(canvas, w, tile)
        ref = W.attention_fp32_ref(q[:, h:h + 1], k[:, h:h + 1], v[:, h:h + 1], 128**-0.5, mask)
        _attn_check(out[:, h:h + 1], ref, f"sta head {h} window {w}")


# ------------------------------------------------------------------ VSA pieces (integer parts bit exact)
def test_vsa_combine_and_its_scatter_form(ops):
    """out = bf16(bf16(out_c * gate) + out_s) per token (fastvideo_kernel/ops.py:120-133) and the form with tile(gate) / untile(out) folded
    in: gate read and result written in TOKEN order through token_of_row (padding rows skipped) — bit-identical to gather, combine, gather."""
    B, H, S_pad, n_tok, blk = 1, 3, 320, 250, 64
    out_c, out_s = rnd((B, H, S_pad // blk, 128), 1).to(DEV), rnd((B, S_pad, H, 128), 2).to(DEV)
    gate_tok = rnd((B, n_tok, 4 * H, 128), 3).to(DEV)[:, :, 3 * H:]            # a column block of a wider buffer (the fused QKV+gate rows)
    token_of_row = torch.full((S_pad,), -1, dtype=torch.int32)
    token_of_row[torch.randperm(S_pad, generator=g(4))[:n_tok]] = torch.randperm(n_tok, generator=g(5)).to(torch.int32)
    rows_of_tok = torch.empty(n_tok, dtype=torch.long)
    real = token_of_row >= 0
    rows_of_tok[token_of_row[real].long()] = torch.arange(S_pad)[real]
    gate_tiled = torch.zeros((B, S_pad, H, 128), dtype=torch.bfloat16, device=DEV)
    gate_tiled[:, rows_of_tok.to(DEV)] = gate_tok
    tiled = ops.vsa_combine(out_c, out_s, gate_tiled, blk, layout="bshd")
    oc = out_c.float().cpu().repeat_interleave(blk, dim=2).transpose(1, 2)     # [B,S_pad,H,D]
    ref = ((oc * gate_tiled.float().cpu()).bfloat16().float() + out_s.float().cpu()).bfloat16()
    assert torch.equal(tiled.cpu(), ref)
    got = ops.vsa_combine(out_c, out_s, gate_tok, blk, layout="bshd", token_of_row=token_of_row.to(DEV), n_tokens=n_tok)
    assert got.shape == (B, n_tok, H, 128) and torch.equal(got, tiled[:, rows_of_tok.to(DEV)])
    nog = ops.vsa_combine(out_c, out_s, None, blk, layout="bshd", token_of_row=token_of_row.to(DEV), n_tokens=n_tok)
    assert torch.equal(nog, ops.vsa_combine(out_c, out_s, None, blk, layout="bshd")[:, rows_of_tok.to(DEV)])


@pytest.mark.parametrize("n,topk", [(50, 9), (624, 125), (624, 63), (1440, 288), (7, 7), (300, 1), (2160, 432), (8192, 100), (65, 64)])
def test_topk_mask_bit_exact(ops, n, topk):
    sc = rnd((3, 5, n), 1, 2.0)  # bf16 scores have many exact ties
    sc[0, 0, :] = 0.5            # an all-equal row
    sc[1, 1, :] = (torch.arange(n) % 3).to(sc.dtype)  # three values only: the tie rule decides almost everything
    ref = V.topk_mask_bisect(sc.float().numpy(), topk)
    got = ops.topk_mask(sc.to(DEV), topk).cpu().numpy()     # one wave per row (the block-per-row variant: scripts/probes/variant_tests.py)
    got32 = ops.topk_mask(sc.float().to(DEV), topk).cpu().numpy()
    assert np.array_equal(got, ref)
    assert (got.sum(-1) == min(topk, n)).all()
    assert np.array_equal(got32, ref)


def test_map_to_index_and_gather(ops):
    rng = np.random.default_rng(3)
    bm = rng.random((2, 3, 9, 200)) < 0.3
    idx, num = ops.map_to_index(torch.from_numpy(bm).to(DEV))
    ri, rn = V.map_to_index(bm)
    assert np.array_equal(idx.cpu().numpy(), ri) and np.array_equal(num.cpu().numpy(), rn)
    # tile / untile through the gather kernel == oracle scatter/gather
    lat = (5, 14, 6)
    md = V.build_metadata(lat)
    S, S_pad = md["total_seq_length"], math.prod(md["num_tiles"]) * 64
    x = rnd((2, S, 3, 128), 4)
    t_ref = V.tile(x, md)
    i32 = lambda a: torch.from_numpy(a.astype(np.int32)).to(DEV)
    t_got = ops.gather_rows(x.to(DEV), S_pad, i32(md["tile_partition_indices"]), i32(md["non_pad_index"]), zero_init=True)
    assert torch.equal(t_got.cpu(), t_ref)
    u_got = ops.gather_rows(t_got, S, i32(md["untile_combined_index"]), None)
    assert torch.equal(u_got.cpu(), x)


def test_block_mean(ops):
    vbs = np.array([64, 48, 1, 64, 24], dtype=np.int32)
    x = rnd((2, 3, 5 * 64, 128), 1)
    for b in range(5):
        x[:, :, b * 64 + vbs[b]:(b + 1) * 64] = 0
    ref = V.block_mean(x, vbs, 64)
    got = ops.block_mean(x.to(DEV), torch.from_numpy(vbs).to(DEV), 64)
    close(got, ref, atol=1e-2, rtol=1e-2, what="block mean")


def test_block_mean_with_the_tile_gather_folded_in(ops):
    """fvk_block_mean_gather_bf16 (round 6): the block means of a token-order tensor through the tile map == the means of the gathered copy, bit
    for bit (ragged tiles: padding rows are zeros in the copy and skipped here); and _vsa_forward with v in token order (v_src_rows) == with the
    tile-major copy."""
    from fastvideo_amd import kernel_api as KA
    lat = (8, 20, 14)  # dit (8,10,7) -> 12 ragged tiles
    md = V.build_metadata(lat)
    S, B, H = md["total_seq_length"], 1, 3
    wide = rnd((B, S, 4 * H * 128), 5).to(DEV)                      # the model host's fused q | k | v | gate rows
    x = wide[:, :, 2 * H * 128:3 * H * 128].view(B, S, H, 128)      # a strided token-order view (row stride 4 d)
    vbs = torch.from_numpy(md["variable_block_sizes"]).to(DEV)
    nb = vbs.numel()
    perm, npi = (torch.from_numpy(md[k_]).int().to(DEV) for k_ in ("tile_partition_indices", "non_pad_index"))
    tiled = ops.gather_rows(x, nb * 64, perm, npi, zero_init=True)
    tok = torch.full(((nb * 64 + 127) // 128 * 128,), -1, dtype=torch.int32)
    tok[npi.cpu().long()] = perm.cpu()
    tok = tok.to(DEV)
    assert torch.equal(ops.block_mean(x, vbs, 64, layout="bshd", src_rows=tok), ops.block_mean(tiled, vbs, 64, layout="bshd"))
    assert torch.equal(ops.v_transpose(x, src_rows=tok)[..., :nb * 64], ops.v_transpose(tiled)[..., :nb * 64])
    q, k, gate = (ops.gather_rows(rnd((B, S, H, 128), s_).to(DEV), nb * 64, perm, npi, zero_init=True) for s_ in (1, 2, 4))
    topk = V.compute_topk(0.5, nb)
    a = KA._vsa_forward(q, k, tiled, vbs, vbs, topk, gate, "bshd")
    b_ = KA._vsa_forward(q, k, x, vbs, vbs, topk, gate, "bshd", v_src_rows=tok)
    assert torch.equal(a, b_)


def test_video_sparse_attn_composite(ops):
    from fastvideo_amd import kernel_api as KA
    lat = (8, 20, 14)  # dit (8,10,7) -> tiles (2,3,2)=12 blocks, ragged
    md = V.build_metadata(lat)
    S = md["total_seq_length"]
    B, H = 1, 2
    q, k, v, gate = (rnd((B, S, H, 128), s) for s in (1, 2, 3, 4))
    tq, tk, tv, tg = (V.tile(t, md).transpose(1, 2).contiguous() for t in (q, k, v, gate))
    vbs = md["variable_block_sizes"]
    topk = V.compute_topk(0.5, len(vbs))
    ref, inter = V.video_sparse_attn(tq, tk, tv, vbs, vbs, topk, 64, tg)
    tvbs = torch.from_numpy(vbs).to(DEV)
    out, gi = KA.video_sparse_attn(tq.to(DEV), tk.to(DEV), tv.to(DEV), tvbs, tvbs, topk, (4, 4, 4), tg.to(DEV),
                                   return_intermediates=True)
    close(gi["q_c"], inter["q_c"], what="q_c")
    close(gi["scores"], inter["scores"], atol=2e-2, rtol=2e-2, what="coarse scores")
    # the mask must be the exact top-k of *our* scores (bit-exact selection rule) ...
    assert np.array_equal(gi["mask"].cpu().numpy(), V.topk_mask_bisect(gi["scores"].float().cpu().numpy(), topk))
    # ... and, given the same mask, the sparse branch must match the oracle
    ref_s = V.block_sparse_attn(tq, tk, tv, gi["mask"].cpu().numpy(), vbs)
    valid_rows = torch.from_numpy(md["non_pad_index"])
    _attn_check(gi["out_s"][:, :, valid_rows], ref_s[:, :, valid_rows], "vsa sparse branch")
    # ... and the whole composite, UNCONDITIONALLY: the oracle composite is evaluated with the device's block selection (a bf16 ulp in one
    # coarse score can flip a near-tie between the two top-k computations; the selection rule itself was checked bit-exactly above)
    ref_same_mask, _ = V.video_sparse_attn(tq, tk, tv, vbs, vbs, topk, 64, tg, mask_override=gi["mask"].cpu().numpy())
    _attn_check(out[:, :, valid_rows], ref_same_mask.float()[:, :, valid_rows], "vsa composite (device mask)")
    if np.array_equal(gi["mask"].cpu().numpy(), inter["mask"]):
        _attn_check(out[:, :, valid_rows], ref.float()[:, :, valid_rows], "vsa composite")


# ------------------------------------------------------------------ glue
def test_patchify_unpatchify_time_silu(ops):
    lat = rnd((2, 16, 3, 10, 14), 1)
    assert torch.equal(ops.patchify(lat.to(DEV)).cpu(), W.patchify(lat))
    x = rnd((2, 3 * 5 * 7, 64), 2)
    assert torch.equal(ops.unpatchify(x.to(DEV), (2, 16, 3, 10, 14)).cpu(), W.unpatchify(x, (3, 5, 7), (1, 2, 2), 16))
    t = torch.tensor([500.0, 3.0, 999.0])
    close(ops.timestep_embedding(t, 256), W.timestep_embedding(t, 256).bfloat16(), atol=8e-3, rtol=8e-3, what="t-emb")
    s = rnd((4, 256), 3, 3.0)
    close(ops.silu(s.to(DEV)), torch.nn.functional.silu(s), what="silu")


def test_errors_are_loud(ops):
    x = rnd((4, 100), 1).to(DEV)
    with pytest.raises(RuntimeError):
        ops.gemm(x, rnd((8, 100), 2).to(DEV))  # K % 64 != 0
    with pytest.raises(RuntimeError):
        ops.ln_modulate(rnd((4, 100), 1))  # CPU tensor: no fallback
    with pytest.raises(RuntimeError):
        ops.attn_dense(rnd((1, 8, 1, 64), 1).to(DEV), rnd((1, 8, 1, 64), 2).to(DEV), rnd((1, 8, 1, 64), 3).to(DEV))


# ------------------------------------------------------------------ large-tile kernels (gemm_pp.hip) at shapes that take them by default
@pytest.mark.parametrize("M,N,K", [(1000, 1536, 1536), (777, 520, 96), (513, 264, 32), (2048, 256, 64), (129, 8, 1536)])
def test_gemm_pp_shapes(ops, M, N, K):
    """Shapes that take the 256x256 LDS-DMA ping-pong kernel: ragged M/N tiles, 1..48 K-steps (ring prologue / tail re-reads)."""
    x, w, b = rnd((M, K), 11), rnd((N, K), 12, K**-0.5), rnd((N, ), 13)
    close(ops.gemm(x.to(DEV), w.to(DEV), b.to(DEV)), _lin_ref(x, w, b), what=f"gemm_pp {M}x{N}x{K}")


def test_gemm_pp_identity_asymmetric(ops):
    K = 512
    x = torch.eye(K).bfloat16()
    w = (torch.arange(K * K).view(K, K) % 251).float().bfloat16()
    assert torch.equal(ops.gemm(x.to(DEV), w.to(DEV), None).cpu(), w.t().contiguous())
    # every K-step of a longer K lands in the right ring slot: x = [0 | I | 0] picks one K-step of w
    K2 = 1536
    x2 = torch.zeros((512, K2)).bfloat16()
    x2[:, 512:1024] = torch.eye(512).bfloat16()
    w2 = (torch.arange(300 * K2).view(300, K2) % 241).float().bfloat16()[:296]
    assert torch.equal(ops.gemm(x2.to(DEV), w2.to(DEV), None).cpu(), w2[:, 512:1024].t().contiguous())
