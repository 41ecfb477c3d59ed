"""Tests of the NON-SHIPPING kernel variants and of cross-kernel agreement, run against the MEASUREMENT build of the library
(scripts/probes/libfvk_probe.so = the product sources with -DFVK_PROBE_BUILD + attn_pp.hip / attn_vsa.hip; the product library
libfvk_amd.so holds the shipped configuration only and refuses non-zero fvk_set_tunable values).

Not collected by `pytest tests/` directly: tests/test_gpu_probe_variants.py runs this file in a subprocess with FVK_PROBE_LIB=1 (one
process binds one of the two libraries).  By hand:  FVK_PROBE_LIB=1 python -m pytest scripts/probes/variant_tests.py -q"""
import os
import sys

assert os.environ.get("FVK_PROBE_LIB") == "1", "run with FVK_PROBE_LIB=1 (see the module docstring)"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

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

def _lin_ref(x, w, b):
    return (x.float() @ w.float().t() + (0 if b is None else b.float())).bfloat16()


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


@pytest.mark.parametrize("impl", [53, 54, 55, 56])
def test_attn_block_sparse_measurement_variants(ops, impl):
    """The A/B variants of the 64-row list entry kept in the measurement build ("attn_impl" 55: the round-1..5 kernel of attn_fwd.hip — two lists
    per 8-wave workgroup; 53: the same with register-staged loader waves, bit-identical to 55; 54: attn_vsa.hip, all waves compute, key-split
    with a final merge: equal to 55 to rounding; 56: the shipped attn_bs16 on hardware workgroup ids, bit-identical to the shipped deal) on ragged
    block sizes, odd list counts (an unpaired last list), an empty list and lists from 1 to 9 tiles, against the oracle."""
    B, H, nq, nk = 1, 3, 7, 9
    q, k, v = rnd((B, H, nq * 64, 128), 1), rnd((B, H, nk * 64, 128), 2), rnd((B, H, nk * 64, 128), 3)
    rng = np.random.default_rng(impl)
    bm = rng.random((B, H, nq, nk)) < 0.5
    bm[..., 0] = True
    bm[0, 0, 2, :] = False
    bm[0, 0, 2, 4] = True          # a one-tile list
    bm[0, 1, 3, :] = True          # all nine
    bm[0, 2, 5, :] = False         # an empty list
    vbs = np.array([64, 64, 48, 64, 1, 33, 24, 64, 17], dtype=np.int32)
    ref = torch.nan_to_num(V.block_sparse_attn(q, k, v, bm, vbs), nan=0.0)
    idx, num = V.map_to_index(bm)
    args = (q.to(DEV), k.to(DEV), v.to(DEV), torch.from_numpy(idx).to(DEV), torch.from_numpy(num).to(DEV), torch.from_numpy(vbs).to(DEV))
    shipped, shipped_lse = ops.attn_block_sparse(*args, layout="bhsd", return_lse=True)
    _attn_check(shipped, ref, "block sparse, shipped (attn_bs16)")
    try:
        ops.set_tunable("attn_impl", 55)
        base, base_lse = ops.attn_block_sparse(*args, layout="bhsd", return_lse=True)
        ops.set_tunable("attn_impl", impl)
        out, lse = ops.attn_block_sparse(*args, layout="bhsd", return_lse=True)
    finally:
        ops.set_tunable("attn_impl", 0)
    _attn_check(out, ref, f"block sparse, attn_impl {impl}")
    assert (out[0, 2, 5 * 64:6 * 64] == 0).all() and (shipped[0, 2, 5 * 64:6 * 64] == 0).all()
    live_h2 = torch.ones(nq, dtype=torch.bool); live_h2[5] = False
    if impl in (53, 55):
        assert torch.equal(out, base) and torch.equal(lse, base_lse)
    elif impl == 56:
        assert torch.equal(out, shipped) and torch.equal(lse, shipped_lse)
    else:
        assert (out.float() - base.float()).abs().max().item() < 8e-3
        sel = lse[0, 2].view(nq, 64)[live_h2.to(DEV)]
        assert (sel - base_lse[0, 2].view(nq, 64)[live_h2.to(DEV)]).abs().max().item() < 1e-3
    # the shipped kernel against the round-1 kernel: the same attention to rounding (fixed vs running softmax reference; LSE from the bf16-P row sum)
    assert (shipped.float() - base.float()).abs().max().item() < 8e-3
    sel = shipped_lse[0, 2].view(nq, 64)[live_h2.to(DEV)]
    assert (sel - base_lse[0, 2].view(nq, 64)[live_h2.to(DEV)]).abs().max().item() < 2e-2


@pytest.mark.parametrize("n,topk", [(50, 9), (624, 125), (1440, 288), (7, 7), (8192, 100), (65, 64)])
def test_topk_mask_block_per_row_variant_bit_exact(ops, n, topk):
    """"vsa_impl" 1 = one workgroup per row (the first version; the shipped kernel runs one wave per row): same masks."""
    sc = rnd((3, 5, n), 1, 2.0)
    sc[0, 0, :] = 0.5
    sc[1, 1, :] = (torch.arange(n) % 3).to(sc.dtype)
    ref = V.topk_mask_bisect(sc.float().numpy(), topk)
    base = ops.topk_mask(sc.to(DEV), topk).cpu().numpy()
    ops.set_tunable("vsa_impl", 1)
    try:
        got = ops.topk_mask(sc.to(DEV), topk).cpu().numpy()
    finally:
        ops.set_tunable("vsa_impl", 0)
    assert np.array_equal(got, ref) and np.array_equal(base, ref)


# ------------------------------------------------------------------ large-tile kernels (gemm_pp.hip / attn_pp.hip) and their A/B switches
@pytest.fixture
def tunables(ops):
    yield ops.set_tunable
    ops.set_tunable("gemm_impl", 0)
    ops.set_tunable("attn_impl", 0)
    ops.set_tunable("vae_conv_impl", 0)


def test_gemm_kernels_agree(ops, tunables):
    """gemm_ph (gemm_impl 228 = its shipped schedule) and the 128x128 kernel accumulate 16-k MFMA steps in the same order: identical outputs
    except for the GELU formulation; the shipped gemm_w1 (32-k MFMA steps) agrees to rounding."""
    M, N, K, B = 600, 768, 256, 2
    x, w, b = rnd((M, K), 1), rnd((N, K), 2, K**-0.5), rnd((N, ), 3)
    res, gate = rnd((M, N), 4, 2.0), rnd((B, N), 5, 0.5, torch.float32)
    outs = {}
    for impl in (228, 1, 0):
        tunables("gemm_impl", impl)
        outs[impl] = [ops.gemm(x.to(DEV), w.to(DEV), b.to(DEV), epilogue=e, residual=res.to(DEV) if e == ops.EPI_RESIDUAL_GATE else None,
                               gate=gate.to(DEV) if e == ops.EPI_RESIDUAL_GATE else None).cpu()
                      for e in (ops.EPI_NONE, ops.EPI_SILU, ops.EPI_RESIDUAL_GATE, ops.EPI_GELU_TANH)]
    for i in range(3):
        assert torch.equal(outs[228][i], outs[1][i]), f"epilogue #{i}: kernels disagree"
        close(outs[0][i], outs[228][i], atol=3e-2, rtol=2e-2, what=f"gemm_w1 vs gemm_ph, epilogue #{i}")
    outs[0] = outs[228]
    close(outs[0][3], outs[1][3], atol=1e-2, rtol=1e-2, what="gelu formulations")
    y = _lin_ref(x, w, b)
    close(outs[0][3], W.gelu_tanh(y), what="gelu (pp)")
    ref = W.scale_residual(res.view(B, M // B, N), y.view(B, M // B, N), gate.view(B, 1, N)).bfloat16().view(M, N)
    close(outs[0][2], ref, what="residual+gate (pp)")


@pytest.mark.parametrize("M,N,K", [(300, 264, 128), (777, 1000, 192), (1030, 520, 1536), (513, 256, 4096)])
def test_gemm_ph_pp_persistent_bit_identical(ops, tunables, M, N, K):
    """gemm_ph (shipped, K-step 64), its persistent-workgroup variant and gemm_pp (K-step 32) accumulate the same 16-k MFMA steps in the
    same order: byte-identical outputs for every epilogue, with a gate whose batch boundary cuts through a wave's 128 rows, a strided A
    operand (column block of a wider buffer) and the batched entry."""
    B = 3
    Mb = M // B * B
    x, w, b = rnd((Mb, 2 * K), 1)[:, K // 2:K // 2 + K], rnd((N, K), 2, K**-0.5), rnd((N, ), 3)
    res, gate = rnd((Mb, N), 4, 2.0), rnd((B, N), 5, 0.5, torch.float32)
    xd = rnd((Mb, 2 * K), 1).to(DEV)[:, K // 2:K // 2 + K]  # row stride 2K
    outs = {}
    for impl in (228, 2, 1252, 5, 13, 21, 29, 0, 61, 125, 1149):  # gemm_ph shipped | gemm_pp | gemm_ph persistent | gemm_w1 with 32x32x16 MFMAs | shipped gemm_w1
        tunables("gemm_impl", impl)
        outs[impl] = [ops.gemm(xd, w.to(DEV), b.to(DEV), epilogue=e, residual=res.to(DEV) if e == ops.EPI_RESIDUAL_GATE else None,
                               gate=gate.to(DEV) if e == ops.EPI_RESIDUAL_GATE else None).cpu()
                      for e in (ops.EPI_NONE, ops.EPI_SILU, ops.EPI_GELU_TANH, ops.EPI_RESIDUAL_GATE)]
        xb, wb = rnd((2, 150, K), 7), rnd((2, 140, K), 8)
        outs[impl].append(ops.gemm_batched(xb.to(DEV), wb.to(DEV), ops.EPI_DIV, 11.0).cpu())
    for impl in (2, 1252, 5, 13, 21, 29):
        for i, (a_, b_) in enumerate(zip(outs[228], outs[impl])):
            assert torch.equal(a_, b_), f"impl {impl}, output #{i}"
    for i, (a_, b_) in enumerate(zip(outs[0], outs[125])):
        assert torch.equal(a_, b_), f"shipped gemm_w1 is variant 125 (or 1149 = the same with streaming stores, for wide outputs), output #{i}"
    for i, (a_, b_) in enumerate(zip(outs[1149], outs[125])):
        assert torch.equal(a_, b_), f"streaming stores changed output #{i}"
    for i, (a_, b_) in enumerate(zip(outs[125], outs[61])):  # direct epilogue + persistent workgroups: the same arithmetic per output element
        assert torch.equal(a_, b_), f"gemm_w1 direct-epilogue variant differs from the bounce epilogue, output #{i}"
    outs[0] = outs[228]
    close(outs[0][0], _lin_ref(x, w, b), what=f"gemm_ph {Mb}x{N}x{K}")
    # gemm_w1 with 16x16x32 MFMAs (K = 32 per instruction): another summation order inside the MFMA, so equal only to rounding
    for impl in (37, 45, 53, 61):
        tunables("gemm_impl", impl)
        got = [ops.gemm(xd, w.to(DEV), b.to(DEV), epilogue=e, residual=res.to(DEV) if e == ops.EPI_RESIDUAL_GATE else None,
                        gate=gate.to(DEV) if e == ops.EPI_RESIDUAL_GATE else None).cpu()
               for e in (ops.EPI_NONE, ops.EPI_SILU, ops.EPI_GELU_TANH, ops.EPI_RESIDUAL_GATE)]
        for i, (a_, b_) in enumerate(zip(outs[0], got)):
            close(b_, a_, atol=3e-2, rtol=2e-2, what=f"impl {impl}, output #{i}")
        assert (got[0].float() - outs[0][0].float()).abs().mean() < 1e-3


@pytest.mark.parametrize("M,N,K,rowwise", [(700, 520, 512, False), (1030, 264, 1536, True), (300, 1000, 256, False)])
def test_gemm_fp8_w1_and_pp_agree(ops, tunables, M, N, K, rowwise):
    """fvk_gemm_fp8 on gemm_w1's kernel (16x16x128 MX-fp8 MFMAs, the default for K % 256 == 0) and on gemm_pp's (32x32x64, gemm_impl 2): the
    same products summed in another order — equal to a bf16 ulp — for every epilogue, with ragged M / N tiles and per-row / per-channel scales."""
    x, w, b = rnd((M, K), 1), rnd((N, K), 2, K**-0.5), rnd((N, ), 3)
    res, gate = rnd((M, N), 4, 2.0), rnd((2, N), 5, 0.5, torch.float32)
    xq, xs = ops.fp8_quantize(x.to(DEV), rowwise=rowwise)
    wq, ws = ops.fp8_quantize(w.to(DEV), rowwise=rowwise)
    outs = {}
    for impl in (2, 0):
        tunables("gemm_impl", impl)
        outs[impl] = [ops.gemm_fp8(xq, xs, wq, ws, b.to(DEV), epilogue=e, residual=res.to(DEV) if e == ops.EPI_RESIDUAL_GATE else None,
                                   gate=gate.to(DEV) if e == ops.EPI_RESIDUAL_GATE else None, rows_per_batch=M // 2 if e == ops.EPI_RESIDUAL_GATE else None).cpu()
                      for e in (ops.EPI_NONE, ops.EPI_SILU, ops.EPI_GELU_TANH, ops.EPI_RESIDUAL_GATE)]
    for i, (a_, b_) in enumerate(zip(outs[2], outs[0])):
        close(b_, a_, atol=3e-2, rtol=2e-2, what=f"fp8 epilogue #{i}")
        assert (a_.float() - b_.float()).abs().mean().item() < 1e-3


@pytest.mark.parametrize("impl", [0, 1, 2, 3, 99, 200, 202, 300, 302, 318])
@pytest.mark.parametrize("B,H,Sq,Skv", [(1, 2, 700, 700), (2, 3, 512, 130), (1, 1, 256, 64), (1, 2, 1030, 1999)])
def test_attn_dense_impls(ops, tunables, impl, B, H, Sq, Skv):
    """attn_impl 0 = the shipped choice (attn_pp2 below 2048 keys, attn_w16 above), 300 = attn_w16 forced, 302 = the same in hardware workgroup
    order, 200 / 202 = attn_w64 (its 32x32x16 predecessor), 99 = attn_pp2, 2 / 3 = attn_pp (64-key tiles, two DMA placements), 1 = the 4-wave
    kernel; ragged Sq / Skv tails, 1..32 KV tiles."""
    tunables("attn_impl", impl)
    q, k, v = rnd((B, Sq, H, 128), 1), rnd((B, Skv, H, 128), 2), rnd((B, Skv, H, 128), 3)
    ref = W.attention_fp32_ref(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2), 128**-0.5).transpose(1, 2)
    out = ops.attn_dense(q.to(DEV), k.to(DEV), v.to(DEV), layout="bshd")
    _attn_check(out, ref, f"dense impl {impl} {B},{H},{Sq},{Skv}")


def test_attn_pp2_schedules_are_bit_identical(ops, tunables):
    """The schedule variants of the 128-key-tile kernel (attn_pp2.hip: attn_impl 99 = its final one-barrier / leading-group in-stream DMA schedule;
    103 = round 1's two-barrier schedule; 111 = V^T pieces inside the trailing matrix segment; 105 / 107 = one barrier with the trailing /
    leading group issuing ahead of the segment) move DMA issue and barriers only: same arithmetic in the same order, so the outputs must
    be bit-identical — also with a late rescale spike and ragged tails, and across repeated launches (a slot reused too early or a
    barrier miscount shows up as a difference or a hang)."""
    B, H, Sq, Skv = 2, 3, 1030, 2999
    q, k, v = rnd((B, Sq, H, 128), 1, 0.7), rnd((B, Skv, H, 128), 2, 0.7), rnd((B, Skv, H, 128), 3)
    k[0, 2500, 1] = q[0, 700, 1] * 6
    qd, kd, vd = q.to(DEV), k.to(DEV), v.to(DEV)
    outs = {}
    for impl in (99, 103, 105, 107, 111, 99):
        tunables("attn_impl", impl)
        outs.setdefault(impl, []).append(ops.attn_dense(qd, kd, vd, layout="bshd").cpu())
    ref = W.attention_fp32_ref(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2), 128**-0.5).transpose(1, 2)
    _attn_check(outs[99][0], ref, "attn_pp2 final schedule")
    for impl, lst in outs.items():
        for o in lst:
            assert torch.equal(o, outs[99][0]), f"attn_impl {impl} differs from attn_pp2's final schedule"
    # attn_w16 (shipped above 2048 keys: fixed softmax reference, 32-k MFMA steps, row sums of the bf16 P) and attn_w64 agree with it to rounding
    for impl in (300, 200):
        tunables("attn_impl", impl)
        o64 = ops.attn_dense(qd, kd, vd, layout="bshd").cpu()
        _attn_check(o64, ref, f"attn_impl {impl}")
        assert (o64.float() - outs[99][0].float()).abs().max().item() < 2e-2


@pytest.mark.parametrize("impl", [0, 2, 3, 99, 200, 300])
def test_attn_pp_rescale_branch_and_repeatability(ops, tunables, impl):
    """Spiked keys force the running-max rescale in late tiles of the ping-pong kernel; 3 launches must agree bit-for-bit
    (a race between the staggered wave groups or an early LDS read would show up as run-to-run differences)."""
    tunables("attn_impl", impl)
    B, H, S = 1, 2, 640
    q, k, v = rnd((B, S, H, 128), 1, 0.5), rnd((B, S, H, 128), 2, 0.5), rnd((B, S, H, 128), 3)
    k[0, 250, 0] = q[0, 7, 0] * 6
    k[0, 600, 1] = q[0, 300, 1] * 6
    ref = W.attention_fp32_ref(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2), 128**-0.5).transpose(1, 2)
    outs = [ops.attn_dense(q.to(DEV), k.to(DEV), v.to(DEV), layout="bshd").cpu() for _ in range(3)]
    _attn_check(outs[0], ref, f"rescale branch impl {impl}")
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])


def test_attn_w64_list_mode_agrees_with_the_shipped_list_kernel(ops, tunables):
    """attn_w64's list mode ("attn_impl" 72; not shipped: attn_fwd.hip says why) over window-class query groups of a ragged sliding-tile
    grid — partial blocks, odd list lengths, scattered output rows — against the shipped attn_pp2 list mode: same lists, same masks, the
    outputs differ only by where P is rounded."""
    from fastvideo_amd import kernel_api
    grid, H = (7, 10, 20), 2
    h = kernel_api.sliding_tile_block_lists(grid, (2, 4, 8), (3, 1, 3))
    gq = torch.Generator().manual_seed(3)
    q = torch.randn((1, h["group_rows"], H, 128), generator=gq).bfloat16().to(DEV)
    k, v = (torch.randn((1, h["S_pad"], H, 128), generator=gq).bfloat16().to(DEV) for _ in range(2))
    ex = lambda t, n: t.to(DEV)[None, None].expand(1, H, *([-1] * n)).contiguous()
    idx, num, bs = ex(h["group_q2k_idx"], 2), ex(h["group_q2k_num"], 1), h["block_sizes"].to(DEV)
    tok = torch.full((h["group_rows"],), -1, dtype=torch.int32)
    tok[h["group_dst"].long()] = h["group_src"]
    n_tok = grid[0] * grid[1] * grid[2]
    outs = {}
    for impl in (0, 72):
        tunables("attn_impl", impl)
        outs[impl] = (ops.attn_tile_lists(q, k, v, idx, num, bs, 256, None, layout="bshd").cpu(),
                      ops.attn_tile_lists(q, k, v, idx, num, bs, 256, None, layout="bshd", o_rows=tok.to(DEV), n_out_rows=n_tok).cpu())
    real = (tok >= 0)
    for a_, b_ in zip(outs[0], outs[72]):
        assert torch.isfinite(b_.float()).all()
    d = (outs[72][0][0, real].float() - outs[0][0][0, real].float()).abs()
    assert d.max().item() < 2e-2 and d.mean().item() < 2e-4, (d.max().item(), d.mean().item())
    assert torch.equal(outs[72][1][0, tok[real].long()], outs[72][0][0, real])   # the scattered form stores the same rows at their tokens


# ------------------------------------------------------------------ Wan VAE 3x3 conv: the one-wave-per-SIMD kernel vs the 8-wave kernel
@pytest.mark.parametrize("Cin,Cout,T,H,W,kt,residual,norm,ups", [
    (96, 96, 2, 37, 70, 3, False, False, False), (96, 96, 1, 16, 32, 1, True, False, False), (192, 192, 2, 19, 45, 3, True, False, False),
    (384, 384, 1, 12, 40, 3, False, False, False), (96, 96, 2, 21, 33, 3, True, True, False), (192, 192, 1, 9, 64, 3, False, True, False),
    (192, 96, 2, 18, 34, 3, False, True, False),
    # the resample convs: 2x nearest-exact upsample folded into the staging (H, W = the upsampled size; odd input sizes, several tiles)
    (192, 96, 2, 38, 70, 1, False, True, True), (384, 192, 1, 22, 74, 1, False, False, True), (384, 192, 2, 18, 66, 1, False, True, True)])
def test_vae_conv3w_vs_8wave_kernel(ops, tunables, Cin, Cout, T, H, W, kt, residual, norm, ups):
    """vae_conv3w.hip (round 4: 4 waves x (4 rows x 32 px x 96 ch), 16x16x32 MFMAs, direct epilogue) against vae_conv3.hip's 8-wave kernel
    ("vae_conv_impl" 3) on multi-tile, ragged shapes: the two sum the same products in a different fp32 order (32-k vs 16-k MFMA steps), so
    every output is within ONE bf16 ulp of the other and the great majority are identical; the fused-norm ring likewise (its ||x||^2 is summed
    in another order too)."""
    ring_in, start = T + kt - 1, (1 if kt == 3 else 0)
    w = rnd((Cout, kt * 9 * Cin), 1, (kt * 9 * Cin)**-0.5).to(DEV)
    b = rnd((Cout,), 2, 0.1).to(DEV)
    x = rnd((ring_in, H // 2, W // 2, Cin) if ups else (ring_in, H, W, Cin), 3).to(DEV)
    res = rnd((T, H, W, Cout), 5).to(DEV) if residual else None
    gamma = (1 + rnd((Cout,), 4, 0.1, torch.float32)).to(DEV)
    outs = {}
    for impl in (0, 3):
        tunables("vae_conv_impl", impl)
        if norm:
            nring = T + 1
            ringbuf = torch.full((nring, H, W, Cout), 3.0, dtype=torch.bfloat16, device=DEV)
            raw = ops.vae_conv_norm(x, w, b, gamma, ringbuf, T=T, H=H, W=W, kt=kt, norm_slot0=1, ring_start=start, residual=res, upsample2x=ups)
            outs[impl] = (raw.float().cpu(), ringbuf.float().cpu())
        else:
            outs[impl] = (ops.vae_conv(x, w, b, T=T, H=H, W=W, kt=kt, ks=3, ring_start=start, residual=res, upsample2x=ups).float().cpu(),)
    for new, old in zip(outs[0], outs[3]):
        assert torch.isfinite(new).all()
        diff = (new - old).abs()
        ulp = old.abs().clamp_min(2.0**-20) * 2.0**-7
        assert (diff <= ulp).all(), f"max diff {diff.max().item():.4g} beyond one bf16 ulp (ref {old.flatten()[diff.argmax()].item():.4g})"
        assert (diff > 0).float().mean().item() < 0.15, f"{(diff > 0).float().mean().item():.3f} of the elements differ: not a summation-order effect"


# ------------------------------------------------------------------ gemm_w1n (256 x 128 tiles) vs gemm_w1 (256 x 256 tiles)
@pytest.mark.parametrize("M,N,K", [(4095, 1536, 1536), (4095, 1536, 8960), (300, 200, 128), (1000, 1160, 256), (8190, 4608, 1536)])
def test_gemm_w1n_is_byte_identical_to_gemm_w1(ops, tunables, M, N, K):
    """gemm_impl 6 forces the 256 x 128 tile kernel, 14 forbids it: same MFMA steps in the same order + the same epilogue code -> identical bytes,
    for every epilogue, ragged M / N edges included."""
    x, w, b = rnd((M, K), 1).to(DEV), rnd((N, K), 2, K**-0.5).to(DEV), rnd((N,), 3).to(DEV)
    res, gate = rnd((M, N), 4, 2.0).to(DEV), rnd((2, N), 5, 0.5, torch.float32).to(DEV)
    outs = {}
    for impl in (6, 14):
        tunables("gemm_impl", impl)
        outs[impl] = [ops.gemm(x, w, b, epilogue=e, residual=res if e == ops.EPI_RESIDUAL_GATE else None, gate=gate if e == ops.EPI_RESIDUAL_GATE else None,
                               rows_per_batch=(M + 1) // 2 if e == ops.EPI_RESIDUAL_GATE else None).cpu()
                      for e in (ops.EPI_NONE, ops.EPI_SILU, ops.EPI_RESIDUAL_GATE, ops.EPI_GELU_TANH)]
    for i, (a_, b_) in enumerate(zip(outs[6], outs[14])):
        assert torch.equal(a_, b_), f"epilogue #{i}: max diff {(a_.float() - b_.float()).abs().max().item():.4g}"
    close(outs[6][0], _lin_ref(x.cpu(), w.cpu(), b.cpu()), atol=3e-2, rtol=2e-2, what="gemm_w1n vs fp32")
