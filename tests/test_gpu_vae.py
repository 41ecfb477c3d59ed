"""GPU parity of the Wan VAE decode path (HIP kernels through the C ABI) against the oracle (oracle/vae_oracle.py, pinned to
the real reference by tests/test_vae_oracle.py) and against plain torch fp32 references of the individual ops.

Tolerances: the HIP path keeps activations in bf16 with fp32 accumulation, the reference default is fp32
(fastvideo/configs/pipelines/wan.py:54) and its own reduced-precision mode is bf16 autocast (spark_performance.md:56).
Per-op: atol = rtol = 1e-2 on bf16 outputs (as fastvideo-kernel/tests/test_turbodiffusion.py:143).  Full decode: pixels in
[-1, 1]; mean |err| <= 1e-2 and max |err| <= 2x the max error of the reference's own bf16-autocast decode of the same latent
(stored in the golden fixture), floor 6e-2."""
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "vae_tiny.pt")


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from fastvideo_amd import ops as o
    return o


def rnd(shape, seed, scale=1.0):
    return (torch.randn(shape, generator=torch.Generator().manual_seed(seed)) * scale)


def cl(x):  # [C,T,H,W] -> channels-last [T,H,W,C] bf16 on the GPU
    return x.permute(1, 2, 3, 0).contiguous().cuda().bfloat16()


def conv_ref(x_hist, w, b, kt, ks):
    """x_hist [Cin, T+kt-1, H, W] fp32 (history frames first); zero padding in space only."""
    p = ks // 2
    return F.conv3d(F.pad(x_hist[None], (p, p, p, p, 0, 0)), w, b)[0]


def flat_w(w):  # [Cout,Cin,kt,kh,kw] -> [Cout, taps*Cin] bf16 on the GPU
    return w.permute(0, 2, 3, 4, 1).reshape(w.shape[0], -1).contiguous().cuda().bfloat16()


@pytest.mark.parametrize("Cin,Cout,T,H,W,kt,ks", [(96, 96, 2, 20, 28, 3, 3), (192, 192, 1, 17, 9, 3, 3), (64, 384, 3, 8, 8, 3, 3),
                                                  (384, 384, 2, 12, 10, 3, 1), (32, 128, 4, 30, 30, 3, 3), (96, 3 * 8, 2, 16, 16, 3, 3),
                                                  # round 4: multi-tile, ragged shapes of the one-wave-per-SIMD kernel (vae_conv3w.hip): 96-wide
                                                  # tiles 16 x 32 px, 192-wide tiles 8 x 32 px, two n-tiles at 384 channels, kt = 1
                                                  (96, 96, 2, 37, 70, 3, 3), (192, 192, 2, 19, 45, 3, 3), (384, 384, 1, 12, 40, 3, 3),
                                                  (96, 96, 1, 16, 32, 1, 3), (192, 96, 1, 33, 65, 3, 3), (96, 192, 2, 9, 31, 3, 3)])
def test_conv_causal_ring(ops, Cin, Cout, T, H, W, kt, ks):
    """Conv over [2 history frames | T frames] stored in a ring at a non-zero start (wrap-around), vs torch conv3d."""
    w = rnd((Cout, Cin, kt, ks, ks), 1, (kt * ks * ks * Cin)**-0.5)
    b = rnd((Cout,), 2, 0.1)
    x = rnd((Cin, T + 2, H, W), 3)
    xb, wb, bb = x.bfloat16().float(), w.bfloat16().float(), b.bfloat16().float()
    ref = conv_ref(xb, wb, bb, kt, ks)[:, :T]                                 # [Cout, T, H, W] (kt = 1: the first T of the T + 2 stored frames)
    ring, start = T + 2, 1
    buf = torch.empty((ring, H, W, Cin), dtype=torch.bfloat16, device="cuda")
    frames = cl(x)
    for l in range(T + 2):
        buf[(start + l) % ring] = frames[l]
    y = ops.vae_conv(buf, flat_w(w), b.cuda().bfloat16(), T=T, H=H, W=W, kt=kt, ks=ks, ring_start=start)
    torch.testing.assert_close(y.float().cpu().permute(3, 0, 1, 2), ref, atol=1e-2, rtol=1e-2)


def test_conv_residual_and_frame_stride(ops):
    Cin, Cout, T, H, W = 96, 96, 2, 12, 20
    w, b, x = rnd((Cout, Cin, 3, 3, 3), 1, (27 * Cin)**-0.5), rnd((Cout,), 2, 0.1), rnd((Cin, T + 2, H, W), 3)
    res = rnd((Cout, T, H, W), 4)
    ref = conv_ref(x.bfloat16().float(), w.bfloat16().float(), b.bfloat16().float(), 3, 3).bfloat16().float() + res.bfloat16().float()
    out = torch.zeros((2 * T, H, W, Cout), dtype=torch.bfloat16, device="cuda")
    ops.vae_conv(cl(x), flat_w(w), b.cuda().bfloat16(), T=T, H=H, W=W, kt=3, ks=3, out=out[1], out_frame_stride=2 * H * W * Cout,
                 residual=cl(res))
    got = out.float().cpu()
    torch.testing.assert_close(got[1::2].permute(3, 0, 1, 2), ref, atol=2e-2, rtol=1e-2)
    assert got[0::2].abs().max() == 0  # the interleaved frames were not touched


def test_conv2d_upsample_fold(ops):
    """nearest-exact 2x upsample + Conv2d (WanResample, wanvae.py:277-284) == our gather with upsample2x."""
    Cin, Cout, T, H, W = 192, 96, 3, 10, 14
    w, b, x = rnd((Cout, Cin, 3, 3), 1, (9 * Cin)**-0.5), rnd((Cout,), 2, 0.1), rnd((T, Cin, H, W), 3)
    up = F.interpolate(x.bfloat16().float(), scale_factor=(2.0, 2.0), mode="nearest-exact")
    ref = F.conv2d(up, w.bfloat16().float(), b.bfloat16().float(), padding=1)          # [T, Cout, 2H, 2W]
    y = ops.vae_conv(x.permute(0, 2, 3, 1).contiguous().cuda().bfloat16(), flat_w(w[:, :, None]), b.cuda().bfloat16(), T=T, H=2 * H,
                     W=2 * W, kt=1, ks=3, upsample2x=True)
    torch.testing.assert_close(y.float().cpu().permute(0, 3, 1, 2), ref, atol=1e-2, rtol=1e-2)


def test_conv_final_epilogue_fp32_planar_clamp(ops):
    Cin, T, H, W = 96, 4, 24, 40
    w, b, x = rnd((3, Cin, 3, 3, 3), 1, 3 * (27 * Cin)**-0.5), rnd((3,), 2, 0.1), rnd((Cin, T + 2, H, W), 3)
    ref = conv_ref(x.bfloat16().float(), w.bfloat16().float(), b.bfloat16().float(), 3, 3).clamp(-1, 1)
    assert (ref.abs() == 1).any(), "the test must exercise the clamp"
    out = torch.full((3, T + 3, H, W), 7.0, device="cuda")
    ops.vae_conv(cl(x), flat_w(w), b.cuda().bfloat16(), T=T, H=H, W=W, kt=3, ks=3, out_f32=out[:, 2:], plane_stride=(T + 3) * H * W)
    got = out.cpu()
    torch.testing.assert_close(got[:, 2:2 + T], ref, atol=5e-3, rtol=0)
    assert (got[:, :2] == 7).all() and (got[:, 2 + T:] == 7).all()


@pytest.mark.parametrize("T,H,W,start", [(1, 16, 32, 0), (2, 17, 33, 2), (5, 50, 70, 3), (7, 9, 100, 6), (16, 48, 64, 11)])
def test_conv_out_rolling_three_frames(ops, T, H, W, start):
    """conv_out (96 -> 3 channels, vae_convout.hip, round 6): every input frame multiplied once by the three time taps' weight rows, the running
    sums of three output frames in three lane groups.  Ring with wrap-around, one to sixteen frames (every residue of T mod 3), ragged tiles,
    one tile and several; vs torch conv3d in fp32, and the same frames computed in passes of other lengths are BIT-identical."""
    Cin = 96
    w, b, x = rnd((3, Cin, 3, 3, 3), 1, 2 * (27 * Cin)**-0.5), rnd((3,), 2, 0.1), rnd((Cin, T + 2, H, W), 3)
    ref = conv_ref(x.bfloat16().float(), w.bfloat16().float(), b.bfloat16().float(), 3, 3).clamp(-1, 1)
    ring = T + 3
    buf = torch.zeros((ring, H, W, Cin), dtype=torch.bfloat16, device="cuda")
    frames = cl(x)
    for l in range(T + 2):
        buf[(start + l) % ring] = frames[l]
    wf, bf = flat_w(w), b.cuda().bfloat16()
    out = torch.full((3, T, H, W), 7.0, device="cuda")
    ops.vae_conv(buf, wf, bf, T=T, H=H, W=W, kt=3, ks=3, ring_start=start, out_f32=out, plane_stride=T * H * W)
    torch.testing.assert_close(out.cpu(), ref, atol=5e-3, rtol=0)
    if T >= 5:  # the same output frames in two passes (3 frames, then the rest)
        out2 = torch.full((3, T, H, W), 7.0, device="cuda")
        ops.vae_conv(buf, wf, bf, T=3, H=H, W=W, kt=3, ks=3, ring_start=start, out_f32=out2, plane_stride=T * H * W)
        ops.vae_conv(buf, wf, bf, T=T - 3, H=H, W=W, kt=3, ks=3, ring_start=(start + 3) % ring, out_f32=out2[:, 3:], plane_stride=T * H * W)
        assert torch.equal(out2, out)


@pytest.mark.parametrize("C,silu", [(96, True), (192, True), (384, False), (32, True), (128, False)])
def test_rmsnorm_silu_into_ring(ops, C, silu):
    T, HW, ring, slot0 = 3, 77, 5, 3
    x, g = rnd((T * HW, C), 1, 2.0), 1 + rnd((C,), 2, 0.1)
    xb = x.bfloat16().float()
    ref = F.normalize(xb, dim=1) * C**0.5 * g
    ref = F.silu(ref) if silu else ref
    buf = torch.zeros((ring, HW, C), dtype=torch.bfloat16, device="cuda")
    ops.vae_rmsnorm_silu(x.cuda().bfloat16(), g.cuda(), buf, HW=HW, slot0=slot0, silu=silu)
    got = buf.float().cpu()
    for t in range(T):
        torch.testing.assert_close(got[(slot0 + t) % ring], ref[t * HW:(t + 1) * HW], atol=1e-2, rtol=1e-2)
    untouched = [s for s in range(ring) if s not in {(slot0 + t) % ring for t in range(T)}]
    assert all(got[s].abs().max() == 0 for s in untouched)


@pytest.mark.parametrize("S", [200, 1000])
def test_mid_block_attention_head_dim_384(ops, S):
    qkv = rnd((S, 1152), 1)
    q, k, v = (qkv[:, i * 384:(i + 1) * 384].bfloat16().float() for i in range(3))
    ref = F.scaled_dot_product_attention(q[None, None], k[None, None], v[None, None])[0, 0]
    g = qkv.cuda().bfloat16()
    o = ops.attn_dense_wide(g[:, :384], g[:, 384:768], g[:, 768:])
    err = (o.float().cpu() - ref).abs()
    assert err.mean() < 3e-3 and err.max() < 4e-2, (err.mean().item(), err.max().item())  # fastvideo-kernel/tests/test_sta.py:88-91


def test_mid_block_attention_frames_in_one_launch(ops):
    """The decoder passes the pass's frames as the batch of ONE launch (WanAttentionBlock folds t into the batch too, wanvae.py:486-489): every
    frame's rows are the bytes the frame gets alone."""
    T, S = 3, 520
    g = rnd((T, S, 1152), 2).cuda().bfloat16()
    o = ops.attn_dense_wide(g[:, :, :384], g[:, :, 384:768], g[:, :, 768:])
    assert o.shape == (T, S, 384)
    for t in range(T):
        assert torch.equal(o[t], ops.attn_dense_wide(g[t, :, :384], g[t, :, 384:768], g[t, :, 768:]))


def _decode_check(sd, z, y_ref, max_tol):
    from fastvideo_amd.wan_vae import WanVaeDecoderHip
    dec = WanVaeDecoderHip(sd, device="cuda")
    y = dec.decode(z.cuda()).cpu()
    assert y.shape == y_ref.shape and y.dtype == torch.float32
    err = (y - y_ref).abs()
    assert err.mean() <= 1e-2 and err.max() <= max_tol, f"mean {err.mean().item():.4g} max {err.max().item():.4g} (tol {max_tol:.3g})"
    assert y.abs().max() <= 1.0
    return y


def test_decode_tiny_vs_reference_golden(ops):
    """Full chunked decode (3 latent frames -> 9 pixel frames, ragged 4x6 latent) vs the real reference's fp32 output."""
    from oracle.vae_oracle import seeded_state_dict
    g = torch.load(GOLD, weights_only=False)
    sd = seeded_state_dict(g["param_spec"], g["seed"])
    ref_bf16_err = (g["y_bf16_autocast"] - g["y"]).abs().max().item()
    y = _decode_check(sd, g["z"], g["y"], max(6e-2, 2 * ref_bf16_err))
    # causality / cache bookkeeping: decoding only the first two latent frames gives the same first five pixel frames
    from fastvideo_amd.wan_vae import WanVaeDecoderHip
    y2 = WanVaeDecoderHip(sd, device="cuda").decode(g["z"][:, :, :2].cuda()).cpu()
    assert torch.equal(y2, y[:, :, :5])


def test_decode_wan21_geometry_vs_oracle(ops):
    """The real Wan2.1 VAE decoder geometry (base_dim 96: 384/384/384/192/96 channels, 384-wide mid attention) on a small
    latent, vs the oracle in fp32 on the host."""
    from oracle.vae_oracle import WanVaeDecoderOracle, seeded_state_dict
    g = torch.load(GOLD, weights_only=False)
    spec = [(n, tuple(d * 3 if d % 32 == 0 else d for d in s)) for n, s in g["param_spec"]
            if n.startswith(("decoder.", "post_quant_conv."))]  # same parameter list, channel counts scaled 32 -> 96
    sd = seeded_state_dict(spec, 5)
    z = rnd((1, 16, 2, 6, 4), 9)
    y_ref = WanVaeDecoderOracle(sd).decode(z)
    _decode_check(sd, z, y_ref, 8e-2)


@pytest.mark.parametrize("C,Cin,residual,ups,want_raw", [(96, 96, False, False, True), (96, 192, False, True, True), (96, 96, True, False, True),
                                                         (192, 192, False, False, False), (192, 192, True, False, True),
                                                         (192, 384, False, True, True)])
def test_conv_norm_epilogue_vs_conv_then_norm_kernel(ops, C, Cin, residual, ups, want_raw):
    """fvk_vae_conv_norm_bf16 at BOTH fused widths (96: one wave holds a pixel's channels; 192: two waves swap partial sums of squares
    through LDS) against the un-fused pair on the same inputs: the raw output must be bit-identical to fvk_vae_conv_bf16's, and the ring
    contents must equal fvk_vae_rmsnorm_silu_bf16 of that raw output up to the fp32 summation order of ||x||^2 — i.e. a rare one-ulp flip of
    a bf16 element, never more (a wrong gamma slice, a missing half of the sum or a wrong ring slot would be far outside that)."""
    T, H, W, kt = 3, 18, 40, (1 if ups else 3)
    Hin, Win = (H // 2, W // 2) if ups else (H, W)
    ring_in = T + kt - 1
    w = flat_w(rnd((C, Cin, kt, 3, 3), 1, (kt * 9 * Cin)**-0.5))
    b = rnd((C,), 2, 0.1).cuda().bfloat16()
    x = rnd((ring_in, Hin, Win, Cin), 3).cuda().bfloat16()
    gamma = (1 + rnd((C,), 4, 0.1)).cuda()
    res = rnd((T, H, W, C), 5).cuda().bfloat16() if residual else None
    nring, slot0, start = T + 2, T, (1 if kt == 3 else 0)
    fused_ring = torch.full((nring, H, W, C), 3.0, dtype=torch.bfloat16, device="cuda")
    raw_f = ops.vae_conv_norm(x, w, b, gamma, fused_ring, T=T, H=H, W=W, kt=kt, norm_slot0=slot0, ring_start=start, residual=res,
                              upsample2x=ups, want_raw=want_raw)
    raw = ops.vae_conv(x, w, b, T=T, H=H, W=W, kt=kt, ks=3, ring_start=start, residual=res, upsample2x=ups)
    if want_raw:
        assert torch.equal(raw_f, raw)
    else:
        assert raw_f is None
    ref_ring = torch.full((nring, H, W, C), 3.0, dtype=torch.bfloat16, device="cuda")
    ops.vae_rmsnorm_silu(raw, gamma, ref_ring, HW=H * W, slot0=slot0, silu=True)
    a_, b_ = fused_ring.float(), ref_ring.float()
    diff = (a_ - b_).abs()
    ulp = b_.abs().clamp_min(2.0**-126) * 2.0**-7     # one bf16 step at the reference's magnitude (8 significant bits)
    assert (diff <= ulp).all(), f"max diff {diff.max().item():.4g} beyond one bf16 ulp"
    assert (diff > 0).float().mean().item() < 2e-2, "more than 2 % of the elements differ: not a summation-order effect"
    untouched = [s_ for s_ in range(nring) if s_ not in {(slot0 + t) % nring for t in range(T)}]
    assert all((fused_ring[s_] == 3.0).all() for s_ in untouched)


def test_fused_norm_equals_separate_norm_kernels():
    """fvk_vae_conv_norm_bf16 (RMS-norm + SiLU of the 96- and 192-channel stages in the producing conv's epilogue) vs the separate norm kernel: same
    arithmetic on the same bf16-rounded conv output; only the fp32 summation order of ||x||^2 differs, so pixels agree to float rounding."""
    from fastvideo_amd.wan_vae import WanVaeDecoderHip
    from fastvideo_amd.wan_config import wan_vae_param_spec
    g = torch.Generator().manual_seed(0)
    sd = {}
    for n, shp in wan_vae_param_spec(base_dim=96):
        fan_in = 1
        for d in shp[1:]:
            fan_in *= d
        if "gamma" in n:
            sd[n] = torch.ones(shp) + 0.05 * torch.randn(shp, generator=g)
        elif len(shp) >= 4:
            sd[n] = (torch.rand(shp, generator=g) * 2 - 1) * (3.0 / fan_in)**0.5
        else:
            sd[n] = 0.02 * torch.randn(shp, generator=g)
    z = torch.randn(1, 16, 3, 12, 20, generator=g).cuda()
    outs = []
    for fuse in (True, False):
        dec = WanVaeDecoderHip(sd, fuse_norm=fuse)
        outs.append(dec.decode(z).float().cpu())
    assert outs[0].shape == (1, 3, 9, 96, 160)
    err = (outs[0] - outs[1]).abs()
    print(f"fused vs separate norm: max {err.max().item():.3g} mean {err.mean().item():.3g}")
    # bf16 one-ulp flips (test_conv_norm_epilogue_vs_conv_then_norm_kernel bounds each fused site to that) propagate through the six
    # 96-channel and seven 192-channel convs downstream of a fused norm: measured max 2.7e-2, mean 3.1e-3 (1.0e-3 with the 96 stage alone)
    assert err.max().item() < 4e-2 and err.mean().item() < 6e-3


def test_frames_per_pass_is_bit_identical_to_the_reference_frame_by_frame_walk():
    """The cached decode walks F latent frames per decoder pass after the first (default 4) where the reference walks one
    (wanvae.py:1222-1233).  Every conv reads its causal history from a ring, so each output element sees the same operands in the same
    order whatever F is: F = 1 (the reference's literal walk), 2, 3 (ragged last group), 4 and 7 must agree bit for bit — on the real
    decoder widths (base_dim 96: the fused-norm epilogue, the 384-wide mid attention and both time-upsampling convs are on the path)."""
    from fastvideo_amd.wan_vae import WanVaeDecoderHip
    from fastvideo_amd.wan_config import wan_vae_param_spec
    g = torch.Generator().manual_seed(3)
    sd = {}
    for n, shp in wan_vae_param_spec(base_dim=96):
        fan_in = 1
        for d in shp[1:]:
            fan_in *= d
        if "gamma" in n:
            sd[n] = torch.ones(shp) + 0.05 * torch.randn(shp, generator=g)
        elif len(shp) >= 4:
            sd[n] = (torch.rand(shp, generator=g) * 2 - 1) * (3.0 / fan_in)**0.5
        else:
            sd[n] = 0.02 * torch.randn(shp, generator=g)
    z = torch.randn(1, 16, 6, 8, 12, generator=g).cuda()
    ref = WanVaeDecoderHip(sd, frames_per_pass=1).decode(z)
    assert ref.shape == (1, 3, 21, 64, 96)
    for F in (2, 3, 4, 7):
        y = WanVaeDecoderHip(sd, frames_per_pass=F).decode(z)
        assert torch.equal(y, ref), f"frames_per_pass={F}: {(y != ref).sum().item()} elements differ"
    with pytest.raises(ValueError):
        WanVaeDecoderHip(sd, frames_per_pass=0)
    # a ring that would outgrow one 32-bit buffer descriptor lowers F instead of reaching the kernel's refusal
    dec = WanVaeDecoderHip(sd, frames_per_pass=4)
    assert dec._fit_frames_per_pass(8, 12) == 4
    widest = lambda F: max((t + 2) * h * w * c * 2 for _, t, h, w, c, _ in dec._site_shapes(8, 12, F))
    dec.RING_BYTES_MAX = widest(3)  # F = 3's widest ring is exactly one byte too many
    assert dec._fit_frames_per_pass(8, 12) == 2
    assert torch.equal(dec.decode(z), ref)
    assert WanVaeDecoderHip(sd, frames_per_pass=4)._fit_frames_per_pass(135, 240) == 2  # 1080p: 4 F + 2 frames of 1080 x 1920 x 96 bf16
