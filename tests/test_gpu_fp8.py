"""GPU parity of the fp8 (OCP e4m3fn) linear path against oracle/fp8_oracle.py (pinned to the reference's fp8_config.py).
Quantisation (bytes and scales) is an integer-like construction: bit-exact.  The GEMM is compared at atol = rtol = 1e-2 on bf16
outputs (the fused-op tolerance of fastvideo-kernel/tests/test_turbodiffusion.py:143)."""
import os
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "fp8_quant.pt")
DEV = "cuda"


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from fastvideo_amd import ops as o
    return o


def rnd(shape, seed, scale=1.0):
    return torch.randn(shape, generator=torch.Generator().manual_seed(seed)) * scale


@pytest.mark.parametrize("rowwise", [False, True])
def test_quantize_matches_reference_golden_bit_exact(ops, rowwise):
    g = torch.load(GOLD)
    q, s = ops.fp8_quantize(g["x"].cuda(), rowwise=rowwise)
    assert torch.equal(q.cpu().view(torch.uint8), g["q_row" if rowwise else "q_tensor"])
    assert torch.equal(s.cpu(), g["s_row" if rowwise else "s_tensor"])


@pytest.mark.parametrize("M,K,rowwise", [(1000, 1536, False), (1000, 1536, True), (32760, 8960, False), (3, 64, True),
                                         # round 6: the one-pass rowwise kernel (a wave keeps its row in registers) at each register tier — up to 2048,
                                         # 9216 and 18432 columns — a ragged last chunk per tier, and past it (the two-kernel form)
                                         (700, 8960, True), (5, 2048, True), (9, 2056, True), (50, 18432, True), (10, 20480, True)])
def test_quantize_bit_exact_vs_oracle(ops, M, K, rowwise):
    from oracle import fp8_oracle as O
    x = (rnd((M, K), 1) * torch.logspace(-2, 1, M).unsqueeze(1)).bfloat16()
    q_ref, s_ref = (O.quantize_rowwise if rowwise else O.quantize_tensorwise)(x)
    q, s = ops.fp8_quantize(x.cuda(), rowwise=rowwise)
    assert torch.equal(s.cpu(), s_ref)
    assert torch.equal(q.cpu().view(torch.uint8), q_ref.view(torch.uint8))
    # strided input (a column block of a wider buffer)
    wide = torch.zeros((M, K + 64), dtype=torch.bfloat16, device="cuda")
    wide[:, :K] = x.cuda()
    q2, s2 = ops.fp8_quantize(wide[:, :K], rowwise=rowwise)
    assert torch.equal(q2, q) and torch.equal(s2, s)


@pytest.mark.parametrize("M,N,K,gran", [(300, 256, 128, "tensor"), (1000, 1536, 1536, "tensor"), (777, 520, 1536, "channel"),
                                        (4096, 4608, 1536, "tensor"), (130, 8960, 1536, "channel")])
def test_fp8_linear_vs_oracle(ops, M, N, K, gran):
    from oracle import fp8_oracle as O
    x = rnd((M, K), 1).bfloat16()
    w = rnd((N, K), 2, K**-0.5).bfloat16()
    b = rnd((N,), 3, 0.1).bfloat16()
    wq, ws = O.quantize_weight(w, gran)
    ref = O.fp8_linear(x, wq, ws, b, gran)
    xq, xs = ops.fp8_quantize(x.cuda(), rowwise=(gran == "channel"))
    wq_g, ws_g = ops.fp8_quantize(w.cuda(), rowwise=(gran == "channel"))          # the weight side uses the same kernel
    assert torch.equal(wq_g.cpu().view(torch.uint8), wq.view(torch.uint8)) and torch.equal(ws_g.cpu().view(-1), ws.view(-1))
    y = ops.gemm_fp8(xq, xs, wq_g, ws_g, b.cuda())
    torch.testing.assert_close(y.float().cpu(), ref.float(), atol=1e-2, rtol=1e-2)
    # and it stays close to the bf16 linear it approximates (sanity of the whole path, loose)
    lin = torch.nn.functional.linear(x.float(), w.float(), b.float())
    assert (y.float().cpu() - lin).abs().mean() < 0.05 * lin.abs().mean() + 1e-2


def test_fp8_gemm_epilogues(ops):
    from oracle import fp8_oracle as O
    M, N, K = 600, 512, 256
    x, w, b = rnd((M, K), 1).bfloat16(), rnd((N, K), 2, K**-0.5).bfloat16(), rnd((N,), 3, 0.1).bfloat16()
    res, gate = rnd((M, N), 4).bfloat16(), rnd((1, N), 5)
    wq, ws = O.quantize_weight(w)
    y = O.fp8_linear(x, wq, ws, b).float()
    xq, xs = ops.fp8_quantize(x.cuda())
    got = ops.gemm_fp8(xq, xs, wq.cuda(), ws.cuda(), b.cuda(), epilogue=ops.EPI_GELU_TANH)
    torch.testing.assert_close(got.float().cpu(), torch.nn.functional.gelu(y, approximate="tanh").bfloat16().float(), atol=1e-2, rtol=1e-2)
    got = ops.gemm_fp8(xq, xs, wq.cuda(), ws.cuda(), b.cuda(), epilogue=ops.EPI_RESIDUAL_GATE, residual=res.cuda(), gate=gate.cuda())
    torch.testing.assert_close(got.float().cpu(), (res.float() + y * gate).bfloat16().float(), atol=2e-2, rtol=1e-2)


def test_fp8_errors_are_loud(ops):
    x = torch.zeros((8, 64), dtype=torch.bfloat16, device="cuda")
    q, s = ops.fp8_quantize(x)
    with pytest.raises(RuntimeError):
        ops.gemm_fp8(q[:, :32].contiguous(), s, q[:, :32].contiguous(), s)   # K % 64 != 0
    with pytest.raises(RuntimeError):
        ops.fp8_quantize(x.float())


@pytest.mark.parametrize("M,d", [(700, 1536), (130, 5120), (64, 256)])
def test_ln_modulate_writes_the_per_token_quantisation(ops, M, d):
    """fvk_ln_modulate_fp8_bf16: the LayerNorm / modulation pass writes the per-token e4m3 quantisation of its bf16 output row — bytes and scales
    identical to fvk_fp8_quantize_bf16(rowwise) applied to that output — in the three forms the DiT block uses (plain modulation; gated residual
    + affine LayerNorm; residual + modulation with the reference's bf16 rounding points), instead of or beside the bf16 output."""
    g = torch.Generator().manual_seed(M + d)
    x, res = (torch.randn((M, d), generator=g) * 3).bfloat16().to(DEV), torch.randn((M, d), generator=g).bfloat16().to(DEV)
    x[5] *= 200.0          # a row whose quotient clamps at 448 nowhere but whose scale differs by orders of magnitude
    x[7] = 0               # an all-zero row: the floor of the scale
    gate, mul, add = (torch.randn((2, d), generator=g).to(DEV) for _ in range(3))
    lw, lb = torch.randn(d, generator=g).to(DEV), torch.randn(d, generator=g).to(DEV)
    rpb = M // 2
    forms = [dict(mul=1 + mul, add=add, rows_per_batch=rpb),
             dict(residual=res, gate=gate, ln_w=lw, ln_b=lb, want_residual=True, rows_per_batch=rpb),
             dict(residual=res, mul=1 + mul, add=add, round_residual=True, round_norm=True, want_residual=True, rows_per_batch=rpb)]
    for kw in forms:
        ref = ops.ln_modulate(x, **kw)
        ref_out, ref_res = ref if kw.get("want_residual") else (ref, None)
        q_ref, s_ref = ops.fp8_quantize(ref_out, rowwise=True)
        for mode in ("only", "both"):
            got = ops.ln_modulate(x, fp8_rowwise=mode, **kw)
            first, got_res = got if kw.get("want_residual") else (got, None)
            out, (q, s) = (None, first) if mode == "only" else first
            assert torch.equal(q.view(torch.uint8), q_ref.view(torch.uint8)) and torch.equal(s, s_ref)
            assert out is None or torch.equal(out, ref_out)
            assert ref_res is None or torch.equal(got_res, ref_res)
    with pytest.raises(ValueError, match="fp8_rowwise"):
        ops.ln_modulate(x, fp8_rowwise="yes")
