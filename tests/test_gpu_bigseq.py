"""Parity at the REAL token counts of BASELINE configs 4 and 5 (VERDICT r5 missing #1): one transformer block of

  * cfg4 — Wan2.2-T2V-A14B geometry (40 heads x 128 = 5120, ffn 13 824) on latent [1,16,21,90,160] = 75 600 tokens, dense attention: fused-QKV
    row stride 15 360, a (batch, head) K slice of 2.32 GB behind one 32-bit buffer descriptor (attn_w16.hip / attn_w64.hip: k_rsrc);
  * the same block through ``SequenceParallel`` on EIGHT processes sharing the GPU — the per-rank shapes of BASELINE config 4's SP = 8 run
    (5 heads x 75 600 keys, 9 450 query rows per rank, the 256 x 128-tile GEMMs) — bit-identical to SP = 1;
  * cfg5 — Wan2.1-1.3B geometry (12 heads, ffn 8960) on latent [1,16,33,90,160] = 118 800 tokens: dense, video-sparse attention (2 160 blocks
    of 64, top-432, S_pad 138 240) and video-sparse attention with fp8_channel linears (what BASELINE config 5 names)

against ``oracle.WanOracle`` (ref: fastvideo/models/dits/wanvideo.py:361-434,520-582,656-766; restatement pinned bit-exactly to the imported
reference at small geometry, tests/test_oracle_golden.py).

How the CPU cost stays inside a test: exact attention over 75 600 / 118 800 keys for EVERY query row is 1e14 FLOP per block on the host.  The
oracle therefore evaluates its attention sub-block on a SAMPLE of query rows / query blocks (first and last rows, rows either side of every
2^k boundary, rows next to the places where a 32-bit byte offset of the fused-QKV buffer would wrap, random rows; ALL keys for each of them)
and the device's self-attention output is compared there; for the rest of the block the oracle continues from the DEVICE's whole attention
output, so that every later tensor (residual stream after self-attention, block output, final norm, model output) is compared WHOLE, at the
reference's DiT bound (atol 1e-1, rtol 1e-2, fastvideo/tests/transformers/test_wanvideo.py:109).  What is sampled is therefore exactly one op
(softmax(QK^T)V rows — each sampled row still crosses the whole key axis); everything else is whole-tensor."""
import os
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import _guard  # noqa: E402
import _mp  # noqa: E402
_guard.maybe_install()   # (spawned workers import this module: FVK_GUARD_ALLOC=1 reaches them too)

CFG4_LATENT = (1, 16, 21, 90, 160)   # 21 x 45 x 80 = 75 600 tokens
CFG5_LATENT = (1, 16, 33, 90, 160)   # 33 x 45 x 80 = 118 800 tokens


def _cmp(y, ref, what, atol=1e-1, rtol=1e-2, mean_tol=1.5e-2):
    y, ref = y.float().cpu(), ref.float()
    assert torch.isfinite(y).all(), what
    err = (y - ref).abs()
    bad = err > atol + rtol * ref.abs()
    print(f"{what}: max|err|={err.max().item():.4g} mean|err|={err.mean().item():.4g} ref_absmean={ref.abs().mean().item():.4g}")
    assert not bad.any(), f"{what}: {int(bad.sum())}/{bad.numel()} outside tolerance, max {err.max().item():.4g}"
    assert err.mean().item() < mean_tol, f"{what}: mean error {err.mean().item():.4g}"


def _cmp_attn_rows(dev, ref, what):
    """Sampled self-attention rows, device (bf16) vs exact fp32 softmax attention of the ORACLE's q / k / v over all keys: the kernel-level bound
    of tests/test_gpu_kernels.py (mean |err| < 3e-3 x mean |ref| + 2e-5) although the two sides' q / k differ by the bf16 rounding of the
    projections upstream (measured in round 6: 1.4e-3 x mean |ref| at both token counts); the maximum within four bf16 ulps of the largest
    output (a bf16 ulp is 2^-8 of the value; measured 0.3 ulp) and inside the reference's attention threshold 4e-2
    (fastvideo-kernel/tests/test_sta.py:88-91)."""
    dev, ref = dev.float().cpu(), ref.float()
    assert torch.isfinite(dev).all(), what
    err = (dev - ref).abs()
    print(f"{what}: {tuple(ref.shape)} max|err|={err.max().item():.4g} mean|err|={err.mean().item():.4g} "
          f"ref_absmean={ref.abs().mean().item():.4g} ref_absmax={ref.abs().max().item():.4g}")
    assert err.mean().item() < 3e-3 * ref.abs().mean().item() + 2e-5, f"{what}: mean error {err.mean().item():.4g}"
    assert err.max().item() < min(4e-2, 1.6e-2 * ref.abs().max().item() + 1e-3), f"{what}: max error {err.max().item():.4g}"


def _sample_rows(S, row_stride_elems, n_random=256, seed=0):
    """Query rows the exact attention is evaluated on: both ends, either side of every power of two and of every row at which a byte offset
    ``row * row_stride_elems * 2`` crosses a multiple of 2^31 (signed 32-bit) or 2^32 (unsigned 32-bit), plus random rows."""
    rows = set(range(64)) | set(range(S - 64, S))
    p = 256
    while p < S:
        rows |= {p - 1, p, p + 1}
        p *= 2
    for lim in (1 << 31, 1 << 32):
        m = 1
        while m * lim // (row_stride_elems * 2) < S:
            r = m * lim // (row_stride_elems * 2)
            rows |= {r - 1, r, r + 1}
            m += 1
    g = np.random.default_rng(seed)
    rows |= set(int(r) for r in g.integers(0, S, n_random))
    return torch.tensor(sorted(r for r in rows if 0 <= r < S))


def _exact_attention_rows(q, k, v, rows, scale):
    """softmax(q[rows] k^T * scale) v in fp32, one head at a time.  q, k, v [1,S,H,D] -> [len(rows), H, D] fp32."""
    H = q.shape[2]
    out = torch.empty((rows.numel(), H, q.shape[3]), dtype=torch.float32)
    for h in range(H):
        s = (q[0, rows, h].float() @ k[0, :, h].float().T) * scale
        out[:, h] = torch.softmax(s, dim=-1) @ v[0, :, h].float()
    return out


def _widen_tables(sd, seed):
    gen = torch.Generator().manual_seed(seed)
    for k in [k for k in sd if k.endswith("scale_shift_table")]:   # so that shift / scale / gate matter
        sd[k] = (torch.randn(sd[k].shape, generator=gen) * 0.3).to(sd[k].dtype)


def _inputs(cfg, latent_shape, seed):
    gen = torch.Generator().manual_seed(seed)
    latent = torch.randn(latent_shape, generator=gen).bfloat16()
    ctx = torch.randn((1, 512, cfg.text_dim), generator=gen).bfloat16()
    return latent, ctx, torch.tensor([500.0])


def _a14b_block():
    from fastvideo_amd import wan_config as WC
    cfg = WC.WanConfig("Wan2.2-A14B geometry, 1 layer", 40, 128, 13824, 1)
    sd = WC.random_state_dict(cfg, seed=3, device="cpu")
    _widen_tables(sd, 7)
    return cfg, sd


def test_one_a14b_block_at_cfg4_tokens_matches_oracle():
    """BASELINE config 4's DiT shapes on ONE GPU: S = 75 600, 40 heads, d = 5120, ffn 13 824, dense attention (module docstring)."""
    from fastvideo_amd.wan_dit import WanTransformer3DModelHip
    from oracle import wan_oracle as W
    cfg, sd = _a14b_block()
    latent, ctx, t = _inputs(cfg, CFG4_LATENT, 1)
    S, H, D = 21 * 45 * 80, cfg.num_heads, cfg.head_dim
    torch.set_num_threads(min(64, os.cpu_count() or 8))
    model = WanTransformer3DModelHip(sd, cfg.num_heads, cfg.head_dim, cfg.patch_size, cfg.eps, cfg.freq_dim)
    tr = {}
    y = model(latent.cuda(), ctx.cuda(), t.cuda(), trace=tr)
    assert model.dense_kernel_ran is not None
    print("dense self-attention kernel:", model.dense_kernel_ran)
    dev_attn = tr["blocks.0.attn"].cpu().view(1, S, H, D)
    rows = _sample_rows(S, 3 * cfg.dim)   # the q | k projection is [S, 2d] on the V^T-GEMM path, [S, 3d] on the fused one: both strides below
    rows = torch.unique(torch.cat([rows, _sample_rows(S, 2 * cfg.dim, n_random=0)]))
    seen = {}

    def attention(q, k, v, scale):
        seen["rows"] = _exact_attention_rows(q, k, v, rows, scale)
        return dev_attn.to(q.dtype)

    orc = W.WanOracle(sd, num_heads=cfg.num_heads)
    orc.attention = attention
    tr_ref = {}
    with torch.no_grad():
        y_ref = orc.forward(latent, ctx, t, trace=tr_ref)
    _cmp_attn_rows(dev_attn[0, rows], seen["rows"], f"cfg4 self-attention, {rows.numel()} sampled query rows x 75 600 keys x 40 heads")
    for key in ("blocks.0.after_self_attn", "blocks.0.out", "norm_out"):
        _cmp(tr[key], tr_ref[key], f"cfg4 {key}")
    _cmp(y, y_ref, "cfg4 model output")
    # the other selectable long-key kernel (attn_autotune may keep either): same rows, same bound
    other = 2 if model.attn_kernel != 2 else 1
    model.attn_kernel = other
    tr2 = {}
    model(latent.cuda(), ctx.cuda(), t.cuda(), trace=tr2)
    print("second dense self-attention kernel:", model.dense_kernel_ran)
    _cmp_attn_rows(tr2["blocks.0.attn"].cpu().view(1, S, H, D)[0, rows], seen["rows"], "cfg4 self-attention, the other long-key kernel")


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _sp_worker(rank, world, port, out_q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from fastvideo_amd.wan_dit import WanTransformer3DModelHip
        cfg, sd = _a14b_block()
        latent, ctx, t = _inputs(cfg, CFG4_LATENT, 1)
        model = WanTransformer3DModelHip(sd, cfg.num_heads, cfg.head_dim, cfg.patch_size, cfg.eps, cfg.freq_dim, device="cuda:0")
        y = model(latent.cuda(), ctx.cuda(), t.cuda()).cpu()
        if rank == 0:
            out_q.put(_mp.ship((y, (model.sp.lay.G, model.sp.lay.U), model.sp.overlap)))
            out_q.close(); out_q.join_thread()
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_cfg4_block_on_eight_ranks_sharing_the_gpu_equals_sp1():
    """BASELINE config 4 IS an SP = 8 run: 40 heads / 8 = 5 heads per rank over all 75 600 keys, 9 450 query rows per rank, N = 5120 / 13 824
    GEMMs on 9 450 rows.  Eight processes share the one GPU and exchange through gloo (host-staged; RCCL on a real node): the forward must
    equal SP = 1 bit for bit (ref: the reference asserts the same for its SP path, fastvideo/tests/distributed/test_sp_wan.py:198-281), and
    SP = 1 is checked against the oracle by the test above."""
    import torch.multiprocessing as mp
    from fastvideo_amd.wan_dit import WanTransformer3DModelHip
    cfg, sd = _a14b_block()
    latent, ctx, t = _inputs(cfg, CFG4_LATENT, 1)
    model = WanTransformer3DModelHip(sd, cfg.num_heads, cfg.head_dim, cfg.patch_size, cfg.eps, cfg.freq_dim)
    ref = model(latent.cuda(), ctx.cuda(), t.cuda()).cpu()
    del model
    torch.cuda.empty_cache()
    world = 8
    mpc = mp.get_context("spawn")
    out_q = mpc.Queue()
    port = _free_port()
    procs = [mpc.Process(target=_sp_worker, args=(r, world, port, out_q)) for r in range(world)]
    for p in procs:
        p.start()
    y, (G, U), overlap = _mp.unship(out_q.get(timeout=900))
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    assert (G, U) == (8, 1) and not overlap
    assert torch.isfinite(y.float()).all()
    assert torch.equal(y, ref), (f"SP = 8 at cfg4 shapes: {int((y != ref).sum())} of {ref.numel()} elements differ from SP = 1, "
                                 f"max {(y.float() - ref.float()).abs().max().item():.4g}")


def _block_1_3b(with_gate):
    from fastvideo_amd import wan_config as WC
    cfg = WC.WanConfig("Wan2.1-T2V-1.3B geometry, 1 layer", 12, 128, 8960, 1)
    sd = WC.random_state_dict(cfg, seed=0, device="cpu", with_vsa_gate=with_gate)
    _widen_tables(sd, 5)
    return cfg, sd


def test_one_block_at_cfg5_tokens_dense_matches_oracle():
    """BASELINE config 5's token count, dense attention: S = 118 800 (the largest key axis any config asks of the dense kernels)."""
    from fastvideo_amd.wan_dit import WanTransformer3DModelHip
    from oracle import wan_oracle as W
    cfg, sd = _block_1_3b(False)
    latent, ctx, t = _inputs(cfg, CFG5_LATENT, 2)
    S, H, D = 33 * 45 * 80, cfg.num_heads, cfg.head_dim
    torch.set_num_threads(min(64, os.cpu_count() or 8))
    model = WanTransformer3DModelHip(sd, cfg.num_heads, cfg.head_dim, cfg.patch_size, cfg.eps, cfg.freq_dim)
    tr = {}
    y = model(latent.cuda(), ctx.cuda(), t.cuda(), trace=tr)
    dev_attn = tr["blocks.0.attn"].cpu().view(1, S, H, D)
    rows = torch.unique(torch.cat([_sample_rows(S, 3 * cfg.dim), _sample_rows(S, 2 * cfg.dim, n_random=0)]))
    seen = {}

    def attention(q, k, v, scale):
        seen["rows"] = _exact_attention_rows(q, k, v, rows, scale)
        return dev_attn.to(q.dtype)

    orc = W.WanOracle(sd, num_heads=cfg.num_heads)
    orc.attention = attention
    tr_ref = {}
    with torch.no_grad():
        y_ref = orc.forward(latent, ctx, t, trace=tr_ref)
    _cmp_attn_rows(dev_attn[0, rows], seen["rows"], f"cfg5 dense self-attention, {rows.numel()} sampled query rows x 118 800 keys x 12 heads")
    for key in ("blocks.0.after_self_attn", "blocks.0.out", "norm_out"):
        _cmp(tr[key], tr_ref[key], f"cfg5 dense {key}")
    _cmp(y, y_ref, "cfg5 dense model output")


def _vsa_hook(V, md, vbs, topk, mask, dev_attn, q_blocks, seen):
    """The oracle's ``video_sparse_attn`` (fastvideo_kernel/ops.py:65-133 restated, oracle/vsa_oracle.py) with the DEVICE's block selection:
    coarse branch whole, sparse branch on the sampled query blocks; returns the device's attention output for the rest of the block."""
    def attention(q, k, v, scale, gate):
        B, S, H, D = q.shape
        tq, tk, tv, tg = (V.tile(x_, md).transpose(1, 2).contiguous() for x_ in (q, k, v, gate))
        o, info = V.video_sparse_attn(tq, tk, tv, vbs, vbs, topk, 64, tg, mask_override=mask, gathered=True, q_blocks=q_blocks)
        own = V.topk_mask_bisect(info["scores"].float().numpy(), topk)
        seen["agree"] = float((own == mask).mean())
        rows = torch.cat([torch.arange(64 * b_, 64 * b_ + 64) for b_ in q_blocks])
        seen["ref"] = o[:, :, rows].float()                                            # [B,H,rows,D]
        seen["dev"] = V.tile(dev_attn.view(B, S, H, D), md).transpose(1, 2)[:, :, rows].float()
        # tile-major pad rows of a block hold no token: the device writes token order only -> compare real rows only
        real = torch.cat([torch.arange(64) < int(vbs[b_]) for b_ in q_blocks])
        seen["ref"], seen["dev"] = seen["ref"][:, :, real], seen["dev"][:, :, real]
        return dev_attn.view(B, S, H, D).to(q.dtype)
    return attention


def _cfg5_vsa_setup(quant):
    from fastvideo_amd.wan_dit import WanTransformer3DModelHip
    from oracle import vsa_oracle as V
    cfg, sd = _block_1_3b(True)
    latent, ctx, t = _inputs(cfg, CFG5_LATENT, 2)
    torch.set_num_threads(min(64, os.cpu_count() or 8))
    model = WanTransformer3DModelHip(sd, cfg.num_heads, cfg.head_dim, cfg.patch_size, cfg.eps, cfg.freq_dim, attention="vsa", vsa_sparsity=0.8,
                                     quantization=quant)
    assert model.vsa_fold and model.vsa_gate
    model.vsa_trace = []
    tr = {}
    y = model(latent.cuda(), ctx.cuda(), t.cuda(), trace=tr)
    mask, model.vsa_trace = model.vsa_trace[0].cpu().numpy(), None
    md = V.build_metadata(tuple(latent.shape[2:]))
    vbs = md["variable_block_sizes"]
    topk = V.compute_topk(0.8, len(vbs))
    assert len(vbs) == 2160 and topk == 432 and mask.shape == (1, 12, 2160, 2160) and (mask.sum(-1) == topk).all()
    g = np.random.default_rng(3)
    ragged = [int(b_) for b_ in np.nonzero(vbs < 64)[0]]
    q_blocks = sorted({0, 1, 2159, 2158, ragged[0], ragged[len(ragged) // 2], ragged[-1], *[int(b_) for b_ in g.integers(0, 2160, 9)]})
    return cfg, sd, latent, ctx, t, model, tr, y, mask, md, vbs, topk, q_blocks, V


def test_one_block_at_cfg5_tokens_vsa_matches_oracle():
    """BASELINE config 5's attention at its real geometry: (9,12,20) tiles of 4x4x4 = 2 160 blocks of 64 rows (S_pad 138 240), sparsity 0.8 ->
    top-432, compress gate, the gather-free model path — block selection compared with the oracle's own top-k, the sparse branch on sampled
    query blocks (first, last, ragged, random), everything after the attention whole."""
    from oracle import wan_oracle as W
    cfg, sd, latent, ctx, t, model, tr, y, mask, md, vbs, topk, q_blocks, V = _cfg5_vsa_setup(None)
    S, H, D = 33 * 45 * 80, cfg.num_heads, cfg.head_dim
    dev_attn = tr["blocks.0.attn"].cpu()
    seen = {}
    orc = W.WanOracle(sd, num_heads=cfg.num_heads)
    orc.attention, orc.vsa_gate = _vsa_hook(V, md, vbs, topk, mask, dev_attn, q_blocks, seen), True
    tr_ref = {}
    with torch.no_grad():
        y_ref = orc.forward(latent, ctx, t, trace=tr_ref)
    print(f"cfg5 vsa block selection: oracle's own top-432 agrees with the device's on {seen['agree']:.6f} of the 12 x 2160 x 2160 entries")
    assert seen["agree"] > 0.999
    _cmp_attn_rows(seen["dev"], seen["ref"], f"cfg5 video-sparse attention, query blocks {q_blocks}")
    for key in ("blocks.0.after_self_attn", "blocks.0.out", "norm_out"):
        _cmp(tr[key], tr_ref[key], f"cfg5 vsa {key}")
    _cmp(y, y_ref, "cfg5 vsa model output")


def test_one_block_at_cfg5_tokens_vsa_fp8_channel_matches_oracle():
    """What BASELINE config 5 names — fp8 linears + video-sparse attention at 118 800 tokens — with per-token fp8 scales (fp8_config.py:55-68,
    119-157).  Bound: tests/test_gpu_fullgeom.py::_cmp_fp8 (the device must be closer to the fp8 oracle than the fp8 oracle is to the bf16 oracle),
    both oracles continuing from the device's attention output; the sparse branch on sampled query blocks against the fp8 oracle's q / k / v
    inside the same kind of bound."""
    from oracle import wan_oracle as W
    from test_gpu_fullgeom import _cmp_fp8
    cfg, sd, latent, ctx, t, model, tr, y, mask, md, vbs, topk, q_blocks, V = _cfg5_vsa_setup("fp8_channel")
    dev_attn = tr["blocks.0.attn"].cpu()
    seen_q, seen_b = {}, {}
    tr_q, tr_b = {}, {}
    with torch.no_grad():
        orc = W.WanOracle(sd, num_heads=cfg.num_heads, quantization="fp8_channel")
        orc.attention, orc.vsa_gate = _vsa_hook(V, md, vbs, topk, mask, dev_attn, q_blocks, seen_q), True
        y_q = orc.forward(latent, ctx, t, trace=tr_q)
        orc = W.WanOracle(sd, num_heads=cfg.num_heads)
        orc.attention, orc.vsa_gate = _vsa_hook(V, md, vbs, topk, mask, dev_attn, q_blocks, seen_b), True
        y_b = orc.forward(latent, ctx, t, trace=tr_b)
    print(f"cfg5 vsa fp8_channel block selection agreement with the fp8 oracle's own top-432: {seen_q['agree']:.6f}")
    assert seen_q["agree"] > 0.99
    # sampled sparse-attention rows: device vs the fp8 oracle, against the fp8 oracle vs the bf16 oracle (same selection in all three)
    err = (seen_q["dev"] - seen_q["ref"]).abs()
    dq = (seen_q["ref"] - seen_b["ref"]).abs()
    print(f"cfg5 vsa fp8_channel attention rows: device vs fp8 oracle mean {err.mean().item():.4g} max {err.max().item():.4g} | "
          f"fp8 oracle vs bf16 oracle mean {dq.mean().item():.4g} max {dq.max().item():.4g}")
    assert torch.isfinite(seen_q["dev"]).all()
    assert err.mean().item() <= dq.mean().item() and err.max().item() <= max(dq.max().item(), 4e-2)
    for key in ("blocks.0.after_self_attn", "blocks.0.out", "norm_out"):
        _cmp_fp8(tr[key], tr_q[key], tr_b[key], f"cfg5 vsa fp8_channel {key}")
    _cmp_fp8(y, y_q, y_b, "cfg5 vsa fp8_channel model output")
