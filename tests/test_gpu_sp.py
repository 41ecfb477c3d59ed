"""Full HIP DiT forward under sequence parallelism on ONE GPU: `world` processes share cuda:0 and exchange through gloo
(host-staged; the production backend is RCCL).  SP=2 (plain Ulysses) and SP=4 with 2 heads (2-D Ulysses, U=2) must equal
the SP=1 forward bit for bit: every kernel is row-independent and the KV tile order does not depend on the partition
(the reference asserts the same property for its SP path, fastvideo/tests/distributed/test_sp_wan.py:198-281)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

import _guard  # noqa: E402
import _mp  # noqa: E402
_guard.maybe_install()   # (spawned workers import this module: FVK_GUARD_ALLOC=1 reaches them too)


# (Round 5 had an `autouse` half-second pause here after one unexplained GPU fault in a parent-process forward.  Round 6 found the cause with the
# guard-page allocator — an out-of-bounds read of one gate row by the last layer's gated-residual GEMM, profiles/r06d_guard_page_runs.md — and fixed
# it; the pause is gone.  This file runs clean under FVK_GUARD_ALLOC=1.)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, fx_path, out_q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from fastvideo_amd.wan_dit import WanTransformer3DModelHip
        fx = torch.load(fx_path, weights_only=False)
        model = WanTransformer3DModelHip(fx["state_dict"], num_heads=fx["config"]["num_heads"], device="cuda:0")
        outs = []
        for case in fx["cases"]:
            outs.append(model(case["latent"].cuda(), case["ctx"].cuda(), case["timestep"].cuda()).cpu())
        if rank == 0:
            out_q.put(_mp.ship((outs, (model.sp.lay.G, model.sp.lay.U))))
            out_q.close(); out_q.join_thread()  # flush before teardown: a crash in runtime shutdown must not truncate the message
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_sp_forward_equals_sp1(world, golden_dir):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    fx_path = os.path.join(golden_dir, "wan_tiny.pt")
    from fastvideo_amd.wan_dit import WanTransformer3DModelHip
    fx = torch.load(fx_path, weights_only=False)
    model = WanTransformer3DModelHip(fx["state_dict"], num_heads=fx["config"]["num_heads"])
    ref = [model(c["latent"].cuda(), c["ctx"].cuda(), c["timestep"].cuda()).cpu() for c in fx["cases"]]
    ctx = mp.get_context("spawn")
    out_q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, fx_path, out_q)) for r in range(world)]
    for p in procs:
        p.start()
    outs, (G, U) = _mp.unship(out_q.get(timeout=300))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert G * U == world and (U == 2 if world == 4 else U == 1)
    for o, r, c in zip(outs, ref, fx["cases"]):
        assert torch.equal(o, r), f"SP={world}: max diff {(o.float() - r.float()).abs().max().item()}"
        err = (o.float() - c["out"].float()).abs()
        assert err.max().item() < 0.1


def _pipelined_model_forward(overlap_expected, heads=6, lat_shape=(1, 16, 5, 18, 30)):
    """A 2-layer model on seeded weights (CPU generator: identical in every process), ragged token count; returns (y, sp.overlap, checked)."""
    from fastvideo_amd import wan_config as WC
    from fastvideo_amd.wan_dit import WanTransformer3DModelHip
    cfg = WC.WanConfig("sp-pipe", heads, 128, 768, 2, text_dim=64)
    sd = WC.random_state_dict(cfg, seed=4, device="cpu")
    model = WanTransformer3DModelHip(sd, cfg.num_heads, device="cuda:0")
    g = torch.Generator().manual_seed(12)
    lat = torch.randn(lat_shape, generator=g).bfloat16()    # default 5 x 9 x 15 = 675 tokens: odd -> one zero-padding row on 2 ranks
    ctx = torch.randn((1, 40, cfg.text_dim), generator=g).bfloat16()
    # Three forwards are returned and ALL must equal SP = 1 bit for bit.  (Round 4: several processes time-slicing one GPU exposed a real bug —
    # a wave of the QK-norm / RoPE pass sharing a SIMD with a gemm_w1 wave of another process got wrong packed-fp32 results, about one forward in
    # four; fixed by FVK_CLAIM_WHOLE_REGISTER_FILE in the one-wave-per-SIMD kernels, DESIGN §5, profiles/r04z_pk_f32_beside_mfma.log.  This test
    # is the regression test of that fix in its natural habitat.)
    ys = [model(lat.cuda(), ctx.cuda(), torch.tensor([333.0]).cuda()).cpu() for _ in range(3)]
    return ys, model.sp.overlap, model.sp._overlap_checked


def _worker_pipelined(rank, world, port, out_q, heads=6, lat_shape=(1, 16, 5, 18, 30), mode="1"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), FVK_SP_OVERLAP=mode)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        out = _pipelined_model_forward(True, heads, lat_shape)
        if rank == 0:
            out_q.put(_mp.ship(out))
            out_q.close(); out_q.join_thread()
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["1", "2"])   # 1: the chunks' attention launches in stream order; 2: chunk B on a second HIP stream
@pytest.mark.parametrize("world,heads,lat_shape", [(2, 6, (1, 16, 5, 18, 30)), (3, 6, (1, 16, 5, 18, 30)), (2, 12, (1, 16, 5, 42, 62))])
def test_sp_pipelined_exchange_equals_sp1(world, heads, lat_shape, mode):
    """FVK_SP_OVERLAP=1 on the device path (round 4): the QK-norm / RoPE pass writes TWO head-chunk send buffers (fvk_qkv_norm_rope_pack2_bf16),
    both exchanges are issued up front, the output exchanges follow their chunks
    (fastvideo_amd/distributed.py: attention_packed_pipelined).  6 heads: world 2 -> 3 heads per group (chunks of 2 + 1), world 3 -> 2 per group.
    The third case is the bench's shape class: 12 heads on 2 ranks (6 per group: chunks of 3 + 3) on a 3 255-token latent — long key axes, i.e. the
    one-wave-per-SIMD attention kernels.  The forward must equal SP = 1 bit for bit, and the mode must have survived its own first-call check
    against the plain exchange."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    refs, ov, _ = _pipelined_model_forward(False, heads, lat_shape)
    ref = refs[0]
    assert not ov and all(torch.equal(ref, r) for r in refs[1:])
    ctx = mp.get_context("spawn")
    out_q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_pipelined, args=(r, world, port, out_q, heads, lat_shape, mode)) for r in range(world)]
    for p in procs:
        p.start()
    out, overlap_kept, checked = _mp.unship(out_q.get(timeout=300))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert checked and overlap_kept, "the pipelined exchange disagreed with the plain exchange on its first call and was switched off"
    for i, o in enumerate(out):
        assert torch.equal(o, ref), (f"pipelined (mode {mode}) SP={world}, forward {i}: {int((o != ref).sum())} of {ref.numel()} elements differ from SP = 1, "
                                     f"max {(o.float() - ref.float()).abs().max().item():.4g}")


def _worker_sparse(rank, world, port, fx_path, mode, out_q, quant=None):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        out = _sparse_forward(fx_path, mode, quant)
        if rank == 0:
            out_q.put(_mp.ship(out))
            out_q.close(); out_q.join_thread()
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _sparse_forward(fx_path, mode, quant=None):
    from fastvideo_amd.wan_dit import WanTransformer3DModelHip
    if quant:
        # poison the caching allocator: the torch.empty buffers of the forward come out of this freed block, so an output row that no
        # kernel writes holds NaN instead of whatever a fresh process happens to find there
        torch.full((96 << 20,), float("nan"), dtype=torch.bfloat16, device="cuda")
        torch.cuda.synchronize()
    fx = torch.load(fx_path, weights_only=False)
    sd = dict(fx["state_dict"])
    d = fx["config"]["num_heads"] * 128
    g = torch.Generator().manual_seed(5)
    for i in range(fx["config"]["num_layers"]):  # VSA compress gate weights (absent from the fixture)
        sd[f"blocks.{i}.to_gate_compress.weight"] = (torch.randn((d, d), generator=g) * d**-0.5).bfloat16()
        sd[f"blocks.{i}.to_gate_compress.bias"] = (torch.randn((d,), generator=g) * 0.02).bfloat16()
    kw = dict(attention="vsa", vsa_sparsity=0.5) if mode == "vsa" else dict(attention="sta", sta_window=(1, 3, 1), sta_tile=(2, 4, 8))
    model = WanTransformer3DModelHip(sd, num_heads=fx["config"]["num_heads"], device="cuda:0", quantization=quant, **kw)
    lat = torch.randn((1, 16, 7, 18, 34), generator=torch.Generator().manual_seed(11)).bfloat16()
    c = fx["cases"][0]
    return model(lat.cuda(), c["ctx"].cuda(), c["timestep"].cuda()).cpu()


@pytest.mark.parametrize("mode,world", [("vsa", 2), ("sta", 2), ("vsa", 4)])
def test_sp_sparse_attention_equals_sp1(mode, world, golden_dir):
    """VSA (its compress gate travelling through exchange #1 as a fourth slot of the packed message row) and sliding-tile attention
    under sequence parallelism == SP=1, bit for bit.  world 2: plain Ulysses (2 heads -> G2 x U1).  world 4 with 2 heads: G2 x U2 — the
    2-D grid of FastWan-1.3B on 8 GPUs (12 heads -> G4 x U2) that round 2 refused: every rank computes its run of tile-major query
    blocks for its head group and the uneven output exchange returns rows to the shard owners (fastvideo_amd/distributed.py: block_plan)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    fx_path = os.path.join(golden_dir, "wan_tiny.pt")
    ref = _sparse_forward(fx_path, mode)
    ctx = mp.get_context("spawn")
    out_q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_sparse, args=(r, world, port, fx_path, mode, out_q)) for r in range(world)]
    for p in procs:
        p.start()
    out = _mp.unship(out_q.get(timeout=300))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert torch.isfinite(out.float()).all()
    assert torch.equal(out, ref), f"{mode} SP={world}: max diff {(out.float() - ref.float()).abs().max().item()}"


@pytest.mark.parametrize("mode", ["vsa", "sta"])
def test_sp_sparse_fp8_ragged_pad_rows_are_finite(mode, golden_dir):
    """Tensor-wise fp8 linears + sparse attention + sequence parallelism on a ragged token count (1071 tokens on 2 ranks: one zero-padding
    row).  The scattering attention epilogues never write the padding rows; exchange #2 ships them into the last shard's o-projection,
    whose per-tensor activation scale is an absmax over ALL rows — so they must be zeroed, not left as allocator garbage (round-3 advisor
    finding).  The allocator is poisoned with NaN first; the result must be finite and within fp8 noise of the SP=1 fp8 forward (the
    per-tensor scales are per rank, so not bit-equal)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    fx_path = os.path.join(golden_dir, "wan_tiny.pt")
    ref = _sparse_forward(fx_path, mode, "fp8")
    ctx = mp.get_context("spawn")
    out_q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_sparse, args=(r, 2, port, fx_path, mode, out_q, "fp8")) for r in range(2)]
    for p in procs:
        p.start()
    out = _mp.unship(out_q.get(timeout=300))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert torch.isfinite(out.float()).all(), "padding rows leaked uninitialised memory into the fp8 activation scale"
    err = (out.float() - ref.float()).abs()
    assert err.mean().item() < 0.05 * ref.float().abs().mean().item() + 1e-3, f"{mode}: mean err {err.mean().item()} vs |ref| {ref.float().abs().mean().item()}"


def test_sta_refuses_u_gt_1(golden_dir):
    """Sliding-tile attention on a G x U grid with U > 1 is not built: refused at construction with the reason (VSA and dense run there)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from fastvideo_amd.distributed import SPLayout
    from fastvideo_amd.wan_dit import WanTransformer3DModelHip
    fx = torch.load(os.path.join(golden_dir, "wan_tiny.pt"), weights_only=False)
    m = WanTransformer3DModelHip(fx["state_dict"], num_heads=fx["config"]["num_heads"], attention="sta")
    m.sp.lay = SPLayout(P=4, rank=1, H=2, G=2, U=2)
    with pytest.raises(NotImplementedError, match="sliding-tile"):
        m._sp_plan((7, 9, 17), 300)
