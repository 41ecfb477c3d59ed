"""GPU parity of the widened Wan VAE decode: tile cross-fade and uint8 post-processing kernels (bit-exact
vs the oracle: fp32 elementwise work), streaming decode, the cache-less decode family (plain / spatial / temporal /
tile-parallel tiling) vs oracle.vae_oracle.WanVaeTiledOracle, which tests/test_vae_tiled_oracle.py pins bit-exact to the
real reference.  Decode tolerances as tests/test_gpu_vae.py (bf16 activations vs the fp32 reference)."""
import os
import socket

import pytest
import torch

import _mp  # tensors across the queue by value (tests/_mp.py)
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "vae_tiled.pt")


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from fastvideo_amd import ops as o
    return o


def rnd(shape, seed, scale=1.0):
    return torch.randn(shape, generator=torch.Generator().manual_seed(seed)) * scale


def _gold():
    from oracle.vae_oracle import seeded_state_dict
    g = torch.load(GOLD, weights_only=False)
    sd = seeded_state_dict(g["param_spec"], g["seed"])
    z = torch.randn(g["z_shape"], generator=torch.Generator().manual_seed(g["z_seed"]))
    return g, sd, z


def _close(y, y_ref, max_tol=6e-2):
    assert y.shape == y_ref.shape and y.dtype == torch.float32, (y.shape, y_ref.shape)
    err = (y - y_ref).abs()
    assert err.mean() <= 1e-2 and err.max() <= max_tol, f"mean {err.mean().item():.4g} max {err.max().item():.4g}"


# ------------------------------------------------------------------ elementwise kernels: bit-exact
@pytest.mark.parametrize("axis,shape_a,shape_b,extent", [
    (1, (3, 19, 40, 56), (3, 19, 40, 56), 8), (1, (3, 20, 24, 40), (3, 7, 24, 40), 16),       # blend_t (extent clamped by the short tile)
    (2, (3, 20, 32, 32), (3, 20, 17, 32), 8), (2, (3, 5, 64, 33), (3, 5, 64, 33), 64),        # blend_v, odd width => scalar path
    (3, (3, 20, 32, 32), (3, 20, 32, 9), 8), (3, (3, 4, 16, 256), (3, 4, 16, 256), 64)])      # blend_h (inner = 1)
def test_blend_kernel_bit_exact(ops, axis, shape_a, shape_b, extent):
    from oracle.vae_oracle import WanVaeTiledOracle
    a, b = rnd(shape_a, 1), rnd(shape_b, 2)
    ref = WanVaeTiledOracle.blend(a[None].clone(), b[None].clone(), extent, axis + 1)[0]
    bg = b.cuda()
    out = ops.vae_blend(a.cuda(), bg, extent, axis)
    assert out is bg and torch.equal(bg.cpu(), ref)


def test_blend_kernel_on_time_sliced_views(ops):
    """The tiles the temporal merge blends are [:, 1:] views (first frame dropped) of larger buffers."""
    from oracle.vae_oracle import WanVaeTiledOracle
    A, B = rnd((3, 21, 24, 40), 3), rnd((3, 20, 24, 40), 4)
    for axis in (1, 2, 3):
        ref = WanVaeTiledOracle.blend(A[None, :, 2:].clone(), B[None, :, 1:].clone(), 6, axis + 1)[0]
        bg = B.cuda()
        ops.vae_blend(A.cuda()[:, 2:], bg[:, 1:], 6, axis)
        assert torch.equal(bg[:, 1:].cpu(), ref) and torch.equal(bg[:, 0].cpu(), B[:, 0])
    # an empty later tile (the reference's last temporal tile can lose all its frames) is a no-op
    e = torch.empty((3, 0, 24, 40), device="cuda")
    assert ops.vae_blend(A.cuda(), e, 6, 1) is e


def test_blend_refuses_mismatched_tiles(ops):
    with pytest.raises(RuntimeError, match="differ off the blend axis"):
        ops.vae_blend(torch.zeros((3, 4, 8, 8), device="cuda"), torch.zeros((3, 4, 9, 8), device="cuda"), 2, 3)


@pytest.mark.parametrize("T,H,W", [(5, 40, 56), (2, 17, 13), (1, 8, 1030)])
def test_postprocess_u8_bit_exact(ops, T, H, W):
    from oracle.vae_oracle import postprocess_u8
    x = rnd((1, 3, T, H, W), 5, 0.8).clamp(-1, 1)
    x.view(-1)[:9] = torch.tensor([-1.0, 1.0, 0.0, -0.999, 0.999, 1.0 / 255, -1.0 / 255, 0.5, -0.5])
    ref = postprocess_u8(x)[:, 0].permute(0, 2, 3, 1).contiguous()      # [T,1,3,H,W] -> [T,H,W,3]
    got = ops.vae_postprocess_u8(x.cuda())
    assert got.dtype == torch.uint8 and got.shape == (T, H, W, 3)
    assert torch.equal(got.cpu(), ref)


# ------------------------------------------------------------------ streaming decode
def test_streaming_decode_equals_cached_decode(ops):
    from fastvideo_amd.wan_vae import WanVaeDecoderHip
    from oracle.vae_oracle import WanVaeTiledOracle
    g, sd, z = _gold()
    dec = WanVaeDecoderHip(sd, device="cuda")
    y = dec.decode(z.cuda())
    cache = dec.get_streaming_cache()
    a, cache = dec.streaming_decode(z[:, :, :3].cuda(), cache, True)
    b, cache = dec.streaming_decode(z[:, :, 3:5].cuda(), cache, False)
    c, cache = dec.streaming_decode(z[:, :, 5:].cuda(), cache, False)
    assert a.shape[2] == 9 and b.shape[2] == 8 and c.shape[2] == 8
    assert torch.equal(torch.cat([a, b, c], 2), y)                       # same kernels, same rings: bit-identical
    o = WanVaeTiledOracle(sd)
    ra, oc = o.streaming_decode(z[:, :, :3], {}, True)
    rb, oc = o.streaming_decode(z[:, :, 3:], oc, False)
    _close(torch.cat([a, b, c], 2).cpu(), torch.cat([ra, rb], 2))


# ------------------------------------------------------------------ cache-less decode family
def _pair(sd, tiles=None, **flags):
    from fastvideo_amd.wan_vae import WanVaeDecoderHip
    from oracle.vae_oracle import WanVaeTiledOracle
    dec = WanVaeDecoderHip(sd, device="cuda", use_feature_cache=False)
    o = WanVaeTiledOracle(sd)
    if tiles is not None:
        dec.enable_tiling(**tiles, **flags)
        o.enable_tiling(**tiles, **flags)
    return dec, o


def test_plain_tile_decode_vs_oracle(ops):
    g, sd, z = _gold()
    dec, o = _pair(sd)
    _close(dec.decode(z.cuda()).cpu(), o.decode_nocache(z))


def test_spatial_tiling_vs_oracle(ops):
    g, sd, z = _gold()
    dec, o = _pair(sd, g["tiles"])
    y = dec.decode(z.cuda()).cpu()
    assert tuple(y.shape) == g["sha256"]["spatial"][1]
    _close(y, o.decode_nocache(z))


def test_temporal_tiling_two_calls_vs_oracle(ops):
    """Both calls: the reference doubles blend_num_frames per call, so the second decode of the same latent differs."""
    g, sd, z = _gold()
    dec, o = _pair(sd, g["tiles"], use_temporal_tiling=True)
    y0, r0 = dec.decode(z.cuda()).cpu(), o.decode_nocache(z)
    assert tuple(y0.shape) == g["sha256"]["tiled0"][1] and dec.blend_num_frames == o.blend_num_frames == 8
    _close(y0, r0)
    _close(y0, g["full"]["tiled0"])
    y1, r1 = dec.decode(z.cuda()).cpu(), o.decode_nocache(z)
    assert dec.blend_num_frames == o.blend_num_frames == 16
    _close(y1, r1)
    assert not torch.equal(y0, y1)


def test_temporal_tiling_ragged_grid_vs_oracle(ops):
    """7x5x7 latent: 3 temporal tiles (the last one short), 2x3 spatial tiles with short edge tiles."""
    g, sd, _ = _gold()
    z = rnd((1, 16, 7, 5, 7), 7)
    dec, o = _pair(sd, g["tiles"], use_temporal_tiling=True)
    _close(dec.decode(z.cuda()).cpu(), o.decode_nocache(z))


def test_tile_parallel_single_rank_vs_oracle(ops):
    g, sd, z = _gold()
    dec, o = _pair(sd, g["tiles"], use_parallel_tiling=True)
    y = dec.parallel_tiled_decode(dec._latents_cl(z.cuda()))[:, :1 + 4 * (z.shape[2] - 1)].unsqueeze(0).cpu()
    _close(y, o.decode_nocache(z, sp_world_size=2))
    _close(y, g["full"]["parallel"])


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from fastvideo_amd.wan_vae import WanVaeDecoderHip
        g, sd, z = _gold()
        dec = WanVaeDecoderHip(sd, device="cuda:0", use_feature_cache=False, sp_group=dist.group.WORLD)
        dec.enable_tiling(**g["tiles"], use_parallel_tiling=True)
        y = dec.decode(z.cuda()).cpu()
        if rank == world - 1:  # every rank holds the full video; report the last one's
            out_q.put(_mp.ship(y))
            out_q.close(); out_q.join_thread()
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_tile_parallel_decode_across_ranks_equals_single_rank(ops, world):
    """`world` processes share cuda:0 and exchange through gloo (host-staged; production = RCCL): the tile-parallel decode must
    equal the single-rank tile-parallel decode bit for bit (tiles are decoded by the same kernels wherever they run)."""
    from fastvideo_amd.wan_vae import WanVaeDecoderHip
    g, sd, z = _gold()
    dec = WanVaeDecoderHip(sd, device="cuda", use_feature_cache=False)
    dec.enable_tiling(**g["tiles"], use_parallel_tiling=True)
    ref = dec.parallel_tiled_decode(dec._latents_cl(z.cuda()))[:, :1 + 4 * (z.shape[2] - 1)].unsqueeze(0).cpu()
    ctx = mp.get_context("spawn")
    out_q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, out_q)) for r in range(world)]
    for p in procs:
        p.start()
    y = _mp.unship(out_q.get(timeout=300))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert torch.equal(y, ref), f"max diff {(y - ref).abs().max().item()}"
    _close(y, g["full"]["parallel"])
