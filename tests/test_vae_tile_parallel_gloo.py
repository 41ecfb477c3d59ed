"""World-size-2/3 gloo test (CPU) of the tile-parallel VAE decode's HOST logic in fastvideo_amd.wan_vae: tile plan, per-rank
runs, the single all_gather_into_tensor, un-flattening from plan-derived shapes, spatial merge and temporal merge.  The two
device ops it calls (the tile decoder and the cross-fade kernel) are replaced by the oracle's fp32 CPU equivalents, so the
merged video must equal the REAL reference's parallel_tiled_decode output (tests/golden/vae_tiled.pt) bit for bit."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

GOLD = os.path.join(os.path.dirname(__file__), "golden", "vae_tiled.pt")


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _host_decoder(sd, tiles):
    """WanVaeDecoderHip on CPU tensors with its two device ops swapped for the oracle's."""
    import fastvideo_amd.wan_vae as wv
    from oracle.vae_oracle import WanVaeTiledOracle
    dec = wv.WanVaeDecoderHip(sd, device="cpu", use_feature_cache=False)
    dec.enable_tiling(**tiles, use_parallel_tiling=True)
    o = WanVaeTiledOracle(sd)

    def decode_tile(zc):                     # channels-last fp32 [T,h,w,64] -> planar [3,4T,8h,8w]
        return o.decode_tile(zc[..., :16].permute(3, 0, 1, 2)[None].contiguous())[0]

    def blend(a, b, extent, axis):
        return WanVaeTiledOracle.blend(a[None], b[None], extent, axis + 1)[0]

    dec._decode_tile = decode_tile
    wv.ops.vae_blend = blend
    return dec


def _latent_cl(z):
    zc = torch.zeros((z.shape[2], z.shape[3], z.shape[4], 64))
    zc[..., :16] = z[0].permute(1, 2, 3, 0)
    return zc


def _worker(rank, world, port, path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle.vae_oracle import seeded_state_dict
        g = torch.load(GOLD, weights_only=False)
        sd = seeded_state_dict(g["param_spec"], g["seed"])
        z = torch.randn(g["z_shape"], generator=torch.Generator().manual_seed(g["z_seed"]))
        dec = _host_decoder(sd, g["tiles"])
        dec.sp_group = dist.group.WORLD
        with torch.no_grad():
            y = dec.parallel_tiled_decode(_latent_cl(z))[:, :1 + 4 * (z.shape[2] - 1)].unsqueeze(0)
        assert dec.blend_num_frames == 8
        torch.save(y, f"{path}.{rank}")
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_tile_parallel_host_logic_matches_reference(world, tmp_path):
    path = str(tmp_path / "y")
    ctx = mp.get_context("spawn")
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, path)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=600)
        assert p.exitcode == 0
    g = torch.load(GOLD, weights_only=False)
    for r in range(world):
        y = torch.load(f"{path}.{r}")
        assert torch.equal(y, g["full"]["parallel"]), f"rank {r}: max diff {(y - g['full']['parallel']).abs().max().item()}"


def test_tile_plan_and_shapes():
    from oracle.vae_oracle import seeded_state_dict
    g = torch.load(GOLD, weights_only=False)
    dec = _host_decoder(seeded_state_dict(g["param_spec"], g["seed"]), g["tiles"])
    plan, grid = dec.tile_plan(7, 5, 7)
    assert grid == (3, 2, 3) and len(plan) == 18 and plan[0] == (0, 0, 0) and plan[-1] == (6, 3, 6)
    L = dec._latent_tiles()
    assert (L["mt"], L["mh"], L["mw"], L["st"], L["sh"], L["sw"], L["bh"], L["bw"]) == (4, 4, 4, 3, 3, 3, 8, 8)
    assert dec._tile_out_shape((0, 0, 0), 7, 5, 7, L) == (3, 20, 32, 32)
    assert dec._tile_out_shape((3, 3, 6), 7, 5, 7, L) == (3, 15, 16, 8)      # later tiles drop their first frame; edge tiles are short
    assert dec._tile_out_shape((6, 0, 0), 7, 5, 7, L) == (3, 3, 32, 32)
