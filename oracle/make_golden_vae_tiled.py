"""Generate tests/golden/vae_tiled.pt from the REAL reference (imported from /root/reference, CPU fp32): the cache-less decode
family of AutoencoderKLWan (use_feature_cache=False) and the streaming decode.

Run in the build container only:  ``python oracle/make_golden_vae_tiled.py``.
Cases (same seeded base_dim-32 decoder as vae_tiny.pt, latent [1,16,7,5,5], 32-px tiles with 24-px stride => 2x2 spatial tiles,
3 temporal tiles, the last of which holds a single latent frame and contributes no frame at all after the overrides' frame drops):
  plain      vae.decode with tiling off                   (_decode, common.py:92)
  spatial    use_tiling                                    (spatial_tiled_decode + Wan override)
  tiled0/1   use_tiling + use_temporal_tiling, 1st and 2nd call (blend_num_frames doubles per call, wanvae.py:1227)
  parallel   use_parallel_tiling on a 2-rank gloo group    (parallel_tiled_decode; sp world size/rank patched in, and
             all_gather_into_tensor routed through all_gather because gloo refuses the [world, N] output NCCL accepts)
  stream     streaming_decode in two calls (3 + 4 latent frames)
Full tensors are stored for tiled0 and parallel; the others are stored as sha256 of the fp32 bytes plus shape."""
from __future__ import annotations

import hashlib
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden", "vae_tiled.pt")
TILES = dict(tile_sample_min_height=32, tile_sample_min_width=32, tile_sample_stride_height=24, tile_sample_stride_width=24)
SEED, ZSHAPE, ZSEED = 3, (1, 16, 7, 5, 5), 11


def latent():
    return torch.randn(ZSHAPE, generator=torch.Generator().manual_seed(ZSEED))


def sha(t):
    return hashlib.sha256(t.contiguous().float().numpy().tobytes()).hexdigest()


def _parallel_rank(rank, ws, port, path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    from oracle.make_golden_vae import build_ref_vae
    vae = build_ref_vae(base_dim=32, seed=SEED)  # also puts the reference checkout on sys.path
    import fastvideo.models.vaes.common as C
    C.get_sp_world_size = lambda: ws
    C.get_sp_parallel_rank = lambda: rank

    def gather_rows(out, inp):  # transport shim only
        parts = [torch.empty_like(inp) for _ in range(ws)]
        dist.all_gather(parts, inp)
        out.copy_(torch.stack(parts).view_as(out))

    C.dist.all_gather_into_tensor = gather_rows
    vae.use_feature_cache = False
    vae.enable_tiling(**TILES, use_parallel_tiling=True)
    with torch.no_grad():
        y = vae.decode(latent())
    if rank == 0:
        torch.save(y, path)
    dist.barrier()
    dist.destroy_process_group()


def reference_parallel(ws=2, port=29655):
    path = f"/tmp/_vae_tiled_parallel_{os.getpid()}.pt"
    ctx = mp.get_context("spawn")
    ps = [ctx.Process(target=_parallel_rank, args=(r, ws, port, path)) for r in range(ws)]
    [p.start() for p in ps]
    [p.join(600) for p in ps]
    assert all(p.exitcode == 0 for p in ps), [p.exitcode for p in ps]
    y = torch.load(path)
    os.remove(path)
    return y


def main():
    from oracle.make_golden_vae import build_ref_vae
    vae = build_ref_vae(base_dim=32, seed=SEED)
    z = latent()
    out = {}
    with torch.no_grad():
        y_cached = vae.decode(z)
        cache = vae.get_streaming_cache()
        s1, cache = vae.streaming_decode(z[:, :, :3], cache, True)
        s2, cache = vae.streaming_decode(z[:, :, 3:], cache, False)
        out["stream"] = torch.cat([s1, s2], 2)
        assert torch.equal(out["stream"], y_cached)
        vae.use_feature_cache = False
        out["plain"] = vae.decode(z)
        vae.enable_tiling(**TILES)
        out["spatial"] = vae.decode(z)
        vae.enable_tiling(**TILES, use_temporal_tiling=True)
        out["tiled0"] = vae.decode(z)
        out["tiled1"] = vae.decode(z)
    out["parallel"] = reference_parallel()
    fix = {"param_spec": vae.param_spec, "seed": SEED, "z_shape": ZSHAPE, "z_seed": ZSEED, "tiles": TILES,
           "full": {k: out[k] for k in ("tiled0", "parallel")},
           "sha256": {k: (sha(v), tuple(v.shape)) for k, v in out.items()}}
    torch.save(fix, OUT)
    print("wrote", OUT, os.path.getsize(OUT) // 1024, "KiB", {k: tuple(v.shape) for k, v in out.items()})


if __name__ == "__main__":
    main()
