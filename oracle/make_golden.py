"""Generate tests/golden/* by running the REAL reference (imported from /root/reference, CPU).

Run in the build container only:  ``python oracle/make_golden.py``
(the GPU box has no /root/reference; it consumes the committed fixtures).

Fixtures
  wan_tiny.pt      tiny Wan2.1-geometry DiT (2 heads x 128, ffn 512, 2 layers, text_dim 64):
                   bf16 state_dict, two seeded inputs (S=48 and a ragged S=105), reference outputs of
                   ``WanTransformer3DModel.forward`` under bf16 autocast + every block's output.
  vsa_meta.npz     ``VideoSparseAttentionMetadataBuilder.build`` index tensors for several canvases
                   (incl. ragged ones and the 81f x 480p canvas 21x30x52), from the reference backend file
                   AND from fastvideo-kernel's vsa_utils.py (they must agree).
  rope_21x30x52.pt first/last rows + checksum of the reference RoPE tables at the cfg2 canvas.
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import ref_loader as R  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
TINY = dict(num_heads=2, head_dim=128, ffn_dim=512, num_layers=2, text_dim=64)


def wan_tiny():
    from fastvideo.forward_context import set_forward_context
    m = R.build_wan(**TINY, seed=0, modulation_std=0.05, dtype=torch.bfloat16)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    cases = []
    for seed, shape, L in [(1, (1, 16, 3, 8, 8), 16), (2, (1, 16, 3, 10, 14), 24)]:
        g = torch.Generator().manual_seed(seed)
        x = torch.randn(shape, generator=g).bfloat16()
        ctx = torch.randn(1, L, TINY["text_dim"], generator=g).bfloat16()
        ts = torch.tensor([500 + seed])
        blocks = []
        hooks = [b.register_forward_hook(lambda mod, i, o: blocks.append(o.detach().clone())) for b in m.blocks]
        with torch.no_grad(), torch.autocast("cpu", dtype=torch.bfloat16), \
                set_forward_context(current_timestep=0, attn_metadata=None):
            y = m(hidden_states=x, encoder_hidden_states=ctx, timestep=ts)
        for h in hooks:
            h.remove()
        cases.append(dict(latent=x, ctx=ctx, timestep=ts, out=y.detach().clone(), blocks=blocks))
    torch.save(dict(config=TINY, state_dict=sd, cases=cases), os.path.join(OUT, "wan_tiny.pt"))
    print("wan_tiny.pt", os.path.getsize(os.path.join(OUT, "wan_tiny.pt")) / 1e6, "MB")


def vsa_meta():
    from fastvideo.attention.backends.video_sparse_attn import VideoSparseAttentionMetadataBuilder
    ku = R.load_kernel_module("python/fastvideo_kernel/vsa_utils.py", "ref_vsa_utils")
    out = {}
    dev = torch.device("cpu")
    for lat in [(8, 32, 32), (9, 20, 14), (5, 14, 6), (2, 4, 4), (21, 60, 104), (9, 64, 64)]:
        md = VideoSparseAttentionMetadataBuilder().build(current_timestep=0, raw_latent_shape=lat,
                                                         patch_size=(1, 2, 2), VSA_sparsity=0.8, device=dev)
        shape = tuple(md.dit_seq_shape)
        km = ku.build_vsa_metadata(shape, device="cpu")
        assert torch.equal(km["tile_partition_indices"], md.tile_partition_indices)
        assert torch.equal(km["reverse_tile_partition_indices"], md.reverse_tile_partition_indices)
        assert torch.equal(km["variable_block_sizes"].long(), md.variable_block_sizes.long())
        assert torch.equal(km["non_pad_index"], md.non_pad_index)
        key = "x".join(map(str, lat))
        out[key + "/perm"] = md.tile_partition_indices.numpy().astype(np.int32)
        out[key + "/rev"] = md.reverse_tile_partition_indices.numpy().astype(np.int32)
        out[key + "/vbs"] = md.variable_block_sizes.numpy().astype(np.int32)
        out[key + "/non_pad"] = md.non_pad_index.numpy().astype(np.int32)
        out[key + "/untile"] = md.untile_combined_index.numpy().astype(np.int32)
        out[key + "/num_tiles"] = np.array(md.num_tiles, dtype=np.int32)
    np.savez_compressed(os.path.join(OUT, "vsa_meta.npz"), **out)
    print("vsa_meta.npz", os.path.getsize(os.path.join(OUT, "vsa_meta.npz")) / 1e6, "MB")


def rope():
    """rope.pt: sampled rows + checksums of the reference tables (start_frame = 0); rope_start.pt (round 4): the FULL fp32 tables of small
    grids with start_frame > 0 (the causal rollout's call, causal_wanvideo.py:586-598) and sampled rows + checksums at the 480p block grid."""
    from fastvideo.layers.rotary_embedding import get_rotary_pos_embed
    res = {}
    for grid in [(21, 30, 52), (3, 4, 4), (3, 5, 7)]:
        cos, sin = get_rotary_pos_embed(grid, 1536, 12, [44, 42, 42], dtype=torch.float64, rope_theta=10000)
        cos, sin = cos.float(), sin.float()
        rows = torch.tensor([0, 1, grid[2], grid[1] * grid[2], cos.shape[0] // 2, cos.shape[0] - 1])
        res["x".join(map(str, grid))] = dict(rows=rows, cos=cos[rows].clone(), sin=sin[rows].clone(),
                                             cos_sum=cos.double().sum().item(), sin_sum=sin.double().sum().item(),
                                             cos_abs=cos.double().abs().sum().item())
    torch.save(res, os.path.join(OUT, "rope.pt"))
    print("rope.pt ok")
    res = {}
    for grid, start in [((3, 4, 4), 0), ((3, 4, 4), 3), ((2, 5, 7), 6), ((3, 30, 52), 18), ((1, 30, 52), 20)]:
        cos, sin = get_rotary_pos_embed(grid, 1536, 12, [44, 42, 42], dtype=torch.float64, rope_theta=10000, start_frame=start)
        cos, sin = cos.float(), sin.float()
        ent = dict(cos_sum=cos.double().sum().item(), sin_sum=sin.double().sum().item(), cos_abs=cos.double().abs().sum().item())
        if cos.shape[0] <= 128:
            ent.update(rows=torch.arange(cos.shape[0]), cos=cos.clone(), sin=sin.clone())
        else:
            rows = torch.tensor([0, 1, grid[2], grid[1] * grid[2] - 1, cos.shape[0] // 2, cos.shape[0] - 1])
            ent.update(rows=rows, cos=cos[rows].clone(), sin=sin[rows].clone())
        res["x".join(map(str, grid)) + f"@{start}"] = ent
    torch.save(res, os.path.join(OUT, "rope_start.pt"))
    print("rope_start.pt ok")


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    R.init_distributed()
    only = sys.argv[1:]
    for fn in (wan_tiny, vsa_meta, rope):
        if not only or fn.__name__ in only:
            fn()
