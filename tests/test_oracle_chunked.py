"""The memory-bounded forms of the sparse-attention oracles that the full-geometry GPU tests use (tests/test_gpu_fullgeom.py) are the
SAME functions as the dense-mask forms pinned elsewhere: compared here at sizes where both fit (CPU, seconds)."""
import numpy as np
import torch

from oracle import vsa_oracle as V
from oracle import wan_oracle as W


def test_block_sparse_gathered_equals_dense_mask_form():
    lat = (5, 20, 28)                       # token grid (5,10,14): ragged tiles in every axis
    md = V.build_metadata(lat)
    vbs = md["variable_block_sizes"]
    nb = len(vbs)
    g = torch.Generator().manual_seed(3)
    B, H, D = 1, 3, 32
    q, k, v = (torch.randn((B, H, nb * 64, D), generator=g).bfloat16() for _ in range(3))
    rng = np.random.default_rng(0)
    mask = rng.random((B, H, nb, nb)) < 0.4
    mask[..., 0] = True                     # no empty row
    a = V.block_sparse_attn(q, k, v, mask, vbs)
    b = V.block_sparse_attn_gathered(q, k, v, mask, vbs)
    assert torch.allclose(a, b, atol=2e-6, rtol=1e-5), (a - b).abs().max()
    topk = V.compute_topk(0.6, nb)
    o1, i1 = V.video_sparse_attn(q, k, v, vbs, vbs, topk)
    o2, i2 = V.video_sparse_attn(q, k, v, vbs, vbs, topk, gathered=True)
    assert np.array_equal(i1["mask"], i2["mask"])
    assert (o1.float() - o2.float()).abs().max() <= 2 ** -7 * o1.float().abs().max()   # one bf16 ulp at most (fp32 summation order)
    assert (o1 != o2).float().mean() < 1e-3


def test_sta_ragged_tilewise_equals_dense_mask_form():
    grid, tile = (7, 9, 17), (2, 4, 8)
    S = 7 * 9 * 17
    g = torch.Generator().manual_seed(4)
    q, k, v = (torch.randn((1, 2, S, 16), generator=g).bfloat16() for _ in range(3))
    for window in ((3, 3, 3), (1, 3, 1), (3, 1, 3)):
        m = V.sta_mask_ragged(grid, window, tile)
        a = W.attention_fp32_ref(q, k, v, 0.25, m)
        b = V.sta_attention_ragged(q, k, v, 0.25, grid, window, tile)
        assert torch.allclose(a, b, atol=2e-6, rtol=1e-5), (window, (a - b).abs().max())
