"""Bit-exact check of the sliding-tile -> block-sparse index construction (fastvideo_amd.kernel_api.sliding_tile_block_lists, host /
C-ABI integer code, no GPU): expanding the lists back to a token mask must reproduce the oracle's mask (oracle/vsa_oracle.py
sta_mask_ragged = the reference's window rule, fastvideo-kernel/tests/support_flex_sta.py:29-59, on the padded tile grid) exactly."""
import numpy as np
import pytest
import torch

from oracle import vsa_oracle as V


@pytest.mark.parametrize("grid,tile,window", [((7, 9, 17), (2, 4, 8), (3, 3, 3)), ((7, 9, 17), (2, 4, 8), (1, 3, 1)),
                                              ((12, 16, 24), (6, 8, 8), (3, 1, 3)), ((5, 6, 9), (2, 2, 16), (3, 3, 1)),
                                              ((4, 8, 8), (2, 4, 8), (5, 3, 3)), ((21, 30, 52), (6, 8, 8), (3, 3, 3))])
def test_block_lists_expand_to_the_oracle_mask(grid, tile, window):
    from fastvideo_amd import kernel_api as K
    m = K.sliding_tile_block_lists(grid, tile, window)
    S = grid[0] * grid[1] * grid[2]
    perm, non_pad = m["tile_partition_indices"].numpy(), m["non_pad_index"].numpy()
    bsz, idx, num, qb = m["block_sizes"].numpy(), m["q2k_idx"].numpy(), m["q2k_num"].numpy(), m["q_block"]
    assert m["S_pad"] % qb == 0 and len(num) == m["S_pad"] // qb
    assert all((np.diff(idx[i, :num[i]]) > 0).all() for i in range(len(num)))  # ascending, no duplicates
    # padded position of every real token (tile-major, real tokens first in each tile) and back
    pos_of_raster = np.empty(S, dtype=np.int64)
    pos_of_raster[perm] = non_pad                     # tile(): dst[non_pad[i]] = src[perm[i]]
    assert len(set(pos_of_raster.tolist())) == S
    raster_of_pos = -np.ones(m["S_pad"], dtype=np.int64)
    raster_of_pos[pos_of_raster] = np.arange(S)
    # block sizes = number of real tokens in each 64-slot block
    real = (raster_of_pos >= 0).reshape(-1, 64)
    assert (real.sum(1) == bsz).all() and all(r[:c].all() and not r[c:].any() for r, c in zip(real, bsz))
    if S > 6000:  # the cfg3 grid: check a sample of query rows (the dense mask would be 1 G entries)
        rows = np.random.default_rng(0).choice(S, 300, replace=False)
    else:
        rows = np.arange(S)
    ref = V.sta_mask_ragged(grid, window, tile).numpy()[rows] if S <= 6000 else None
    got = np.zeros((len(rows), S), dtype=bool)
    for i, r in enumerate(rows):
        qblock = pos_of_raster[r] // qb
        for b in idx[qblock, :num[qblock]]:
            keys = raster_of_pos[b * 64:b * 64 + bsz[b]]
            got[i, keys] = True
    if ref is None:
        nt = m["num_tiles"]
        coord = np.stack(np.meshgrid(*[np.arange(n) for n in grid], indexing="ij"), -1).reshape(-1, 3) // np.array(tile)
        ref = np.ones((len(rows), S), dtype=bool)
        for ax in range(3):
            w = np.array([V.sta_window(q, nt[ax], window[ax]) for q in range(nt[ax])])
            lo, hi = w[coord[rows, ax], 0], w[coord[rows, ax], 1]
            ref &= (coord[None, :, ax] >= lo[:, None]) & (coord[None, :, ax] < hi[:, None])
    assert np.array_equal(got, ref)
    # the per-TILE form (fvk_attn_tile_lists_bf16, one list per tile) and the query-GROUPED form (queries packed by window class, one list
    # per 256 packed rows) must expand to the same mask
    tok = m["tile_tokens"]
    t_idx, t_num, t_val = m["tile_q2k_idx"].numpy(), m["tile_q2k_num"].numpy(), m["tile_rows_valid"].numpy()
    assert len(t_num) == m["S_pad"] // tok and (t_val == real.reshape(len(t_num), -1).sum(1)).all()
    g_src, g_dst, g_unt = m["group_src"].numpy(), m["group_dst"].numpy(), m["group_untile"].numpy()
    g_idx, g_num = m["group_q2k_idx"].numpy(), m["group_q2k_num"].numpy()
    assert sorted(g_src.tolist()) == list(range(S)) and len(set(g_dst.tolist())) == S and g_dst.max() < m["group_rows"]
    assert m["group_rows"] == 256 * len(g_num) and (g_unt[g_src] == g_dst).all()
    assert m["group_rows"] - S < 256 * m["n_window_classes"]  # less than one 256-row group of padding per window class
    for form in ("tile", "grouped"):
        got2 = np.zeros((len(rows), S), dtype=bool)
        for i, r in enumerate(rows):
            lst = t_idx[pos_of_raster[r] // tok, :t_num[pos_of_raster[r] // tok]] if form == "tile" else g_idx[g_unt[r] // 256, :g_num[g_unt[r] // 256]]
            for b in lst:
                got2[i, raster_of_pos[b * 64:b * 64 + bsz[b]]] = True
        assert np.array_equal(got2, ref), form
    assert abs(m["density"] - (got.mean() if len(rows) == S else m["density"])) < 1e-9 and 0 < m["density"] <= 1


@pytest.mark.parametrize("canvas,tile,windows", [((12, 16, 24), (6, 8, 8), ((3, 3, 3), (1, 3, 1), (3, 1, 5), (2, 2, 3))),
                                                 ((6, 16, 32), (6, 8, 8), ((1, 1, 1), (3, 1, 10)))])
def test_canvas_tile_lists_expand_to_the_oracle_mask(canvas, tile, windows):
    """kernel_api.sliding_tile_attention's per-(head, tile) KV lists (the form fvk_attn_tile_lists_bf16 consumes) against the oracle's
    sliding-tile mask (oracle/vsa_oracle.py sta_mask = fastvideo-kernel/tests/support_flex_sta.py:29-59), even window sizes included."""
    from fastvideo_amd import kernel_api as K
    tok = tile[0] * tile[1] * tile[2]
    tiles = tuple(c // t for c, t in zip(canvas, tile))
    idx, num, sizes = K._canvas_tile_lists(tiles, tok, windows, 2, torch.device("cpu"))
    S = canvas[0] * canvas[1] * canvas[2]
    assert idx.shape[:3] == (2, len(windows), S // tok) and (sizes == 64).all() and sizes.numel() == S // 64
    assert torch.equal(idx[0], idx[1]) and torch.equal(num[0], num[1])
    for h, w in enumerate(windows):
        ref = V.sta_mask(canvas, w, tile).numpy()           # [S, S] in tile-major token order
        got = np.zeros((S, S), dtype=bool)
        for t in range(S // tok):
            blocks = idx[0, h, t, :num[0, h, t]].numpy()
            assert len(set(blocks.tolist())) == len(blocks)
            for b in blocks:
                got[t * tok:(t + 1) * tok, b * 64:(b + 1) * 64] = True
        assert np.array_equal(got, ref), (h, w)
