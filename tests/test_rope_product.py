"""the PRODUCT's 3-D RoPE tables (``fastvideo_amd.rope.get_rotary_pos_embed``, what the HIP kernels are fed) must be
value-exact — bit for bit — against the reference's ``get_rotary_pos_embed`` (fastvideo/layers/rotary_embedding.py:468-564, called by
wanvideo.py:685-687 and, with ``start_frame``, by causal_wanvideo.py:586-598).  Fixtures: tests/golden/rope.pt / rope_start.pt, generated
from the real reference by oracle/make_golden.py; with /root/reference present the live function is compared on whole tables too."""
import math
import os

import pytest
import torch

from fastvideo_amd import rope as P
from oracle import ref_loader as R


def _grid_start(key):
    grid, _, start = key.partition("@")
    return tuple(int(v) for v in grid.split("x")), int(start or 0)


@pytest.mark.parametrize("fixture", ["rope.pt", "rope_start.pt"])
def test_product_rope_tables_equal_golden(golden_dir, fixture):
    g = torch.load(os.path.join(golden_dir, fixture), weights_only=False)
    assert len(g) >= 3
    for key, ref in g.items():
        grid, start = _grid_start(key)
        P._CACHE.clear()
        cos, sin = P.get_rotary_pos_embed(grid, 128, start_frame=start)
        assert cos.shape == (math.prod(grid), 128) and cos.dtype == torch.float32 and sin.dtype == torch.float32
        assert torch.equal(cos[ref["rows"]], ref["cos"]), key
        assert torch.equal(sin[ref["rows"]], ref["sin"]), key
        assert cos.double().sum().item() == ref["cos_sum"] and sin.double().sum().item() == ref["sin_sum"], key
        assert cos.double().abs().sum().item() == ref["cos_abs"], key
        # the cache returns the same object for the same key and a different table for a different start frame
        assert P.get_rotary_pos_embed(grid, 128, start_frame=start)[0] is cos
        other = P.get_rotary_pos_embed(grid, 128, start_frame=start + 1)[0]
        assert not torch.equal(other, cos)


def test_start_frame_is_a_temporal_shift(golden_dir):
    """Frames [s, s+T) of a start_frame = 0 table over T+s frames ARE the start_frame = s table (positions are exact small integers in
    fp32) — the property the causal rollout relies on (cache rows roped once, rotary_embedding.py:387-388)."""
    hw = 4 * 4
    full = P.get_rotary_pos_embed((7, 4, 4), 128)
    part = P.get_rotary_pos_embed((3, 4, 4), 128, start_frame=4)
    assert torch.equal(full[0][4 * hw:], part[0]) and torch.equal(full[1][4 * hw:], part[1])


@pytest.mark.skipif(not R.available(), reason="no reference tree (live comparison)")
@pytest.mark.parametrize("grid,start", [((21, 30, 52), 0), ((3, 30, 52), 18), ((5, 6, 10), 2), ((1, 45, 80), 0)])
def test_product_rope_tables_equal_live_reference(grid, start):
    R.install()
    from fastvideo.layers.rotary_embedding import get_rotary_pos_embed
    rc, rs = get_rotary_pos_embed(grid, 1536, 12, P.rope_dim_list(128), dtype=torch.float64, rope_theta=10000, start_frame=start)
    P._CACHE.clear()
    cos, sin = P.get_rotary_pos_embed(grid, 128, start_frame=start)
    assert torch.equal(cos, rc.float()) and torch.equal(sin, rs.float())


def test_dim_list_is_wans():
    assert P.rope_dim_list(128) == [44, 42, 42]  # wanvideo.py:679-684: d - 4*(d//6), 2*(d//6), 2*(d//6)
