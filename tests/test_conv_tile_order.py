"""The XCD-aware tile order of vae_conv3w_kernel (fastvideo_amd/csrc/vae_conv3w.hip, `decode`), restated in Python and checked for what the
kernel relies on: every linear id < ntiles maps to exactly one tile (a bijection, whatever ntiles is — launches with fewer tiles than CUs and
tile counts that are not multiples of 8 included), the ids of one XCD (id & 7) walk ONE contiguous run of the spatial-major / frame-fastest
sequence, and consecutive ids of an XCD are consecutive frames of one spatial position (the input slabs they share).  Host-side integer logic
only — the device code is compared with the 8-wave kernel (which keeps the plain n, w, h, t order) in scripts/probes/variant_tests.py."""
import itertools

import pytest


def decode(i, ntiles, T, tiles_h, tiles_w, ntn):
    base, rem = ntiles >> 3, ntiles & 7
    x, j = i & 7, i >> 3
    m = x * base + min(x, rem) + j
    sp, t = divmod(m, T)
    pid_n = sp % ntn
    sp //= ntn
    return t, sp // tiles_w, sp % tiles_w, pid_n          # (t_out, th_i, tw_i, pid_n)


@pytest.mark.parametrize("T,tiles_h,tiles_w,ntn", [(16, 30, 26, 1), (16, 30, 13, 1), (8, 15, 7, 2), (4, 8, 4, 2), (1, 30, 26, 1), (1, 4, 4, 2), (3, 1, 1, 1),
                                                 (5, 3, 7, 4), (16, 1, 1, 1)])
def test_decode_is_a_bijection_and_xcd_runs_are_contiguous(T, tiles_h, tiles_w, ntn):
    ntiles = T * tiles_h * tiles_w * ntn
    tiles = [decode(i, ntiles, T, tiles_h, tiles_w, ntn) for i in range(ntiles)]
    assert set(tiles) == set(itertools.product(range(T), range(tiles_h), range(tiles_w), range(ntn)))   # every tile exactly once
    seq = lambda t, th, tw, n: ((th * tiles_w + tw) * ntn + n) * T + t                                    # spatial-major, frame fastest
    for x in range(8):
        ms = [seq(*tiles[i]) for i in range(x, ntiles, 8)]
        assert ms == list(range(ms[0], ms[0] + len(ms))) if ms else True                                 # one contiguous run, walked in order
    # the runs tile [0, ntiles) in XCD order
    starts = [seq(*tiles[x]) for x in range(min(8, ntiles))]
    assert starts == sorted(starts) and starts[0] == 0


def test_an_xcds_tiles_in_flight_share_input_frames():
    """The contract's full-resolution stage: 30 x 26 spatial tiles x 16 frames on 256 persistent workgroups (32 per XCD).  The 32 tiles an XCD
    starts with are 2 spatial positions x 16 frames: 32 x 3 input slabs requested, 2 x 18 distinct — every slab beyond those is an L2 hit."""
    T, th, tw, ntn = 16, 30, 26, 1
    ntiles = T * th * tw * ntn
    first = [decode(b, ntiles, T, th, tw, ntn) for b in range(0, 256, 8)]     # XCD 0's workgroups: b & 7 == 0
    spatial = {(h, w) for _, h, w, _ in first}
    assert len(first) == 32 and len(spatial) == 2
    slabs = {(h, w, t + dt) for t, h, w, _ in first for dt in range(3)}
    assert len(slabs) == 2 * 18
