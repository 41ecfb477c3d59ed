"""3-D rotary tables for the Wan DiT (host side, computed once per latent grid and cached).

Value-exact restatement of the reference's table construction — float64 frequencies, fp32 positions from
``linspace(0, n, n+1)[:n]``, ``outer``, cos/sin, ``repeat_interleave(2)``, concatenation over the (t,h,w) axes and the
final ``.float()`` — fastvideo/layers/rotary_embedding.py:245-346, 349-450, 468-564; fastvideo/models/dits/wanvideo.py:679-687.
The tables are small ([S,128] fp32), built with torch on the CPU exactly as the reference does, then uploaded."""
from __future__ import annotations

from collections import OrderedDict

import torch

_CACHE: "OrderedDict[tuple, tuple[torch.Tensor, torch.Tensor]]" = OrderedDict()
_CACHE_MAX = 16  # same bound as the reference's LRU (rotary_embedding.py:455-460)


def rope_dim_list(head_dim: int) -> list[int]:
    return [head_dim - 4 * (head_dim // 6), 2 * (head_dim // 6), 2 * (head_dim // 6)]


def get_rotary_pos_embed(grid_thw, head_dim: int, theta: float = 10000.0, device="cpu", start_frame: int = 0):
    """``start_frame`` shifts the temporal positions (``full_grid[0] += start_frame``, rotary_embedding.py:387-388): the causal
    model's block-by-block rollout (causal_wanvideo.py:586-598; that path keeps float64 tables, the kernels take their fp32 cast)."""
    key = (tuple(grid_thw), head_dim, theta, str(device), int(start_frame))
    hit = _CACHE.get(key)
    if hit is not None:
        _CACHE.move_to_end(key)
        return hit
    dims = rope_dim_list(head_dim)
    assert sum(dims) == head_dim
    axes = [torch.linspace(0, n, n + 1, dtype=torch.float32)[:n] for n in grid_thw]
    grid = torch.stack(torch.meshgrid(*axes, indexing="ij"), dim=0)
    if start_frame > 0:
        grid[0] += start_frame
    cos_l, sin_l = [], []
    for i, dim in enumerate(dims):
        pos = grid[i].reshape(-1)
        freqs = 1.0 / (theta**(torch.arange(0, dim, 2)[:(dim // 2)].to(torch.float64) / dim))
        fr = torch.outer(pos * 1.0, freqs)
        cos_l.append(fr.cos().repeat_interleave(2, dim=-1))
        sin_l.append(fr.sin().repeat_interleave(2, dim=-1))
    out = (torch.cat(cos_l, dim=1).float().contiguous().to(device), torch.cat(sin_l, dim=1).float().contiguous().to(device))
    if len(_CACHE) >= _CACHE_MAX:
        _CACHE.popitem(last=False)
    _CACHE[key] = out
    return out
