"""Drop-in for the public surface of the reference's separate kernel wheel ``fastvideo_kernel``
(fastvideo-kernel/python/fastvideo_kernel/__init__.py:40-62) — same names, argument meaning and error
behaviour, MI355X HIP kernels underneath.  With ``sys.modules['fastvideo_kernel'] = fastvideo_amd.kernel_api``
 the reference's unmodified ``VideoSparseAttentionBackend`` runs on gfx950
(fastvideo/attention/backends/video_sparse_attn.py:8-15 imports these names at module import).

Layouts: ``[B, H, S, D]`` as in the reference functions.  bf16 only (the reference kernels are bf16 only:
fastvideo-kernel/csrc/attention/st_attn_h100.cu:386-411, block_sparse_h100.cu:700-717)."""
from __future__ import annotations

import functools
import math

import torch

from . import ops

VSA_TILE_SIZE = (4, 4, 4)  # ref: fastvideo_kernel/vsa_utils.py / video_sparse_attn.py:28


# ------------------------------------------------------------------ index helpers (ref: vsa_utils.py:30-156)
@functools.lru_cache(maxsize=16)
def _meta(dit_seq_shape, tile_size):
    return ops.vsa_build_metadata_host(tuple(dit_seq_shape), tuple(tile_size))


def get_tile_partition_indices(dit_seq_shape, tile_size=VSA_TILE_SIZE, device="cpu"):
    return _meta(tuple(dit_seq_shape), tuple(tile_size))["tile_partition_indices"].long().to(device)


def get_reverse_tile_partition_indices(dit_seq_shape, tile_size=VSA_TILE_SIZE, device="cpu"):
    return _meta(tuple(dit_seq_shape), tuple(tile_size))["reverse_tile_partition_indices"].long().to(device)


def construct_variable_block_sizes(dit_seq_shape, num_tiles=None, device="cpu", tile_size=VSA_TILE_SIZE):
    m = _meta(tuple(dit_seq_shape), tuple(tile_size))
    if num_tiles is not None and tuple(num_tiles) != m["num_tiles"]:
        raise ValueError(f"num_tiles {tuple(num_tiles)} inconsistent with shape/tile ({m['num_tiles']})")
    return m["variable_block_sizes"].to(device)


def get_non_pad_index(variable_block_sizes, max_block_size):
    vbs = variable_block_sizes
    n = vbs.shape[0]
    pad = torch.arange(n, device=vbs.device)[:, None] * max_block_size + torch.arange(max_block_size, device=vbs.device)[None, :]
    return pad[torch.arange(max_block_size, device=vbs.device)[None, :] < vbs[:, None]]


def build_vsa_metadata(dit_seq_shape, tile_size=VSA_TILE_SIZE, device="cpu"):
    """ref: vsa_utils.py build_vsa_metadata — same keys; volumes other than 64/128/256 are refused."""
    vol = math.prod(tile_size)
    if vol not in (64, 128, 256):
        raise ValueError(f"Unsupported VSA tile volume {vol}")
    m = _meta(tuple(dit_seq_shape), tuple(tile_size))
    return {
        "tile_partition_indices": m["tile_partition_indices"].long().to(device),
        "reverse_tile_partition_indices": m["reverse_tile_partition_indices"].long().to(device),
        "variable_block_sizes": m["variable_block_sizes"].to(device),
        "non_pad_index": m["non_pad_index"].long().to(device),
        "num_tiles": m["num_tiles"],
        "max_block_size": vol,
    }


# ------------------------------------------------------------------ attention entry points
def _check_bf16(*ts):
    for t in ts:
        if t.dtype != torch.bfloat16:
            raise RuntimeError(f"fastvideo_amd kernels are bf16 only, got {t.dtype}")


def _check_cuda(*ts):
    """Called after the argument checks, right before an op is dispatched: said here in one line — the dispatcher's own refusal of a CPU
    tensor is a page long."""
    for t in ts:
        if not t.is_cuda:
            raise RuntimeError("fastvideo_amd: q / k / v must be ROCm device tensors (the HIP path has no CPU fallback)")


def block_sparse_attn_from_indices(q, k, v, q2k_idx, q2k_num, variable_block_sizes):
    """ref: fastvideo_kernel/block_sparse_attn.py (``block_sparse_attn_from_indices``) -> (o, lse).  Registered as
    ``torch.ops.fastvideo_kernel.block_sparse_attn_gfx950`` beside the reference's ``_triton`` / ``_sm90`` ops (block_sparse_attn.py:103-145,
    224-267); forward only (the reference registers backward ops for training, out of scope here: SURVEY §8)."""
    _check_bf16(q, k, v)
    _check_cuda(q, k, v)
    return torch.ops.fastvideo_kernel.block_sparse_attn_gfx950(q, k, v, q2k_idx, q2k_num, variable_block_sizes)


def block_sparse_attn(q, k, v, block_map, variable_block_sizes):
    """ref: fastvideo_kernel/block_sparse_attn.py:103-145 — bool block map [B,H,Nq,Nkv] -> (o, lse)."""
    idx, num = ops.map_to_index(block_map)
    return block_sparse_attn_from_indices(q, k, v, idx, num, variable_block_sizes)


def video_sparse_attn(q, k, v, variable_block_sizes, q_variable_block_sizes, topk, block_size=64,
                      compress_attn_weight=None, return_intermediates=False):
    """ref: fastvideo_kernel/ops.py:65-133.  q,k,v(,gate) [B,H,S_pad,D] bf16, tile-major, zero padded."""
    if isinstance(block_size, int):
        block_size = (block_size, block_size, block_size)
    block_elements = block_size[0] * block_size[1] * block_size[2]
    batch, heads, q_seq_len, dim = q.shape
    kv_seq_len = k.shape[2]
    if k.shape[0] != batch or v.shape[0] != batch or k.shape[1] != heads or v.shape[1] != heads:
        raise ValueError("Expected q/k/v to have the same batch and head dimensions.")
    if v.shape[2] != kv_seq_len:
        raise ValueError(f"Expected k and v to have the same sequence length, got k.shape[2]={kv_seq_len}, "
                         f"v.shape[2]={v.shape[2]}")
    if block_elements not in (64, 128, 256):
        raise ValueError(f"video_sparse_attn: block_elements must be 64, 128 or 256 (got {block_elements})")
    if q_seq_len % block_elements != 0 or kv_seq_len % block_elements != 0:
        raise ValueError(f"q_seq_len and kv_seq_len must be divisible by block_elements={block_elements}, "
                         f"got q_seq_len={q_seq_len}, kv_seq_len={kv_seq_len}")
    q_num_blocks, kv_num_blocks = q_seq_len // block_elements, kv_seq_len // block_elements
    if variable_block_sizes.numel() != kv_num_blocks:
        raise ValueError(f"variable_block_sizes must have length kv_num_blocks={kv_num_blocks}, got {variable_block_sizes.numel()}")
    if q_variable_block_sizes.numel() != q_num_blocks:
        raise ValueError(f"q_variable_block_sizes must have length q_num_blocks={q_num_blocks}, got {q_variable_block_sizes.numel()}")
    _check_bf16(q, k, v)
    vbs = variable_block_sizes.to(device=q.device, dtype=torch.int32)
    qvbs = q_variable_block_sizes.to(device=q.device, dtype=torch.int32)

    _check_cuda(q, k, v)
    if return_intermediates:  # test / debug form: the composition's intermediates as a dict, not an op
        return _vsa_forward(q, k, v, vbs, qvbs, topk, compress_attn_weight, "bhsd", True, block_elements)
    return torch.ops.fastvideo_kernel.video_sparse_attn_gfx950(q, k, v, vbs, qvbs, int(topk), block_elements, compress_attn_weight, "bhsd")


def _expand_block_lists(idx, num, vbs, block_elements):
    """KV lists over blocks of 128 / 256 tokens -> lists over the kernel's 64-key blocks: block b is sub-blocks b*sub + t with sizes
    clamp(vbs[b] - 64 t, 0, 64) (a block's real tokens come first, video_sparse_attn.py:170-189); a query block of 256 rows is two
    128-row lists sharing one KV list.  Pure integer index arithmetic on [B, H, Nq, max_kv]-sized tensors."""
    sub = block_elements // 64
    t = torch.arange(sub, device=idx.device, dtype=idx.dtype)
    idx_s = (idx.unsqueeze(-1) * sub + t).flatten(-2)
    num_s = num * sub
    sizes = (vbs.to(torch.int32)[:, None] - 64 * t.to(torch.int32)[None, :]).clamp_(0, 64).flatten()
    rep = block_elements // 128
    if rep > 1:
        idx_s, num_s = idx_s.repeat_interleave(rep, dim=2), num_s.repeat_interleave(rep, dim=2)
    return idx_s.contiguous(), num_s.contiguous(), sizes.contiguous()


def _vsa_forward(q, k, v, vbs, qvbs, topk, gate, layout, return_intermediates=False, block_elements=64, token_of_row=None, n_tokens=None,
                 v_src_rows=None):
    """The VSA composition (fastvideo_kernel/ops.py:108-128) on tensors of either layout: "bhsd" (the package API) or "bshd" (what the model
    host holds — strides go to the kernels, nothing is transposed or copied).  ``block_elements`` 64 (Wan: tile (4,4,4)) runs the 64-row
    list kernel; 128 / 256 (the reference's Blackwell CuTe paths, ops.py:125-128) run the 128-row list kernel over expanded lists."""
    if layout == "bhsd":
        batch, heads, q_seq_len, dim = q.shape
        kv_seq_len = k.shape[2]
    else:
        batch, q_seq_len, heads, dim = q.shape
        kv_seq_len = k.shape[1]
    q_num_blocks, kv_num_blocks = q_seq_len // block_elements, kv_seq_len // block_elements
    # compression branch (ops.py:108-118): block means, coarse scores (bf16, /sqrt(D)), coarse attention
    q_c = ops.block_mean(q, qvbs, block_elements, layout=layout)
    k_c = ops.block_mean(k, vbs, block_elements, layout=layout)
    # v_src_rows (model host, 64-token blocks, "bshd"): v arrives in TOKEN order [B, S, H, D]; tile-major key row p is token v_src_rows[p] (int32,
    # a multiple of 128 entries, negative = padding) — tile(v) folded into the block means and into the V^T layout pass, no gathered copy
    vt = None
    if v_src_rows is not None:
        if block_elements != 64 or layout != "bshd":
            raise ValueError("_vsa_forward: v_src_rows serves the 64-token-block bshd path only")
        v_c = ops.block_mean(v, vbs, block_elements, layout=layout, src_rows=v_src_rows)
        vt = ops.v_transpose(v, src_rows=v_src_rows)
    else:
        v_c = ops.block_mean(v, vbs, block_elements, layout=layout)
    scores = ops.gemm_batched(q_c.view(batch * heads, q_num_blocks, dim), k_c.view(batch * heads, kv_num_blocks, dim),
                              epilogue=ops.EPI_DIV, scalar=dim**0.5).view(batch, heads, q_num_blocks, kv_num_blocks)
    out_c = ops.attn_dense(q_c, k_c, v_c, scale=dim**-0.5, layout="bhsd")
    # sparse branch (ops.py:120-128): exact top-k mask -> ascending index lists -> block-sparse attention
    mask = ops.topk_mask(scores, min(int(topk), kv_num_blocks))
    idx, num = ops.map_to_index(mask)
    if block_elements == 64:
        out_s = ops.attn_block_sparse(q, k, v, idx, num, vbs, layout=layout, vt=vt)
    else:
        idx_s, num_s, sizes = _expand_block_lists(idx, num, vbs, block_elements)
        out_s = ops.attn_block_sparse(q, k, v, idx_s, num_s, sizes, layout=layout, q_block=128)
    # token_of_row (model host only): gate arrives in TOKEN order and the result is returned in token order — tile(gate) and untile(out)
    # folded into the combine pass
    out = ops.vsa_combine(out_c, out_s, gate, block_elements, layout=layout, token_of_row=token_of_row, n_tokens=n_tokens)
    if return_intermediates:
        return out, dict(q_c=q_c, k_c=k_c, v_c=v_c, scores=scores, mask=mask, q2k_idx=idx, q2k_num=num, out_c=out_c,
                         out_s=out_s)
    return out


def video_sparse_attn_bshd(q, k, v, variable_block_sizes, q_variable_block_sizes, topk, block_size=64, compress_attn_weight=None):
    """ref: fastvideo_kernel ``video_sparse_attn_bshd`` (fastvideo-kernel/python/fastvideo_kernel/ops.py:136-200): q,k,v(,gate)
    [B,S_pad,H,D] bf16, tile-major, zero padded; 128- / 256-token blocks as in the reference, and — an extension the model host uses —
    the 64-token block too (the reference's bshd entry refuses 64 only because its 64-block kernels want [B,H,S,D])."""
    if isinstance(block_size, (tuple, list)):
        block_size = block_size[0] * block_size[1] * block_size[2]
    if block_size not in (64, 128, 256):
        raise ValueError(f"video_sparse_attn_bshd: block_elements must be 64, 128 or 256 (got {block_size})")
    if k.shape[0] != q.shape[0] or v.shape[0] != q.shape[0] or k.shape[2] != q.shape[2] or v.shape[2] != q.shape[2]:
        raise ValueError("Expected q/k/v to have the same batch and head dimensions.")
    if v.shape[1] != k.shape[1]:
        raise ValueError(f"Expected k and v to have the same sequence length, got k.shape[1]={k.shape[1]}, v.shape[1]={v.shape[1]}")
    if q.shape[1] % block_size or k.shape[1] % block_size:
        raise ValueError(f"q_seq_len and kv_seq_len must be divisible by block_elements={block_size}, "
                         f"got q_seq_len={q.shape[1]}, kv_seq_len={k.shape[1]}")
    _check_bf16(q, k, v)
    vbs = variable_block_sizes.to(device=q.device, dtype=torch.int32)
    qvbs = q_variable_block_sizes.to(device=q.device, dtype=torch.int32)
    if vbs.numel() != k.shape[1] // block_size:
        raise ValueError(f"variable_block_sizes must have length kv_num_blocks={k.shape[1] // block_size}, got {vbs.numel()}")
    if qvbs.numel() != q.shape[1] // block_size:
        raise ValueError(f"q_variable_block_sizes must have length q_num_blocks={q.shape[1] // block_size}, got {qvbs.numel()}")
    _check_cuda(q, k, v)
    return torch.ops.fastvideo_kernel.video_sparse_attn_gfx950(q, k, v, vbs, qvbs, int(topk), int(block_size), compress_attn_weight, "bshd")


_STA_CANVAS = {"30x48x80": (30, 48, 80), "36x48x48": (36, 48, 48), "18x48x80": (18, 48, 80)}


def sliding_tile_attention(q, k, v, window_size, text_length, has_text=True, seq_shape="30x48x80", tile_size=(6, 8, 8)):
    """ref: fastvideo_kernel/ops.py:21-62 — same positional arguments and defaults (``text_length`` required, ``has_text=True``,
    ``seq_shape="30x48x80"``).  q,k,v [B,H,S,D] bf16, image tokens first in tile-major order; ``window_size`` is a per-head list of
    (t,h,w) windows in tiles.  ``seq_shape`` is "TxHxW" (the reference's three canvases or any other canvas divisible by
    ``tile_size`` — the reference kernels hard-code theirs).

    ``has_text`` (HunyuanVideo / StepVideo): rows past the canvas are text tokens, the first ``text_length`` of them valid.  Mask rule of
    fastvideo-kernel/tests/support_flex_sta.py:52-55: an image query attends its window plus the valid text keys, a text query attends
    every image key plus the valid text keys.  As in the reference (ops.py:36-43) the sequence is padded up to whole 384-row tiles with
    copies of its last rows and the result sliced back.  ``has_text=False``: the sequence is exactly the canvas; ``text_length`` is ignored
    (the reference hands it to a kernel that does not read it then).

    Registered as ``torch.ops.fastvideo_kernel.sliding_tile_attention_gfx950`` (opaque to torch.compile, fake kernel = empty_like(q))."""
    _check_bf16(q, k, v)
    canvas = _STA_CANVAS.get(seq_shape) or tuple(int(x) for x in seq_shape.split("x"))
    if len(canvas) != 3 or any(c % t for c, t in zip(canvas, tile_size)):
        raise ValueError(f"canvas {canvas} is not divisible by tile {tuple(tile_size)}")
    if len(window_size) != q.shape[1]:
        raise ValueError(f"window_size must list one (t,h,w) per head ({q.shape[1]}), got {len(window_size)}")
    img = canvas[0] * canvas[1] * canvas[2]
    if has_text:
        if q.shape[2] < img or not 0 <= int(text_length) <= q.shape[2] - img:
            raise ValueError(f"sliding_tile_attention: {q.shape[2]} rows do not hold the {img}-token canvas {seq_shape} plus {text_length} text tokens")
    elif q.shape[2] != img:
        raise ValueError(f"sliding_tile_attention: {q.shape[2]} rows != the {img} tokens of canvas {seq_shape} (has_text=False)")
    flat = [int(x) for w in window_size for x in w]
    _check_cuda(q, k, v)
    return torch.ops.fastvideo_kernel.sliding_tile_attention_gfx950(q, k, v, flat, int(text_length) if has_text else -1, list(canvas),
                                                                    [int(t) for t in tile_size])


def _sta_impl(q, k, v, windows_flat, text_length, canvas, tile_size):
    """Body of the registered op.  text_length < 0: no text rows."""
    tok = math.prod(tile_size)
    tiles = tuple(c // t for c, t in zip(canvas, tile_size))
    windows = tuple(tuple(windows_flat[3 * i:3 * i + 3]) for i in range(len(windows_flat) // 3))
    if text_length < 0:
        return sliding_tile_attention_canvas(q, k, v, tiles, tok, windows, layout="bhsd")
    if tok < 256 or tok % 128:
        raise NotImplementedError(f"sliding_tile_attention with text rows needs tiles of 256 / 384 / 512 ... tokens (tile {tuple(tile_size)} holds {tok})")
    S = q.shape[2]
    pad = (-S) % tok
    if pad:  # ops.py:36-43: whole tiles, the filler is masked (keys) or sliced off (queries)
        q, k, v = (torch.cat([t, t[:, :, -pad:]], dim=2) for t in (q, k, v))
    img = math.prod(canvas)
    idx, num, sizes = _canvas_tile_lists(tiles, tok, windows, q.shape[0], q.device, text_rows=S + pad - img, text_length=int(text_length))
    o = ops.attn_tile_lists(q.contiguous(), k.contiguous(), v.contiguous(), idx, num, sizes, tok, None, layout="bhsd")
    return o[:, :, :S] if pad else o


def sliding_tile_attention_canvas(q, k, v, tiles, tile_tokens, window_size, layout="bhsd"):
    """Sliding-tile attention over tile-major tokens of a whole-tile canvas (``tiles`` = tiles per axis), either layout.  Tiles of >= 256
    tokens: one KV block list per (head, tile), 256 of a tile's query rows per workgroup on the dense kernel's schedule
    (fvk_attn_tile_lists_bf16; 907 -> 1030 TF on 18x48x80 against the per-128-row canvas kernel, profiles/r02_sta_lists_ab.md);
    smaller tiles: the canvas kernel (fvk_attn_sta_bf16)."""
    tok = int(tile_tokens)
    windows = tuple(tuple(int(x) for x in w) for w in window_size)
    if tok >= 256 and tok % 128 == 0 and max(w[0] * w[1] * w[2] for w in windows) * (tok // 64) <= 4096:
        idx, num, sizes = _canvas_tile_lists(tuple(tiles), tok, windows, q.shape[0], q.device)
        return ops.attn_tile_lists(q, k, v, idx, num, sizes, tok, None, layout=layout)
    return ops.attn_sta(q, k, v, tiles, tok, window_size, layout=layout)


_CANVAS_LISTS = {}


def _canvas_tile_lists(tiles, tok, windows, batch, device, text_rows=0, text_length=0):
    """Per-(head, tile) KV lists of 64-token blocks for a whole-tile canvas (every block full), cached per geometry on the device.
    Window rule: fastvideo-kernel/tests/support_flex_sta.py:44-51 (clamped centre, integer k//2; identical to fvk_attn_sta_bf16's).
    ``text_rows`` (a multiple of ``tok``) rows follow the canvas, the first ``text_length`` valid (support_flex_sta.py:52-55): their
    64-row blocks join every image tile's list with their valid-key counts, and the text rows form extra query tiles whose list is
    every image block plus the valid text blocks."""
    key = (tiles, tok, windows, batch, str(device), text_rows, text_length)
    hit = _CANVAS_LISTS.get(key)
    if hit is not None:
        return hit
    import numpy as np
    sub = tok // 64
    n_tiles = tiles[0] * tiles[1] * tiles[2]

    def win(q_, n, k_):
        c = min(max(q_, k_ // 2), (n - 1) - k_ // 2)
        return range(max(c - k_ // 2, 0), min(c + k_ // 2 + 1, n))

    per_head = {}
    for w in set(windows):
        lists = [[((x * tiles[1] + y) * tiles[2] + z) * sub + s_ for x in win(a, tiles[0], w[0]) for y in win(b, tiles[1], w[1])
                  for z in win(c, tiles[2], w[2]) for s_ in range(sub)]
                 for a in range(tiles[0]) for b in range(tiles[1]) for c in range(tiles[2])]
        per_head[w] = lists
    sizes = np.full((n_tiles * sub + text_rows // 64,), 64, dtype=np.int32)
    if text_rows:
        if text_rows % tok:
            raise ValueError(f"text rows ({text_rows}) must fill whole {tok}-row tiles")
        sizes[n_tiles * sub:] = np.clip(text_length - 64 * np.arange(text_rows // 64), 0, 64)
        text_blocks = [n_tiles * sub + j for j in range(text_rows // 64) if sizes[n_tiles * sub + j] > 0]
        text_query = list(range(n_tiles * sub)) + text_blocks
        for w in per_head:
            per_head[w] = [l + text_blocks for l in per_head[w]] + [text_query] * (text_rows // tok)
    n_lists = n_tiles + text_rows // tok
    mx = max(len(l) for ls in per_head.values() for l in ls)
    idx = np.zeros((len(windows), n_lists, mx), dtype=np.int32)
    num = np.zeros((len(windows), n_lists), dtype=np.int32)
    for h_, w in enumerate(windows):
        for t, l in enumerate(per_head[w]):
            idx[h_, t, :len(l)], num[h_, t] = l, len(l)
    out = (torch.from_numpy(idx).to(device)[None].expand(batch, -1, -1, -1).contiguous(),
           torch.from_numpy(num).to(device)[None].expand(batch, -1, -1).contiguous(),
           torch.from_numpy(sizes).to(device))
    if len(_CANVAS_LISTS) >= 16:
        _CANVAS_LISTS.clear()
    _CANVAS_LISTS[key] = out
    return out


def sliding_tile_block_lists(grid, tile_size=(6, 8, 8), window=(3, 3, 3), group_order="first_tile"):
    """Host-side (CPU, pure integer) index construction that turns sliding-tile attention on an ARBITRARY token grid into block-sparse
    attention: the grid is padded up to whole tiles, tokens are gathered tile-major with each tile's real tokens first, and the
    clamped-centre window rule (fastvideo-kernel/tests/support_flex_sta.py:44-51) selects, per query block, the 64-token KV blocks of
    the tiles in the window — empty blocks dropped, partially filled ones carried with their size.
    Returns a dict of CPU tensors: tile_partition_indices / non_pad_index / untile_combined_index (as build_vsa_metadata),
    block_sizes int32 [n_blocks64], q2k_idx int32 [n_qblocks, max_kv] (ascending), q2k_num int32 [n_qblocks], q_block (64 or 128 query
    rows per list), S_pad, num_tiles; and the per-TILE form of the same lists for ``ops.attn_tile_lists`` (tiles of >= 256 tokens):
    tile_q2k_idx int32 [n_tiles, max_kv], tile_q2k_num int32 [n_tiles], tile_rows_valid int32 [n_tiles], tile_tokens; and the
    query-grouped form (see below): group_src / group_dst / group_untile int32 [S], group_rows, group_q2k_idx int32 [n_groups, max_kv],
    group_q2k_num int32 [n_groups], n_window_classes."""
    import numpy as np
    tok = tile_size[0] * tile_size[1] * tile_size[2]
    if tok % 64:
        raise ValueError(f"sliding-tile tile {tuple(tile_size)} must hold a multiple of 64 tokens")
    qb = 128 if tok % 128 == 0 else 64  # every query block of a tile shares the tile's window
    h = ops.vsa_build_metadata_host(tuple(grid), tuple(tile_size))
    nt = h["num_tiles"]
    sub = tok // 64
    vbs = h["variable_block_sizes"].numpy()
    bsz = np.clip(vbs[:, None] - 64 * np.arange(sub)[None, :], 0, 64).astype(np.int32).reshape(-1)

    def win(q, n, k):
        c = min(max(q, k // 2), (n - 1) - k // 2)
        return range(max(c - k // 2, 0), min(c + k // 2 + 1, n))

    lists, tile_lists, tile_class = [], [], []
    for a in range(nt[0]):
        for b in range(nt[1]):
            for c in range(nt[2]):
                tiles = [(x * nt[1] + y) * nt[2] + z for x in win(a, nt[0], window[0]) for y in win(b, nt[1], window[1])
                         for z in win(c, nt[2], window[2])]
                blocks = [t * sub + s_ for t in tiles for s_ in range(sub) if bsz[t * sub + s_] > 0]
                # a tile's real tokens come first: query lists that hold only padding rows (edge tiles of a ragged grid: 13 % of the
                # lists at 21x30x52) get an EMPTY KV list — the kernel then does nothing for them and their rows are dropped by untile
                q_tile = (a * nt[1] + b) * nt[2] + c
                real = int(vbs[q_tile])
                lists += [blocks if real > i * qb else [] for i in range(tok // qb)]
                tile_lists.append(blocks)
                tile_class.append(tuple(tiles))
    mx = max(len(l) for l in lists)
    idx = np.zeros((len(lists), mx), dtype=np.int32)
    num = np.zeros((len(lists),), dtype=np.int32)
    for i, l in enumerate(lists):
        idx[i, :len(l)], num[i] = l, len(l)
    # the same lists once per TILE (all of a tile's query rows share them) for fvk_attn_tile_lists_bf16, which runs 256 of a tile's rows per
    # workgroup: tile_rows_valid = real tokens of the tile (its row groups that start past it are skipped)
    t_idx = np.zeros((len(tile_lists), mx), dtype=np.int32)
    t_num = np.zeros((len(tile_lists),), dtype=np.int32)
    for i, l in enumerate(tile_lists):
        t_idx[i, :len(l)], t_num[i] = l, len(l)
    # QUERY rows grouped by window class: the clamped-centre rule gives many tiles the SAME window (edge tiles share their neighbour's:
    # 20 distinct windows for the 112 tiles of 21x30x52), and a query's output depends only on its KV set — so the real query tokens of
    # all tiles of a class are packed back to back (no per-tile padding) and cut into 256-row groups that share the class's list.  K / V
    # keep the tile-major layout.  group_src[i] = source token of the i-th packed row, group_dst[i] = its row, group_untile[t] = row of
    # token t, group_rows = padded row count (a multiple of 256), group_q2k_idx / _num = one list per 256-row group.
    tile_off = np.concatenate([[0], np.cumsum(vbs)]).astype(np.int64)
    perm = h["tile_partition_indices"].numpy()
    classes = {}
    for t, key in enumerate(tile_class):
        classes.setdefault(key, []).append(t)
    g_src, g_dst, g_lists, row0 = [], [], [], 0
    items = list(classes.items())
    if group_order == "longest_first":  # measurement: longest KV lists first
        items.sort(key=lambda kv: -len(tile_lists[kv[1][0]]))
    for key, tiles_c in items:
        toks = np.concatenate([perm[tile_off[t]:tile_off[t + 1]] for t in tiles_c])
        if len(toks) == 0:
            continue
        n256 = -(-len(toks) // 256)
        g_src.append(toks)
        g_dst.append(row0 + np.arange(len(toks)))
        g_lists += [tile_lists[tiles_c[0]]] * n256
        row0 += n256 * 256
    g_src, g_dst = np.concatenate(g_src).astype(np.int32), np.concatenate(g_dst).astype(np.int32)
    g_untile = np.empty(len(g_src), dtype=np.int32)
    g_untile[g_src] = g_dst
    gmx = max(len(l) for l in g_lists)
    g_idx = np.zeros((len(g_lists), gmx), dtype=np.int32)
    g_num = np.zeros((len(g_lists),), dtype=np.int32)
    for i, l in enumerate(g_lists):
        g_idx[i, :len(l)], g_num[i] = l, len(l)
    n_tok = grid[0] * grid[1] * grid[2]
    return dict(group_src=torch.from_numpy(g_src), group_dst=torch.from_numpy(g_dst), group_untile=torch.from_numpy(g_untile), group_rows=row0,
                group_q2k_idx=torch.from_numpy(g_idx), group_q2k_num=torch.from_numpy(g_num), n_window_classes=len(classes),
                tile_q2k_idx=torch.from_numpy(t_idx), tile_q2k_num=torch.from_numpy(t_num),
                tile_rows_valid=torch.from_numpy(vbs.astype(np.int32)), tile_tokens=tok, tile_partition_indices=h["tile_partition_indices"], non_pad_index=h["non_pad_index"],
                untile_combined_index=h["untile_combined_index"], block_sizes=torch.from_numpy(bsz), q2k_idx=torch.from_numpy(idx),
                q2k_num=torch.from_numpy(num), q_block=qb, S_pad=len(vbs) * tok, num_tiles=nt,
                density=float(sum(int(bsz[i * (qb // 64):(i + 1) * (qb // 64)].sum()) * sum(int(bsz[b]) for b in l)
                                  for i, l in enumerate(lists))) / float(n_tok)**2)  # attended (query, key) pairs / S^2


# ------------------------------------------------------------------ torch.library registration
# The reference wraps every kernel entry in torch.library.custom_op + register_fake (fastvideo_kernel/block_sparse_attn.py:103-145,224-267) so
# that torch.compile (component_loader.py:1127 ``enable_torch_compile``) sees ONE opaque node per kernel instead of graph-breaking on the
# foreign call.  Same here: the ctypes calls into libfvk_amd.so live inside three ops in the reference's own ``fastvideo_kernel`` namespace,
# named like its per-backend ops (``_triton`` / ``_sm90`` -> ``_gfx950``); the fake kernels give shapes / dtypes only.  device_types="cuda"
# (ROCm tensors are "cuda" tensors): a CPU tensor is refused by the dispatcher, there is no CPU fallback.  Forward only.
@torch.library.custom_op("fastvideo_kernel::block_sparse_attn_gfx950", mutates_args=(), device_types="cuda")
def _block_sparse_attn_op(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, q2k_idx: torch.Tensor, q2k_num: torch.Tensor,
                          variable_block_sizes: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor]:
    return ops.attn_block_sparse(q, k, v, q2k_idx.int(), q2k_num.int(), variable_block_sizes.int(), layout="bhsd", return_lse=True)


@torch.library.register_fake("fastvideo_kernel::block_sparse_attn_gfx950")
def _block_sparse_attn_fake(q, k, v, q2k_idx, q2k_num, variable_block_sizes):
    return torch.empty_like(q), torch.empty((q.shape[0], q.shape[1], q.shape[2]), device=q.device, dtype=torch.float32)


@torch.library.custom_op("fastvideo_kernel::video_sparse_attn_gfx950", mutates_args=(), device_types="cuda")
def _video_sparse_attn_op(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, variable_block_sizes: torch.Tensor,
                          q_variable_block_sizes: torch.Tensor, topk: int, block_elements: int, compress_attn_weight: torch.Tensor | None,
                          layout: str) -> torch.Tensor:
    return _vsa_forward(q, k, v, variable_block_sizes, q_variable_block_sizes, topk, compress_attn_weight, layout, False, block_elements)


@torch.library.register_fake("fastvideo_kernel::video_sparse_attn_gfx950")
def _video_sparse_attn_fake(q, k, v, variable_block_sizes, q_variable_block_sizes, topk, block_elements, compress_attn_weight, layout):
    return torch.empty_like(q)


@torch.library.custom_op("fastvideo_kernel::sliding_tile_attention_gfx950", mutates_args=(), device_types="cuda")
def _sliding_tile_attention_op(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, windows: list[int], text_length: int, canvas: list[int],
                               tile_size: list[int]) -> torch.Tensor:
    return _sta_impl(q, k, v, windows, text_length, canvas, tile_size)


@torch.library.register_fake("fastvideo_kernel::sliding_tile_attention_gfx950")
def _sliding_tile_attention_fake(q, k, v, windows, text_length, canvas, tile_size):
    return torch.empty_like(q)


__all__ = [
    "sliding_tile_attention", "video_sparse_attn", "video_sparse_attn_bshd", "block_sparse_attn", "block_sparse_attn_from_indices", "VSA_TILE_SIZE",
    "get_tile_partition_indices", "get_reverse_tile_partition_indices", "construct_variable_block_sizes",
    "get_non_pad_index", "build_vsa_metadata", "sliding_tile_block_lists",
]
