"""CPU restatement of the reference's video-sparse-attention (VSA) and sliding-tile-attention
(STA) algorithms: index/mask construction in numpy (bit-exact targets) and attention in torch.

TEST INFRASTRUCTURE ONLY (see oracle/wan_oracle.py header for the import rule).

The reference's *kernels* for this path (Triton / ThunderKittens in ``fastvideo-kernel``) cannot
run in the build container (Triton needs a GPU driver at import; TK needs nvcc), so
the integer pieces are pinned against (a) the standalone-loadable
``fastvideo-kernel/python/fastvideo_kernel/vsa_utils.py`` and
``fastvideo/attention/backends/video_sparse_attn.py`` run here, and (b) the known-answer tests of
``fastvideo-kernel/tests/test_vsa_utils.py`` / ``test_fused_compress_topk.py``; the attention
numerics follow the dense fp32 masked oracle those tests themselves use.
"""
from __future__ import annotations

import math

import numpy as np
import torch

VSA_TILE_SIZE = (4, 4, 4)  # fastvideo/attention/backends/video_sparse_attn.py:28

# ------------------------------------------------------------------ index construction


def tile_partition_indices(dit_seq_shape, tile_size=VSA_TILE_SIZE) -> np.ndarray:
    """raster → tile-major permutation; video_sparse_attn.py:31-48 ≡ vsa_utils.py:30-50."""
    T, H, W = dit_seq_shape
    ts, hs, ws = tile_size
    idx = np.arange(T * H * W, dtype=np.int64).reshape(T, H, W)
    out = []
    for t in range(math.ceil(T / ts)):
        for h in range(math.ceil(H / hs)):
            for w in range(math.ceil(W / ws)):
                out.append(idx[t * ts:min(t * ts + ts, T), h * hs:min(h * hs + hs, H),
                               w * ws:min(w * ws + ws, W)].reshape(-1))
    return np.concatenate(out)


def reverse_tile_partition_indices(dit_seq_shape, tile_size=VSA_TILE_SIZE) -> np.ndarray:
    """``torch.argsort(perm)`` — video_sparse_attn.py:51-57 (perm is a permutation → unique)."""
    return np.argsort(tile_partition_indices(dit_seq_shape, tile_size), kind="stable").astype(np.int64)


def num_tiles_of(dit_seq_shape, tile_size=VSA_TILE_SIZE):
    return tuple(math.ceil(s / t) for s, t in zip(dit_seq_shape, tile_size))


def variable_block_sizes(dit_seq_shape, num_tiles=None, tile_size=VSA_TILE_SIZE) -> np.ndarray:
    """valid tokens per tile — video_sparse_attn.py:60-101 (int32 like ``torch.int``)."""
    num_tiles = num_tiles or num_tiles_of(dit_seq_shape, tile_size)

    def sizes(dim_len, tile, n):
        s = np.full((n, ), tile, dtype=np.int32)
        rem = dim_len - (n - 1) * tile
        s[-1] = rem if rem > 0 else tile
        return s

    t, h, w = (sizes(d, ts, n) for d, ts, n in zip(dit_seq_shape, tile_size, num_tiles))
    return (t[:, None, None] * h[None, :, None] * w[None, None, :]).reshape(-1)


def non_pad_index(vbs: np.ndarray, max_block_size: int) -> np.ndarray:
    """video_sparse_attn.py:104-114."""
    n = vbs.shape[0]
    pad = np.arange(n, dtype=np.int64)[:, None] * max_block_size + np.arange(max_block_size, dtype=np.int64)[None, :]
    mask = np.arange(max_block_size)[None, :] < vbs[:, None]
    return pad[mask]


def build_metadata(raw_latent_shape, patch_size=(1, 2, 2), tile_size=VSA_TILE_SIZE) -> dict:
    """``VideoSparseAttentionMetadataBuilder.build`` — video_sparse_attn.py:200-235."""
    shape = tuple(r // p for r, p in zip(raw_latent_shape, patch_size))
    nt = num_tiles_of(shape, tile_size)
    perm = tile_partition_indices(shape, tile_size)
    rev = reverse_tile_partition_indices(shape, tile_size)
    vbs = variable_block_sizes(shape, nt, tile_size)
    npi = non_pad_index(vbs, math.prod(tile_size))
    return dict(dit_seq_shape=shape, num_tiles=nt, total_seq_length=math.prod(shape),
                tile_partition_indices=perm, reverse_tile_partition_indices=rev,
                variable_block_sizes=vbs, non_pad_index=npi, untile_combined_index=npi[rev])


def compute_topk(sparsity: float, num_blocks: int) -> int:
    """video_sparse_attn.py:161-163 — uses the *padded* block count, clamped to [1, n]."""
    return max(1, min(math.ceil((1 - sparsity) * num_blocks), num_blocks))


# ------------------------------------------------------------------ tile / untile


def tile(x: torch.Tensor, meta: dict) -> torch.Tensor:
    """zero-padded scatter — video_sparse_attn.py:170-189, 254-283.  x [B,S,H,D]."""
    n_pad = math.prod(meta["num_tiles"]) * math.prod(VSA_TILE_SIZE)
    buf = torch.zeros((x.shape[0], n_pad, x.shape[2], x.shape[3]), dtype=x.dtype)
    buf[:, torch.from_numpy(meta["non_pad_index"])] = x[:, torch.from_numpy(meta["tile_partition_indices"])]
    return buf


def untile(x: torch.Tensor, meta: dict) -> torch.Tensor:
    """video_sparse_attn.py:285-290."""
    return x[:, torch.from_numpy(meta["untile_combined_index"])]


# ------------------------------------------------------------------ compress / top-k / index


def block_mean(x: torch.Tensor, vbs: np.ndarray, block_elements: int) -> torch.Tensor:
    """``_fused_block_mean_kernel`` — triton_kernels/fused_compress_topk.py:22-60: fp32 sum over the
    (zero-padded) block ÷ valid count, cast to x.dtype.  x [B,H,S_pad,D]."""
    B, H, S, D = x.shape
    nb = S // block_elements
    s = x.float().view(B, H, nb, block_elements, D).sum(dim=3)
    return (s / torch.from_numpy(vbs.astype(np.float32)).view(1, 1, nb, 1)).to(x.dtype)


def topk_mask_bisect(scores: np.ndarray, topk: int) -> np.ndarray:
    """``_fused_topk_mask_kernel`` — triton_kernels/fused_compress_topk.py:211-277, restated
    operation-for-operation in fp32 (32 bisection steps on the threshold, then first-come
    tie-break by cumulative count).  scores [..., kv_blocks] → bool mask, same shape."""
    sc = scores.astype(np.float32)
    flat = sc.reshape(-1, sc.shape[-1])
    topk = min(topk, flat.shape[-1])
    out = np.zeros(flat.shape, dtype=bool)
    for r in range(flat.shape[0]):
        row = flat[r]
        finite = row > -np.inf
        lo = np.float32(row[finite].min()) if finite.any() else np.float32(np.inf)
        hi = np.float32(row.max())
        lo = np.float32(min(lo, hi))
        for _ in range(32):
            mid = np.float32(np.float32(lo + hi) * np.float32(0.5))
            if int((row >= mid).sum()) >= topk:
                lo = mid
            else:
                hi = mid
        above = row > lo
        at = row == lo
        need = topk - int(above.sum())
        sel = at & (np.cumsum(at.astype(np.int32)) <= need)
        out[r] = above | sel
    return out.reshape(scores.shape)


def map_to_index(block_map: np.ndarray):
    """``map_to_index_kernel`` — triton_kernels/index.py:33-61: ascending compaction.
    block_map bool [B,H,Nq,Nkv] → (q2k_idx int32 [B,H,Nq,Nkv] zero-filled tail, q2k_num int32 [B,H,Nq])."""
    B, H, Nq, Nkv = block_map.shape
    idx = np.zeros((B, H, Nq, Nkv), dtype=np.int32)
    num = block_map.sum(-1).astype(np.int32)
    for b in range(B):
        for h in range(H):
            for q in range(Nq):
                nz = np.nonzero(block_map[b, h, q])[0]
                idx[b, h, q, :len(nz)] = nz
    return idx, num


# ------------------------------------------------------------------ attention


def block_sparse_attn(q, k, v, block_map: np.ndarray, vbs: np.ndarray, block: int = 64) -> torch.Tensor:
    """Dense fp32 masked restatement of ``_attn_fwd_sparse`` (block_sparse_attn_triton.py:32-160):
    a query block attends the KV blocks selected in ``block_map``; within a KV block only the first
    ``vbs[j]`` columns are valid (``:133-134``); scale 1/sqrt(D).  q,k,v [B,H,S_pad,D] → fp32."""
    B, H, S, D = q.shape
    Skv = k.shape[2]
    nq, nk = S // block, Skv // block
    bm = torch.from_numpy(block_map).view(B, H, nq, 1, nk, 1).expand(B, H, nq, block, nk, block)
    col_ok = (torch.arange(block)[None, :] < torch.from_numpy(vbs.astype(np.int64))[:, None])  # [nk, block]
    mask = (bm & col_ok.view(1, 1, 1, 1, nk, block)).reshape(B, H, S, Skv)
    s = torch.matmul(q.float(), k.float().transpose(-1, -2)) * (D**-0.5)
    s = s.masked_fill(~mask, float("-inf"))
    return torch.matmul(torch.softmax(s, dim=-1), v.float())


def block_sparse_attn_gathered(q, k, v, block_map: np.ndarray, vbs: np.ndarray, block: int = 64, q_blocks=None) -> torch.Tensor:
    """The SAME arithmetic as ``block_sparse_attn`` (exact fp32 softmax over the valid columns of the selected KV blocks, scale
    1/sqrt(D)), evaluated one query block at a time over ITS selected blocks only, so that the real geometry (S_pad = 39 936, 12 heads,
    top-125 of 624 blocks) needs megabytes instead of the 76 GB of a dense [B,H,S,S] score tensor.  Equal to the dense-mask form up to
    fp32 summation order (tests/test_oracle_chunked.py).  q,k,v [B,H,S_pad,D] -> fp32.  ``q_blocks``: evaluate only these query blocks
    (the other rows of the result stay zero) — the sampled form for BASELINE config 5's 2 160 blocks (tests/test_gpu_bigseq.py)."""
    B, H, S, D = q.shape
    nq = S // block
    out = torch.zeros((B, H, S, D), dtype=torch.float32)
    vb = torch.from_numpy(vbs.astype(np.int64))
    scale = D**-0.5
    kf, vf = k.float(), v.float()
    ar = torch.arange(block)
    for b in range(B):
        for h in range(H):
            kb = kf[b, h].view(-1, block, D)
            vbk = vf[b, h].view(-1, block, D)
            for i in (range(nq) if q_blocks is None else q_blocks):
                sel = torch.from_numpy(np.nonzero(block_map[b, h, i])[0])
                if sel.numel() == 0:
                    continue
                ok = (ar[None, :] < vb[sel][:, None]).reshape(-1)                    # valid columns of the selected blocks
                ks, vs = kb[sel].reshape(-1, D)[ok], vbk[sel].reshape(-1, D)[ok]
                sc = (q[b, h, i * block:(i + 1) * block].float() @ ks.T) * scale
                out[b, h, i * block:(i + 1) * block] = torch.softmax(sc, dim=-1) @ vs
    return out


def video_sparse_attn(q, k, v, vbs, q_vbs, topk: int, block_elements: int = 64, gate=None, mask_override=None, gathered: bool = False,
                      q_blocks=None):
    """``video_sparse_attn`` — fastvideo-kernel/python/fastvideo_kernel/ops.py:65-133.
    q,k,v(,gate) [B,H,S_pad,D] bf16.  Returns (out bf16, dict of intermediates).
    ``mask_override`` (bool [B,H,Nq,Nkv]) replaces the top-k selection: tests feed the mask the device computed from ITS coarse
    scores (a bf16 ulp in one score can flip a near-tie), so that the composite is compared block for block.  ``gathered``: evaluate
    the sparse branch block by block (block_sparse_attn_gathered) — the form that fits the real geometry in memory; ``q_blocks`` (gathered
    form only): the sparse branch on these query blocks only (the rows of every other block hold the coarse branch alone)."""
    B, H, S, D = q.shape
    q_c = block_mean(q, q_vbs, block_elements)
    k_c = block_mean(k, vbs, block_elements)
    v_c = block_mean(v, vbs, block_elements)
    scores = torch.matmul(q_c, k_c.transpose(-2, -1)) / (D**0.5)
    attn = torch.softmax(scores, dim=-1)
    out_c = torch.matmul(attn, v_c)
    out_c = out_c.view(B, H, S // block_elements, 1, D).repeat(1, 1, 1, block_elements, 1).view(B, H, S, D)
    mask = topk_mask_bisect(scores.float().numpy(), topk) if mask_override is None else mask_override
    if q_blocks is not None:
        assert gathered, "q_blocks: gathered form only"
        out_s = block_sparse_attn_gathered(q, k, v, mask, vbs, block_elements, q_blocks=q_blocks).to(q.dtype)
    else:
        out_s = (block_sparse_attn_gathered if gathered else block_sparse_attn)(q, k, v, mask, vbs, block_elements).to(q.dtype)
    out = out_c * gate + out_s if gate is not None else out_c + out_s
    return out, dict(q_c=q_c, k_c=k_c, v_c=v_c, scores=scores, mask=mask, out_c=out_c, out_s=out_s)


# ------------------------------------------------------------------ STA


def sta_window(q_tile, n_tiles, kernel):
    """Clamped-centre window along one axis — fastvideo-kernel/tests/support_flex_sta.py:44-51 ≡
    st_attn_triton.py:158-176.  Returns [start, end) in tile coordinates."""
    c = min(max(q_tile, kernel // 2), (n_tiles - 1) - kernel // 2)
    return max(c - kernel // 2, 0), min(c + kernel // 2 + 1, n_tiles)  # ∩ canvas (mask semantics)


def sta_tile_lists(canvas_tiles, kernel) -> list[list[int]]:
    """For every query tile (raster t,h,w over tiles) the ascending list of KV tile ids it attends."""
    nt, nh, nw = canvas_tiles
    out = []
    for t in range(nt):
        t0, t1 = sta_window(t, nt, kernel[0])
        for h in range(nh):
            h0, h1 = sta_window(h, nh, kernel[1])
            for w in range(nw):
                w0, w1 = sta_window(w, nw, kernel[2])
                out.append([(a * nh + b) * nw + c for a in range(t0, t1) for b in range(h0, h1)
                            for c in range(w0, w1)])
    return out


def sta_mask(canvas_thw, kernel, tile_thw, text_length: int = 0, total_len: int | None = None) -> torch.Tensor:
    """Boolean [Sq,Skv] mask of ``generate_sta_mask`` — support_flex_sta.py:11-59 (tokens are in
    tile-major order, tile = idx // tile_volume)."""
    tv = math.prod(tile_thw)
    ct = tuple(c // t for c, t in zip(canvas_thw, tile_thw))
    img = math.prod(canvas_thw)
    total = total_len or (img + text_length)
    idx = torch.arange(total)

    def txy(i):
        tid = i // tv
        return tid // (ct[1] * ct[2]), (tid % (ct[1] * ct[2])) // ct[2], tid % ct[2]

    qt, qx, qy = txy(idx[:, None])
    kt, kx, ky = txy(idx[None, :])
    c_t = qt.clamp(kernel[0] // 2, (ct[0] - 1) - kernel[0] // 2)
    c_x = qx.clamp(kernel[1] // 2, (ct[1] - 1) - kernel[1] // 2)
    c_y = qy.clamp(kernel[2] // 2, (ct[2] - 1) - kernel[2] // 2)
    m = ((c_t - kt).abs() <= kernel[0] // 2) & ((c_x - kx).abs() <= kernel[1] // 2) & ((c_y - ky).abs() <= kernel[2] // 2)
    q_img, k_img = idx[:, None] < img, idx[None, :] < img
    img2txt = q_img & (idx[None, :] >= img) & (idx[None, :] < img + text_length)
    txt2all = (idx[:, None] >= img) & (idx[None, :] < img + text_length)
    return (q_img & k_img & m) | img2txt | txt2all


def sta_mask_ragged(grid_thw, kernel, tile_thw) -> torch.Tensor:
    """Boolean [S,S] sliding-tile mask for a token grid that is NOT a whole number of tiles, tokens in RASTER order (t,h,w):
    token i may attend token j iff j's tile lies in the clamped-centre window (sta_window, per axis, on the tile grid of the
    canvas padded up to whole tiles) of i's tile; padding tokens do not exist.  For grids divisible by the tile this is sta_mask
    after the tile permutation.  The reference has no kernel for such canvases (its kernels hard-code three canvases);
    this is the natural extension that the 81f x 480p grid (21,30,52) with tile (6,8,8) needs (BASELINE config 3)."""
    T, H, W = grid_thw
    nt = tuple(-(-g // t) for g, t in zip(grid_thw, tile_thw))
    t_idx = torch.arange(T).view(T, 1, 1).expand(T, H, W).reshape(-1) // tile_thw[0]
    h_idx = torch.arange(H).view(1, H, 1).expand(T, H, W).reshape(-1) // tile_thw[1]
    w_idx = torch.arange(W).view(1, 1, W).expand(T, H, W).reshape(-1) // tile_thw[2]
    m = torch.ones((T * H * W, T * H * W), dtype=torch.bool)
    for idx, n, k in ((t_idx, nt[0], kernel[0]), (h_idx, nt[1], kernel[1]), (w_idx, nt[2], kernel[2])):
        win = torch.tensor([sta_window(q, n, k) for q in range(n)])      # [n, 2] = [start, end) per query tile coordinate
        lo, hi = win[idx, 0], win[idx, 1]
        m &= (idx[None, :] >= lo[:, None]) & (idx[None, :] < hi[:, None])
    return m


def sta_attention_ragged(q, k, v, scale: float, grid_thw, kernel, tile_thw) -> torch.Tensor:
    """Exact fp32 attention under ``sta_mask_ragged(grid_thw, kernel, tile_thw)`` without building the [S,S] mask or score tensor: every
    query TILE (all of its tokens share one window) attends the tokens whose tile lies in its clamped-centre window.  Tokens in raster
    order (t,h,w).  Equal to wan_oracle.attention_fp32_ref(q, k, v, scale, sta_mask_ragged(...)) up to fp32 summation order
    (tests/test_oracle_chunked.py); fits BASELINE config 3's real geometry (grid (21,30,52), tile (6,8,8): 112 tiles of <= 384 tokens,
    windows of <= 27 tiles).  q,k,v [B,H,S,D] -> fp32 [B,H,S,D]."""
    T, H_, W_ = grid_thw
    nt = tuple(-(-g // t) for g, t in zip(grid_thw, tile_thw))
    t_idx = torch.arange(T).view(T, 1, 1).expand(T, H_, W_).reshape(-1) // tile_thw[0]
    h_idx = torch.arange(H_).view(1, H_, 1).expand(T, H_, W_).reshape(-1) // tile_thw[1]
    w_idx = torch.arange(W_).view(1, 1, W_).expand(T, H_, W_).reshape(-1) // tile_thw[2]
    qf, kf, vf = q.float(), k.float(), v.float()
    out = torch.zeros_like(qf)
    for a in range(nt[0]):
        t0, t1 = sta_window(a, nt[0], kernel[0])
        for b in range(nt[1]):
            h0, h1 = sta_window(b, nt[1], kernel[1])
            for c in range(nt[2]):
                w0, w1 = sta_window(c, nt[2], kernel[2])
                rows = torch.nonzero((t_idx == a) & (h_idx == b) & (w_idx == c)).flatten()
                if rows.numel() == 0:
                    continue
                keys = torch.nonzero((t_idx >= t0) & (t_idx < t1) & (h_idx >= h0) & (h_idx < h1) & (w_idx >= w0) & (w_idx < w1)).flatten()
                sc = torch.matmul(qf[:, :, rows], kf[:, :, keys].transpose(-1, -2)) * scale
                out[:, :, rows] = torch.matmul(torch.softmax(sc, dim=-1), vf[:, :, keys])
    return out
