"""Sliding-tile attention at the 81f x 480p grid (21,30,52), tile (6,8,8), window (3,3,3), 12 heads: one KV list per 128-row query block on
the 4-wave kernel (fvk_attn_block_sparse_bf16, the round-1 / early round-2 path) vs one list per 384-token TILE with 256-row workgroups on
the dense kernel's ping-pong schedule + a 128-row remainder (fvk_attn_tile_lists_bf16).  Prints ms per launch, algorithmic TFLOP/s and the
difference between the two outputs over the real (non-padding) rows."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fastvideo_amd import kernel_api, ops

grid = tuple(int(x) for x in sys.argv[1:4]) if len(sys.argv) >= 4 else (21, 30, 52)
h = kernel_api.sliding_tile_block_lists(grid, (6, 8, 8), (3, 3, 3))
H, S_pad = 12, h["S_pad"]
g = torch.Generator(device="cuda").manual_seed(0)
q, k, v = (torch.randn((1, S_pad, H, 128), generator=g, device="cuda").bfloat16() for _ in range(3))
ex = lambda t, n: t.cuda()[None, None].expand(1, H, *([-1] * n)).contiguous()
idx, num, bs = ex(h["q2k_idx"], 2), ex(h["q2k_num"], 1), h["block_sizes"].cuda()
tidx, tnum, tval = ex(h["tile_q2k_idx"], 2), ex(h["tile_q2k_num"], 1), h["tile_rows_valid"].cuda()
old = lambda: ops.attn_block_sparse(q, k, v, idx, num, bs, layout="bshd", q_block=h["q_block"])
new = lambda: ops.attn_tile_lists(q, k, v, tidx, tnum, bs, h["tile_tokens"], tval, layout="bshd")
res = {"old": [], "new": []}
for r in range(3):
    for name, fn in (("old", old), ("new", new)):
        fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(5):
            fn()
        e.record()
        torch.cuda.synchronize()
        res[name].append(s.elapsed_time(e) / 5)
a, b = old(), new()
# real rows only: padding rows of partially real 128-row groups are computed by both, fully padded groups are zeros in both
rows = torch.zeros(S_pad, dtype=torch.bool)
tok = h["tile_tokens"]
for t, nv in enumerate(h["tile_rows_valid"].tolist()):
    rows[t * tok:t * tok + nv] = True
err = (a[0, rows.cuda()].float() - b[0, rows.cuda()].float()).abs()
n_tok = grid[0] * grid[1] * grid[2]
flops = 4.0 * H * 128 * h["density"] * n_tok * n_tok
out = {"grid": grid, "S_pad": S_pad, "density": round(h["density"], 4)}
for name in ("old", "new"):
    ms = sorted(res[name])[1]
    out[name] = {"ms": round(ms, 3), "tflops": round(flops / ms / 1e9, 1)}
out["max_abs_diff_real_rows"] = round(err.max().item(), 5)
out["mean_abs_diff_real_rows"] = float(f"{err.mean().item():.3g}")
print(json.dumps(out))
# the query-grouped form: queries packed by window class (kernel alone; the model swaps its q gather / untile index arrays, same cost)
gq = torch.zeros((1, h["group_rows"], H, 128), dtype=torch.bfloat16, device="cuda")
raster = torch.empty((1, n_tok, H, 128), dtype=torch.bfloat16, device="cuda")
perm, non_pad = h["tile_partition_indices"].cuda().long(), h["non_pad_index"].cuda().long()
raster[0, perm] = q[0, non_pad]                       # the un-tiled q that the tile-major q above came from
gq[0, h["group_dst"].cuda().long()] = raster[0, h["group_src"].cuda().long()]
gidx, gnum = ex(h["group_q2k_idx"], 2), ex(h["group_q2k_num"], 1)
grouped = lambda: ops.attn_tile_lists(gq, k, v, gidx, gnum, bs, 256, None, layout="bshd")
ts = []
for r in range(3):
    grouped()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5):
        grouped()
    e.record()
    torch.cuda.synchronize()
    ts.append(s.elapsed_time(e) / 5)
og = grouped()[0, h["group_untile"].cuda().long()]   # raster order
ot = torch.empty_like(raster)
ot[0, perm] = b[0, non_pad]
eg = (og.float() - ot[0].float()).abs()
ms = sorted(ts)[1]
print(json.dumps({"grouped": {"ms": round(ms, 3), "tflops": round(flops / ms / 1e9, 1), "rows": h["group_rows"], "window_classes": h["n_window_classes"],
                              "max_abs_diff_vs_tile_form": round(eg.max().item(), 5), "mean_abs_diff": float(f"{eg.mean().item():.3g}")}}))
# split of the new path: the 256-row part alone (remainder groups skipped through q_rows_valid)
tval256 = torch.clamp(tval, max=256)
only256 = lambda: ops.attn_tile_lists(q, k, v, tidx, tnum, bs, h["tile_tokens"], tval256, layout="bshd")
ts = []
for r in range(3):
    only256()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5):
        only256()
    e.record()
    torch.cuda.synchronize()
    ts.append(s.elapsed_time(e) / 5)
print(json.dumps({"new_256_row_part_only_ms": round(sorted(ts)[1], 3)}))
