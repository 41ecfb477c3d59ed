"""Sliding-tile attention over window-class query groups (fvk_attn_tile_lists_bf16; grid 21x30x52, tile (6,8,8), window (3,3,3), 12 heads):
attn_pp2's list mode (shipped, attn_impl 0) vs attn_w64's (attn_impl 72, measurement build only: see attn_fwd.hip) — interleaved timing and the difference of the outputs."""
import os as _os
_os.environ.setdefault("FVK_PROBE_LIB", "1")
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fastvideo_amd import kernel_api, ops
grid = tuple(int(x) for x in sys.argv[1:4]) if len(sys.argv) >= 4 else (21, 30, 52)
H = 12
g = torch.Generator(device="cuda").manual_seed(0)
h = kernel_api.sliding_tile_block_lists(grid, (6, 8, 8), (3, 3, 3))
q = torch.randn((1, h["group_rows"], H, 128), generator=g, device="cuda").bfloat16()
k, v = (torch.randn((1, h["S_pad"], H, 128), generator=g, device="cuda").bfloat16() for _ in range(2))
vt = ops.v_transpose(v)
ex = lambda t, n: t.cuda()[None, None].expand(1, H, *([-1] * n)).contiguous()
idx, num, bs = ex(h["group_q2k_idx"], 2), ex(h["group_q2k_num"], 1), h["block_sizes"].cuda()
fn = lambda: ops.attn_tile_lists(q, k, None, idx, num, bs, 256, None, layout="bshd", vt=vt)
res, outs = {}, {}
for r in range(4):
    for impl in (0, 72):
        ops.set_tunable("attn_impl", impl)
        outs[impl] = fn(); torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10): fn()
        e.record(); torch.cuda.synchronize()
        res.setdefault(impl, []).append(s.elapsed_time(e) / 10)
ops.set_tunable("attn_impl", 0)
n_tok = grid[0] * grid[1] * grid[2]
flops = 4.0 * H * 128 * h["density"] * n_tok * n_tok
d = (outs[72].float() - outs[0].float()).abs()
print(json.dumps({"grid": grid, "pp2_lists_ms": round(sorted(res[0])[1], 4), "w64_lists_ms": round(sorted(res[72])[1], 4),
                  "pp2_tflops": round(flops / sorted(res[0])[1] / 1e9, 1), "w64_tflops": round(flops / sorted(res[72])[1] / 1e9, 1),
                  "max_abs_diff": round(d.max().item(), 5), "mean_abs_diff": float(f"{d.mean().item():.3g}"), "finite": bool(torch.isfinite(outs[72].float()).all())}))
