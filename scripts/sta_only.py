"""The shipped sliding-tile attention kernel alone at the cfg2 grid (queries packed by window class, 12 heads) — for PMC passes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fastvideo_amd import kernel_api, ops
H = 12
g = torch.Generator(device="cuda").manual_seed(0)
h = kernel_api.sliding_tile_block_lists((21, 30, 52), (6, 8, 8), (3, 3, 3))
q = torch.randn((1, h["group_rows"], H, 128), generator=g, device="cuda").bfloat16()
k, v = (torch.randn((1, h["S_pad"], H, 128), generator=g, device="cuda").bfloat16() for _ in range(2))
vt = ops.v_transpose(v)
ex = lambda t, n: t.cuda()[None, None].expand(1, H, *([-1] * n)).contiguous()
idx, num, bs = ex(h["group_q2k_idx"], 2), ex(h["group_q2k_num"], 1), h["block_sizes"].cuda()
for _ in range(int(os.environ.get("N_LAUNCH", "3"))):
    o = ops.attn_tile_lists(q, k, None, idx, num, bs, 256, None, layout="bshd", vt=vt)
torch.cuda.synchronize()
n_tok = 21 * 30 * 52
print("ok", float(o.float().abs().mean()), "algorithmic FLOP per launch", 4.0 * H * 128 * h["density"] * n_tok * n_tok,
      "staged K+V^T bytes per launch", float(((num + 1) // 2).sum()) * 2 * 128 * 128 * 2)
