"""fvk_qkv_norm_rope_pack2_bf16 (two head-chunk send buffers) against fvk_qkv_norm_rope_pack_bf16 (one buffer): the chunks must be the columns."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fastvideo_amd import ops, rope
torch.manual_seed(0)
for (H, G, U, Sl, S, pos0) in ((12, 2, 1, 1628, 3255, 1628), (6, 3, 1, 225, 675, 450), (6, 2, 1, 338, 675, 0), (12, 4, 2, 4095, 32760, 8190)):
    D, d = 128, H * 128
    buf = torch.randn((Sl, 3 * d), device="cuda").bfloat16()
    wq, wk = (1 + 0.1 * torch.randn(d, device="cuda")).bfloat16(), (1 + 0.1 * torch.randn(d, device="cuda")).bfloat16()
    grid = (S // 15 // 9 if False else 1, 1, S)  # any grid with S positions
    cos = torch.randn((S + 8, D), device="cuda"); sin = torch.randn((S + 8, D), device="cuda")
    hg = H // G
    ha = (hg + 1) // 2
    args = (buf[:, :d], buf[:, d:2 * d], buf[:, 2 * d:], wq, wk, cos, sin, G, U)
    kw = dict(head_dim=D, seq_len=S, eps=1e-6, pos_offset=pos0)
    full = ops.qkv_norm_rope_pack(*args, **kw)                      # [P, Sl, 3, hg*D]
    outs = [ops.qkv_norm_rope_pack(*args, heads_a=ha, **kw) for _ in range(3)]
    ok_a = all(torch.equal(o[0], full[..., :ha * D]) for o in outs)
    ok_b = all(torch.equal(o[1], full[..., ha * D:]) for o in outs)
    print(f"H={H} G={G} U={U} Sl={Sl}: chunk A == columns [0,{ha * D}): {ok_a}; chunk B == the rest: {ok_b}; shapes {tuple(outs[0][0].shape)} {tuple(outs[0][1].shape)}")
