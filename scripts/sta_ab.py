"""A/B of the 128-row block-sparse (sliding-tile) attention at the 81f x 480p grid: attn_impl 0 (4 compute waves issuing their own DMA, 2 workgroups per CU)
vs 51 (4 compute + 4 loader waves, 1 workgroup per CU).  Measured on MI355X: 2.87 vs 3.04 ms per launch, bit-identical."""
import os as _os
_os.environ.setdefault("FVK_PROBE_LIB", "1")  # A/B switches exist only in the measurement build (scripts/probes/libfvk_probe.so)
import sys, json, torch
sys.path.insert(0, "/root/repo")
from fastvideo_amd import ops, kernel_api
# STA block lists at the cfg3 grid, timed alone: impl 0 vs 51
h = kernel_api.sliding_tile_block_lists((21, 30, 52), (6, 8, 8), (3, 3, 3))
H = 12; S_pad = h["S_pad"]
q, k, v = (torch.randn(1, S_pad, H, 128, device="cuda").bfloat16() for _ in range(3))
idx = h["q2k_idx"].cuda()[None, None].expand(1, H, -1, -1).contiguous(); num = h["q2k_num"].cuda()[None, None].expand(1, H, -1).contiguous(); bs = h["block_sizes"].cuda()
outs = {}
for r in range(3):
    for impl in (0, 51):
        ops.set_tunable("attn_impl", impl)
        o = ops.attn_block_sparse(q, k, v, idx, num, bs, layout="bshd", q_block=h["q_block"]); torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(5): ops.attn_block_sparse(q, k, v, idx, num, bs, layout="bshd", q_block=h["q_block"])
        e.record(); torch.cuda.synchronize()
        outs.setdefault(impl, []).append(s.elapsed_time(e) / 5)
        if r == 0: outs[("o", impl)] = o.clone()
ops.set_tunable("attn_impl", 0)
print("equal", torch.equal(outs[("o", 0)], outs[("o", 51)]), {i: sorted(outs[i])[1] for i in (0, 51)})
