"""Workgroup order of the query-grouped sliding-tile attention (fvk_attn_tile_lists_bf16, grid 21x30x52, 12 heads): XCD-contiguous deal vs
hardware order ("attn_impl" 70), window classes in first-tile order vs longest KV list first."""
import os as _os
_os.environ.setdefault("FVK_PROBE_LIB", "1")  # A/B switches exist only in the measurement build (scripts/probes/libfvk_probe.so)
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fastvideo_amd import kernel_api, ops
grid = tuple(int(x) for x in sys.argv[1:4]) if len(sys.argv) >= 4 else (21, 30, 52)
H = 12
g = torch.Generator(device="cuda").manual_seed(0)
out = {}
for order in ("first_tile", "longest_first"):
    h = kernel_api.sliding_tile_block_lists(grid, (6, 8, 8), (3, 3, 3), group_order=order)
    q = torch.randn((1, h["group_rows"], H, 128), generator=g, device="cuda").bfloat16()
    k, v = (torch.randn((1, h["S_pad"], H, 128), generator=g, device="cuda").bfloat16() for _ in range(2))
    vt = ops.v_transpose(v)
    ex = lambda t, n: t.cuda()[None, None].expand(1, H, *([-1] * n)).contiguous()
    idx, num, bs = ex(h["group_q2k_idx"], 2), ex(h["group_q2k_num"], 1), h["block_sizes"].cuda()
    fn = lambda: ops.attn_tile_lists(q, k, None, idx, num, bs, 256, None, layout="bshd", vt=vt)
    for impl in (0, 70, 0, 70):
        ops.set_tunable("attn_impl", impl)
        fn(); torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10): fn()
        e.record(); torch.cuda.synchronize()
        out.setdefault(f"{order}/{'xcd_contiguous' if impl == 0 else 'hardware_order'}", []).append(round(s.elapsed_time(e) / 10, 4))
    ops.set_tunable("attn_impl", 0)
print(json.dumps(out))
