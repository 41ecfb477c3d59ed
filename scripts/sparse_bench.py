"""Stand-alone timing of the block-sparse attention kernel at BASELINE cfg2 geometry — not the contract bench.
  VSA: 624 blocks of 64 (grid 21x30x52 -> S_pad 39 936), random top-125 lists, 12 heads, the real variable block sizes
  STA: window (3,3,3) of (6,8,8) tiles on the ragged 21x30x52 grid as 128-row block lists (kernel_api.sliding_tile_block_lists)
Prints ms per launch and algorithmic TFLOP/s (FLOPs of the selected (query, key) pairs only)."""
import os as _os
_os.environ.setdefault("FVK_PROBE_LIB", "1")  # A/B switches exist only in the measurement build (scripts/probes/libfvk_probe.so)
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fastvideo_amd import kernel_api, ops

H, D = 12, 128
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
out = {}


def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(n):
            fn()
        e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) / n)
    return sorted(ts)[1]


# ---- VSA
meta = ops.vsa_build_metadata_host((21, 30, 52))
vbs = meta["variable_block_sizes"].to(dev)
n = vbs.numel()
S_pad = n * 64
q, k, v = (torch.randn((1, S_pad, H, D), generator=g, device=dev).bfloat16() for _ in range(3))
scores = torch.randn((1, H, n, n), generator=g, device=dev)
mask = ops.topk_mask(scores, 125)
idx, num = ops.map_to_index(mask)
IMPLS = [int(x) for x in os.environ.get("IMPLS", "0").split(",")]
ms_vt = timeit(lambda: ops.v_transpose(v))
pairs = float((vbs.float()[None, None, :, None] * (mask.float() * vbs.float()[None, None, None, :])).sum())
ref_o = None
for impl in IMPLS:
    ops.set_tunable("attn_impl", impl)
    o = ops.attn_block_sparse(q, k, v, idx, num, vbs, layout="bshd")
    ref_o = o if ref_o is None else ref_o
    ms = timeit(lambda: ops.attn_block_sparse(q, k, v, idx, num, vbs, layout="bshd"))
    out[f"vsa_impl{impl}"] = dict(ms=round(ms, 4), ms_kernel_only=round(ms - ms_vt, 4), tflops_real_pairs=round(4 * pairs * D / ((ms - ms_vt) * 1e-3) / 1e12, 1),
                                  tflops_padded_blocks=round(4.0 * S_pad * 125 * 64 * H * D / ((ms - ms_vt) * 1e-3) / 1e12, 1),
                                  equal_to_first=bool(torch.equal(o, ref_o)))
ops.set_tunable("attn_impl", 0)
# ---- STA lists
h = kernel_api.sliding_tile_block_lists((21, 30, 52), (6, 8, 8), (3, 3, 3))
S2 = h["S_pad"]
q2, k2, v2 = (torch.randn((1, S2, H, D), generator=g, device=dev).bfloat16() for _ in range(3))
idx2 = h["q2k_idx"].to(dev)[None, None].expand(1, H, -1, -1).contiguous()
num2 = h["q2k_num"].to(dev)[None, None].expand(1, H, -1).contiguous()
bs2 = h["block_sizes"].to(dev)
ms2_vt = timeit(lambda: ops.v_transpose(v2))
fl2 = 4.0 * (21 * 30 * 52)**2 * h["density"] * H * D
ref_o = None
for impl in IMPLS:
    ops.set_tunable("attn_impl", impl)
    o = ops.attn_block_sparse(q2, k2, v2, idx2, num2, bs2, layout="bshd", q_block=h["q_block"])
    ref_o = o if ref_o is None else ref_o
    ms2 = timeit(lambda: ops.attn_block_sparse(q2, k2, v2, idx2, num2, bs2, layout="bshd", q_block=h["q_block"]))
    out[f"sta_lists_impl{impl}"] = dict(ms=round(ms2, 4), ms_kernel_only=round(ms2 - ms2_vt, 4),
                                        tflops_algorithmic=round(fl2 / ((ms2 - ms2_vt) * 1e-3) / 1e12, 1), equal_to_first=bool(torch.equal(o, ref_o)))
ops.set_tunable("attn_impl", 0)
print(json.dumps(out))
