"""Block-sparse (VSA) kernel at cfg2 geometry (624 blocks of 64, random top-20 % lists, 12 heads, real block sizes), interleaved:
  0  = shipped: two lists per workgroup, 4 compute + 4 loader waves, one-stage-ahead LDS-DMA (attn_fwd.hip)
  53 = the same with register-staged loader waves, two tiles ahead (bit-identical)
  54 = the key-split kernel (attn_vsa.hip): 8 compute waves, register-staged prefetch two tiles ahead (merges two key halves per row:
       equal to rounding)
LAYOUT=bhsd makes every 64-key K block one contiguous 16 KiB (bshd: 64 pieces of 256 B, 3 KiB apart)."""
import os as _os
_os.environ.setdefault("FVK_PROBE_LIB", "1")  # A/B switches exist only in the measurement build (scripts/probes/libfvk_probe.so)
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fastvideo_amd import ops
H, D, dev = 12, 128, "cuda"
g = torch.Generator(device=dev).manual_seed(0)
grid = tuple(int(x) for x in sys.argv[1:4]) if len(sys.argv) >= 4 else (21, 30, 52)
meta = ops.vsa_build_metadata_host(grid)
vbs = meta["variable_block_sizes"].to(dev)
n = vbs.numel()
topk = max(1, -(-n // 5))
S_pad = n * 64
LAYOUT = os.environ.get("LAYOUT", "bshd")  # bhsd: every 64-key K block is one contiguous 16 KiB
shape = (1, S_pad, H, D) if LAYOUT == "bshd" else (1, H, S_pad, D)
q, k, v = (torch.randn(shape, generator=g, device=dev).bfloat16() for _ in range(3))
mask = ops.topk_mask(torch.randn((1, H, n, n), generator=g, device=dev), topk)
idx, num = ops.map_to_index(mask)
fn = lambda: ops.attn_block_sparse(q, k, v, idx, num, vbs, layout=LAYOUT)
IMPLS = (0, 53, 54)
res, outs = {i: [] for i in IMPLS}, {}
for r in range(4):
    for impl in IMPLS:
        ops.set_tunable("attn_impl", impl)
        o = fn(); torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(5): fn()
        e.record(); torch.cuda.synchronize()
        res[impl].append(round(s.elapsed_time(e) / 5, 4))
        if r == 0: outs[impl] = o
ops.set_tunable("attn_impl", 0)
err = (outs[54].float() - outs[0].float()).abs()
kv_bytes = float(num.sum()) * 2 * 64 * 128 * 2  # K + V^T bytes staged per launch
best = min(sorted(res[0])[1], 1e9)
print(json.dumps({"layout": LAYOUT, "grid": grid, "blocks": n, "topk": topk, "lds_dma_ms": res[0], "register_staged_loaders_ms": res[53],
                  "key_split_ms": res[54], "loader_forms_bit_identical": bool(torch.equal(outs[0], outs[53])),
                  "key_split_vs_shipped_max_abs": round(err.max().item(), 6), "mean_abs": float(f"{err.mean().item():.3g}"),
                  "staged_GB_per_launch": round(kv_bytes / 1e9, 2), "shipped_ingest_TB_per_s": round(kv_bytes / best / 1e9, 2),
                  "shipped_ingest_B_per_clk_per_CU_at_2GHz": round(kv_bytes / (best * 1e-3) / 256 / 2e9, 1)}))
