"""Segment timing probe of the 8-wave ping-pong attention kernel (fvk tunable attn_impl=14): s_memtime stamps of workgroup 0."""
import os as _os
_os.environ.setdefault("FVK_PROBE_LIB", "1")  # A/B switches exist only in the measurement build (scripts/probes/libfvk_probe.so)
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fastvideo_amd import ops, _lib
S, H, D = 32760, 12, 128
q, k, v = (torch.randn(1, S, H, D, device="cuda").bfloat16() for _ in range(3))
vt = ops.v_transpose(v); o = torch.empty_like(q)
buf = torch.zeros(2 * 8 * 8, dtype=torch.int64, device="cuda")
impl = int(sys.argv[1]) if len(sys.argv) > 1 else 14
ops.set_tunable("attn_impl", impl)
a = ops._attn_args(q, k, vt, o, D**-0.5, "bshd", lse=buf.view(torch.float32))
for _ in range(2):
    _lib.call("fvk_attn_dense_bf16", C.byref(a), ops._stream())
torch.cuda.synchronize()
t = buf.cpu()[:64].view(8, 8).double() / 256
names = ["loopback", "M(mfma)", "wait_dma", "barrier1", "dma_issue", "V(softmax)", "barrier2"]
if impl >= 100:
    names = ["loopback", "M(mfma)", "wait+barrier1", "dma_issue", "V(softmax)", "wait+barrier2"]
    t = buf.cpu()[:64].view(8, 8).double() / 128
    ab = buf.cpu()[64:].view(8, 8)[:, :6]
    ab = ab - ab.min()
    print("absolute stamps of tile 100 (M start, M end, after b1, after dma issue, softmax end, after b2):")
    for w in range(8): print("  wave", w, ab[w].tolist())
for w in range(8):
    print(f"wave {w}: " + " ".join(f"{n}={t[w, i]:.0f}" for i, n in enumerate(names)) + f" | tile {t[w, :len(names)].sum():.0f} cycles")
