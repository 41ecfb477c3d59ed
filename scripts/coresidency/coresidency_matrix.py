"""Round 5: round 4's co-residency bug with the REAL kernels, in the pre-fix state rebuilt on purpose.

scripts/probes/libfvk_bug.so (fastvideo_amd/_build.build_bug) = the measurement build WITHOUT the two fences: the one-wave-per-SIMD kernels do not
claim the whole register file, the small kernels are compiled WITH packed-fp32 instructions.  One process, two streams: stream A loops an
AGGRESSOR (an MFMA kernel of this library in one of its variants, or the vendor GEMM), stream B runs a VICTIM (an HBM-bound kernel of this library)
on FIXED inputs; every victim output is compared with the one computed alone.  A cell = `wrong` of `iters` victim launches differ.
Which aggressor property is needed?  gemm_w1's variants differ in ONE thing each:  gemm_impl 125 (VAR 15, shipped: 16x16x32, direct epilogue,
persistent) | 61 (VAR 7: the same MFMAs, LDS-bounce epilogue, one workgroup per tile) | 29 (VAR 3: 32x32x16 MFMAs) | fp8 (16x16x128 f8f6f4).
usage: FVK_PROBE_LIB=bug python scripts/coresidency/coresidency_matrix.py [iters=120]"""
import os as _os
if _os.environ.get("FVK_PROBE_LIB") not in ("bug", "bug2"):   # bug2: gemm_w1's MFMAs as compiler builtins (scripts/coresidency/build_bug2.sh)
    _os.environ["FVK_PROBE_LIB"] = "bug"
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from fastvideo_amd import _lib, ops
assert _lib.LIB_PATH.endswith(("libfvk_bug.so", "libfvk_bug2.so")), _lib.LIB_PATH
print(json.dumps({"library": os.path.basename(_lib.LIB_PATH)}), flush=True)
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 120
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(5)
rn = lambda *s, sc=1.0, dt=torch.bfloat16: (torch.randn(s, generator=g) * sc).to(dt).to(dev)
Sl, d, D = 338, 768, 128
qkv = rn(Sl, 3 * d)
wq, wk = (1 + 0.1 * torch.randn(d, generator=g)).bfloat16().to(dev), (1 + 0.1 * torch.randn(d, generator=g)).bfloat16().to(dev)
ang = torch.rand((2 * Sl, D), generator=g) * 6.28
cos, sin = torch.cos(ang).float().to(dev), torch.sin(ang).float().to(dev)
lnx, lnr = rn(4096, 1536), rn(4096, 1536)
mul, gate = rn(1, 1536, dt=torch.float32), rn(1, 1536, dt=torch.float32)
lnw, lnb = rn(1536, dt=torch.float32), rn(1536, dt=torch.float32)
v4 = rn(1, 4096, 12, 128)
idx = torch.randperm(4096, generator=g)[:3000].to(torch.int32).to(dev)
bm = rn(1, 4, 64 * 32, 128)
vbs = torch.full((32,), 64, dtype=torch.int32, device=dev)
lat = rn(1, 16, 5, 32, 32)
sc = rn(4, 624, 624, sc=2.0)
victims = [
    ("QK-norm + RoPE + pack (round 4's victim)", lambda: ops.qkv_norm_rope_pack(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], wq, wk, cos, sin, 2, 1, head_dim=D, seq_len=2 * Sl)),
    ("QK-norm + RoPE (no pack)", lambda: torch.cat(ops.rmsnorm_rope([qkv[:, :d], qkv[:, d:2 * d]], [wq, wk], cos[:Sl], sin[:Sl], head_dim=D, seq_len=Sl), 1)),
    ("RMS norm only (no RoPE tables)", lambda: ops.rmsnorm_rope([qkv[:, :d]], [wq], head_dim=D, seq_len=Sl)[0]),
    ("LayerNorm + modulate", lambda: ops.ln_modulate(lnx, mul=mul, add=mul, rows_per_batch=4096)),
    ("gated residual + LayerNorm(affine) + both outputs", lambda: torch.cat(ops.ln_modulate(lnx, residual=lnr, gate=gate, ln_w=lnw, ln_b=lnb, want_residual=True, rows_per_batch=4096), 1)),
    ("LayerNorm + modulate -> fp8 rows", lambda: ops.ln_modulate(lnx, mul=mul, add=mul, rows_per_batch=4096, fp8_rowwise="only")[0].view(torch.uint8)),
    ("gated residual (scale_residual)", lambda: ops.scale_residual(lnr, lnx, gate, rows_per_batch=4096)),
    ("V transpose", lambda: ops.v_transpose(v4)),
    ("fp8 quantise, per row", lambda: ops.fp8_quantize(lnx, rowwise=True)[0].view(torch.uint8)),
    ("fp8 quantise, per tensor", lambda: ops.fp8_quantize(lnx, rowwise=False)[0].view(torch.uint8)),
    ("gather rows", lambda: ops.gather_rows(v4, 3000, src_index=idx)),
    ("block mean", lambda: ops.block_mean(bm, vbs, 64)),
    ("softmax rows", lambda: ops.softmax_rows(sc)),
    ("patchify", lambda: ops.patchify(lat, (1, 2, 2))),
    ("timestep embedding + silu", lambda: ops.silu(ops.timestep_embedding(torch.tensor([500.0, 7.0], device=dev), 256).bfloat16())),
]
A, B = rn(8192, 4096), rn(4096, 4096)
A8, As = ops.fp8_quantize(A, rowwise=False)
B8, Bs = ops.fp8_quantize(B, rowwise=False)
q4 = rn(1, 8192, 12, 128)
xc = rn(10, 240, 416, 96)
wc = (torch.randn((96, 27 * 96), generator=g) * (27 * 96)**-0.5).bfloat16().to(dev)
bc = torch.zeros(96).bfloat16().to(dev)


def gemm_var(impl):
    def f():
        ops.set_tunable("gemm_impl", impl)
        return ops.gemm(A, B)
    return f


aggressors = [
    ("nothing", None),
    ("gemm_w1 shipped schedule (16x16x32, AGPR accumulators, direct epilogue, persistent) WITHOUT the register claim", gemm_var(0)),
    ("gemm_w1 VAR 7 (same MFMAs; LDS-bounce epilogue, one workgroup per tile)", gemm_var(5 + 8 * 7)),
    ("gemm_w1 VAR 3 (32x32x16 MFMAs)", gemm_var(5 + 8 * 3)),
    ("gemm_w1n (256 x 128 tiles) without the claim", gemm_var(6)),
    ("gemm_w1 fp8 (16x16x128 f8f6f4 MFMAs) without the claim", lambda: ops.gemm_fp8(A8, As, B8, Bs)),
    ("gemm_ph (2 waves per SIMD, 32x32x16, compiler-placed accumulators)", gemm_var(4)),
    ("attn_w16 without the claim (474 registers)", lambda: ops.attn_dense(q4, q4, q4, layout="bshd")),
    ("conv3w 96->96 without the claim (459-490 registers)", lambda: ops.vae_conv(xc, wc, bc, T=8, H=240, W=416, kt=3, ks=3)),
    ("vendor GEMM (torch.matmul)", lambda: A @ B.t()),
]
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
table = {}
for vname, vic in victims:
    ops.set_tunable("gemm_impl", 0)
    ref = vic().clone()
    torch.cuda.synchronize()
    for aname, load in aggressors:
        bad, outs = 0, []
        for i in range(iters):
            if load is not None:
                with torch.cuda.stream(sa):
                    keep = load()
            with torch.cuda.stream(sb):
                outs.append(vic())
            if len(outs) == 20 or i == iters - 1:
                torch.cuda.synchronize()
                bad += sum(0 if torch.equal(o, ref) else 1 for o in outs)
                outs = []
        torch.cuda.synchronize()
        table.setdefault(vname, {})[aname] = bad
        print(json.dumps({"victim": vname, "aggressor": aname, "wrong": bad, "of": iters}), flush=True)
ops.set_tunable("gemm_impl", 0)
print(json.dumps({"summary_wrong_launches_of_%d" % iters: {v: {a: n for a, n in row.items() if n} for v, row in table.items()}}), flush=True)
