"""Round 5, co-residency bug, step 3: SYNTHETIC victims beside the REAL aggressors.

scripts/coresidency/coresidency_matrix.py showed: only the kernels that do the RoPE rotation fail, only beside the bf16 gemm_w1 family (inline-asm MFMAs on AGPR
accumulators; both MFMA shapes, both epilogues, 256 x 256 and 256 x 128 tiles) — not beside the fp8 form of the same kernel (builtin MFMAs), gemm_ph,
the attention / conv kernels or the vendor GEMM, although the victim's 56 registers fit beside all of them.  The first synthetic victim (packed
forms on register-resident values) stayed clean beside a synthetic MFMA stream.  Here the victims of scripts/probes/pk_victims.hip — the real
kernel's instruction sequence rebuilt one feature at a time (values from registers | loaded from global memory | packed ops IN PLACE on the
just-loaded pairs | the same with s_nop 7 after each s_waitcnt | scalar arithmetic | loads only) — run on stream B beside the real aggressors on
stream A.  Loaded values are re-derivable from their addresses, so wrong DATA is told apart from wrong ARITHMETIC.
usage: python scripts/coresidency/coresidency_victims.py [launches per cell = 60]"""
import os as _os
_os.environ["FVK_PROBE_LIB"] = "bug"
import ctypes as C, json, os, sys
HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))   # scripts/ (this file lives in scripts/coresidency/)
sys.path.insert(0, os.path.dirname(HERE))
import torch
from fastvideo_amd import _lib, ops
assert _lib.LIB_PATH.endswith("libfvk_bug.so"), _lib.LIB_PATH
pk = C.CDLL(os.path.join(HERE, "probes", "libpk_victims.so"))
pk.pkv_launch.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
pk.pkv_fill_launch.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
launches = int(sys.argv[1]) if len(sys.argv) > 1 else 60
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(5)
rn = lambda *s, sc=1.0: (torch.randn(s, generator=g) * sc).bfloat16().to(dev)
N4 = 1 << 16
tab = torch.empty(4 * N4, dtype=torch.float32, device=dev)
pk.pkv_fill_launch(tab.data_ptr(), 4 * N4, torch.cuda.current_stream().cuda_stream)
counters = torch.zeros(4, dtype=torch.int64, device=dev)
torch.cuda.synchronize()
A, B = rn(8192, 4096), rn(4096, 4096)
A8, As = ops.fp8_quantize(A, rowwise=False)
B8, Bs = ops.fp8_quantize(B, rowwise=False)
Sl, d, D = 338, 768, 128
qkv = rn(Sl, 3 * d)
wq, wk = (1 + 0.1 * torch.randn(d, generator=g)).bfloat16().to(dev), (1 + 0.1 * torch.randn(d, generator=g)).bfloat16().to(dev)
ang = torch.rand((2 * Sl, D), generator=g) * 6.28
cos, sin = torch.cos(ang).float().to(dev), torch.sin(ang).float().to(dev)
real = lambda: ops.qkv_norm_rope_pack(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], wq, wk, cos, sin, 2, 1, head_dim=D, seq_len=2 * Sl)


def gemm_var(impl):
    def f():
        ops.set_tunable("gemm_impl", impl)
        return ops.gemm(A, B)
    return f


aggressors = [("nothing", None), ("gemm_w1 bf16, shipped schedule, no register claim (408 registers)", gemm_var(0)),
              ("gemm_w1 bf16 VAR 3 (32x32x16 MFMAs, 452 registers)", gemm_var(5 + 8 * 3)), ("gemm_w1n bf16 (264 registers)", gemm_var(6)),
              ("gemm_w1 fp8 (builtin MFMAs, 444 registers)", lambda: ops.gemm_fp8(A8, As, B8, Bs)), ("gemm_ph (builtin MFMAs, arch-VGPR accumulators)", gemm_var(4)),
              ("vendor GEMM (torch.matmul)", lambda: A @ B.t())]
VIC = ["0 packed ops on register values", "1 loaded values, packed ops NOT in place", "2 loaded values, packed ops IN PLACE (the real sequence)",
       "3 as 2 + s_nop 7 after each s_waitcnt", "4 as 2 with SCALAR arithmetic", "5 loads only"]
BLOCKS, ITERS = 2048, 200
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
ref_real = real().clone()
torch.cuda.synchronize()
for aname, load in aggressors:
    # positive control: the real victim
    bad, outs = 0, []
    for i in range(launches):
        if load is not None:
            with torch.cuda.stream(sa):
                keep = load()
        with torch.cuda.stream(sb):
            outs.append(real())
    torch.cuda.synchronize()
    bad = sum(0 if torch.equal(o, ref_real) else 1 for o in outs)
    print(json.dumps({"aggressor": aname, "victim": "REAL QK-norm + RoPE + pack kernel", "wrong_launches": bad, "of": launches}), flush=True)
    for v in range(6):
        counters.zero_()
        torch.cuda.synchronize()
        for i in range(launches):
            if load is not None:
                with torch.cuda.stream(sa):
                    keep = load()
            rc = pk.pkv_launch(v, tab.data_ptr(), counters.data_ptr(), BLOCKS, ITERS, N4, sb.cuda_stream)
            assert rc == 0, rc
        torch.cuda.synchronize()
        c = counters.tolist()
        print(json.dumps({"aggressor": aname, "victim": VIC[v], "wrong_results": c[0], "wrong_low_half": c[1], "wrong_high_half": c[2], "wrong_loaded_values": c[3],
                          "of": launches * BLOCKS * 64 * ITERS}), flush=True)
ops.set_tunable("gemm_impl", 0)
