"""Round 5, co-residency bug, step 4: WHICH PART of gemm_w1 is the aggressor?  The synthetic victim of step 3 (packed RoPE forms on
register values: no loads, 18 registers) and the real QK-norm / RoPE kernel run beside gemm_w1 from a bug library whose gemm_w1 main loop has parts
removed (FVK_PROBE_LIB=bug_s<N>, scripts/coresidency/build_bug_strips.sh: 1 no LDS-DMA, 2 no fragment reads, 4 no barriers, 8 no MFMAs; sums combine), and beside
its no-epilogue timing variant (gemm_impl 253 = VAR 31).  usage: FVK_PROBE_LIB=bug_s8 python scripts/coresidency/coresidency_strips.py [launches = 40]"""
import os as _os
assert _os.environ.get("FVK_PROBE_LIB", "").startswith("bug"), "set FVK_PROBE_LIB=bug | bug_s<N>"
import ctypes as C, json, os, sys
HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))   # scripts/ (this file lives in scripts/coresidency/)
sys.path.insert(0, os.path.dirname(HERE))
import torch
from fastvideo_amd import _lib, ops
pk = C.CDLL(os.path.join(HERE, "probes", "libpk_victims.so"))
pk.pkv_launch.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
pk.pkv_fill_launch.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
launches = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(5)
rn = lambda *s: torch.randn(s, generator=g).bfloat16().to(dev)
N4 = 1 << 16
tab = torch.empty(4 * N4, dtype=torch.float32, device=dev)
pk.pkv_fill_launch(tab.data_ptr(), 4 * N4, torch.cuda.current_stream().cuda_stream)
counters = torch.zeros(4, dtype=torch.int64, device=dev)
A, B = rn(8192, 4096), rn(4096, 4096)
Sl, d, D = 338, 768, 128
qkv = rn(Sl, 3 * d)
wq, wk = (1 + 0.1 * torch.randn(d, generator=g)).bfloat16().to(dev), (1 + 0.1 * torch.randn(d, generator=g)).bfloat16().to(dev)
ang = torch.rand((2 * Sl, D), generator=g) * 6.28
cos, sin = torch.cos(ang).float().to(dev), torch.sin(ang).float().to(dev)
real = lambda: ops.qkv_norm_rope_pack(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], wq, wk, cos, sin, 2, 1, head_dim=D, seq_len=2 * Sl)
ref_real = real().clone()
torch.cuda.synchronize()
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
lib = os.path.basename(_lib.LIB_PATH)
for aname, impl in (("gemm_w1 (this library's build)", 0), ("the same without its epilogue (VAR 31)", 5 + 8 * 31)):
    ops.set_tunable("gemm_impl", impl)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ops.gemm(A, B); torch.cuda.synchronize()
    s.record()
    for _ in range(10): ops.gemm(A, B)
    e.record(); torch.cuda.synchronize()
    gemm_us = s.elapsed_time(e) * 100
    outs = []
    for i in range(launches):
        with torch.cuda.stream(sa):
            keep = ops.gemm(A, B)
        with torch.cuda.stream(sb):
            outs.append(real())
    torch.cuda.synchronize()
    bad_real = sum(0 if torch.equal(o, ref_real) else 1 for o in outs)
    res = {"library": lib, "aggressor": aname, "aggressor_us_alone": round(gemm_us, 1), "real_victim_wrong_launches": bad_real, "of_launches": launches}
    for v, vname in ((0, "synthetic_packed_on_registers"), (2, "synthetic_loaded_in_place")):
        counters.zero_(); torch.cuda.synchronize()
        for i in range(launches):
            with torch.cuda.stream(sa):
                keep = ops.gemm(A, B)
            pk.pkv_launch(v, tab.data_ptr(), counters.data_ptr(), 2048, 200, N4, sb.cuda_stream)
        torch.cuda.synchronize()
        c = counters.tolist()
        res[vname] = {"wrong": c[0], "low": c[1], "high": c[2], "loads": c[3]}
    print(json.dumps(res), flush=True)
ops.set_tunable("gemm_impl", 0)
