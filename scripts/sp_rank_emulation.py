"""Per-rank COMPUTE of one DiT layer at sequence-parallel shapes, timed on ONE GPU (no exchange): what a rank of an SP = P run executes
between the collectives.  cfg2 (S = 32 760, d = 1536, 12 heads, ffn 8960); P in {1, 2, 4, 8}: Sl = ceil(S/P) local tokens for every
token-local op, attention on G*Sl query rows x (12/G) heads x all S keys (G = gcd(12, P), U = P/G).  Not the contract bench: it shows how
tile quantisation of the per-rank shapes (workgroups vs 256 CUs) bounds the strong scaling before any xGMI cost.
usage: python scripts/sp_rank_emulation.py"""
import json
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fastvideo_amd import ops

S, d, H, D, F, Lc = 32760, 1536, 12, 128, 8960, 512
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
rn = lambda *s: torch.randn(s, generator=g, device=dev).bfloat16()
W = dict(qkv=rn(3 * d, d) * d**-0.5, o=rn(d, d) * d**-0.5, f1=rn(F, d) * d**-0.5, f2=rn(d, F) * F**-0.5)
b = dict(qkv=rn(3 * d), o=rn(d), f1=rn(F), f2=rn(d))
nw = rn(d)
cos = torch.randn((S, D), generator=g, device=dev)
sin = torch.randn((S, D), generator=g, device=dev)


def t(fn, n=8):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3  # us


out = {}
for P in (1, 2, 4, 8):
    G = math.gcd(H, P); U = P // G
    Sl = (S + P - 1) // P
    hg = H // G
    x = rn(Sl, d)
    mod = torch.randn((1, d), generator=g, device=dev)
    qkv = rn(Sl, 3 * d)
    ff = rn(Sl, F)
    qb, kall, vall = rn(1, G * Sl, hg, D), rn(1, P * Sl, hg, D), rn(1, P * Sl, hg, D)
    cq, ck, cv = rn(1, Sl, H, D), rn(1, Lc, H, D), rn(1, Lc, H, D)
    r = {}
    r["ln_modulate x3"] = 3 * t(lambda: ops.ln_modulate(x, mul=mod, add=mod))
    r["gemm qkv"] = t(lambda: ops.gemm(x, W["qkv"], b["qkv"]))
    r["norm_rope(+pack)"] = t(lambda: ops.rmsnorm_rope([qkv[:, :d], qkv[:, d:2 * d]], [nw, nw], cos, sin, head_dim=D, seq_len=S))
    r["v_transpose"] = t(lambda: ops.v_transpose(vall))
    vt = ops.v_transpose(vall)
    r["self-attention"] = t(lambda: ops.attn_dense(qb, kall[:, :S], vt=vt, layout="bshd"))   # automatic key split for small grids
    unsplit = t(lambda: ops.attn_dense(qb, kall[:, :S], vt=vt, layout="bshd", key_splits=1))
    r["gemm o / cq / co (x3)"] = 3 * t(lambda: ops.gemm(x, W["o"], b["o"]))
    r["cross-attention"] = t(lambda: ops.attn_dense(cq, ck, cv, layout="bshd"))
    r["gemm ffn-in + gelu"] = t(lambda: ops.gemm(x, W["f1"], b["f1"], epilogue=ops.EPI_GELU_TANH))
    r["gemm ffn-out + gated residual"] = t(lambda: ops.gemm(ff, W["f2"], b["f2"], epilogue=ops.EPI_RESIDUAL_GATE, residual=x, gate=mod))
    tot = sum(r.values())
    # round 6: the same ops IN THE ORDER A LAYER RUNS THEM (every op once per layer, 30 layers back to back), issued eagerly and replayed as ONE
    # HIP graph — what a rank's compute costs when launch gaps are the eager host's and when they are the graph's
    def layer():
        ops.ln_modulate(x, mul=mod, add=mod)
        ops.gemm(x, W["qkv"], b["qkv"])
        ops.rmsnorm_rope([qkv[:, :d], qkv[:, d:2 * d]], [nw, nw], cos, sin, head_dim=D, seq_len=S)
        ops.v_transpose(vall)
        ops.attn_dense(qb, kall[:, :S], vt=vt, layout="bshd")
        ops.gemm(x, W["o"], b["o"])
        ops.ln_modulate(x, mul=mod, add=mod)
        ops.gemm(x, W["o"], b["o"])
        ops.attn_dense(cq, ck, cv, layout="bshd")
        ops.gemm(x, W["o"], b["o"])
        ops.ln_modulate(x, mul=mod, add=mod)
        ops.gemm(x, W["f1"], b["f1"], epilogue=ops.EPI_GELU_TANH)
        ops.gemm(ff, W["f2"], b["f2"], epilogue=ops.EPI_RESIDUAL_GATE, residual=x, gate=mod)
    def thirty():
        for _ in range(30):
            layer()
    seq_eager = t(thirty, n=3) / 30
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        thirty()
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            thirty()
    torch.cuda.current_stream().wait_stream(side)
    seq_graph = t(gr.replay, n=3) / 30
    del gr
    nqb = ((G * Sl + 255) // 256) * hg
    out[f"P{P}"] = dict(layout=f"G{G}xU{U}", Sl=Sl, attn_workgroups=nqb, attn_key_splits=ops.attn_key_splits(nqb, (S + 127) // 128),
                        self_attention_unsplit_us=round(unsplit, 1), layer_us=round(tot, 1), forward_ms_30_layers=round(tot * 30 / 1e3, 2),
                        layer_us_in_sequence_eager=round(seq_eager, 1), layer_us_in_sequence_hip_graph=round(seq_graph, 1),
                        per_op_us={k_: round(v_, 1) for k_, v_ in r.items()})
base = out["P1"]["layer_us"]
for P in (2, 4, 8):
    out[f"P{P}"]["compute_only_speedup"] = round(base / out[f"P{P}"]["layer_us"], 2)
    out[f"P{P}"]["compute_only_speedup_in_sequence_eager"] = round(out["P1"]["layer_us_in_sequence_eager"] / out[f"P{P}"]["layer_us_in_sequence_eager"], 2)
    out[f"P{P}"]["compute_only_speedup_in_sequence_hip_graph"] = round(out["P1"]["layer_us_in_sequence_hip_graph"] / out[f"P{P}"]["layer_us_in_sequence_hip_graph"], 2)
print(json.dumps(out))
