"""Is a head-chunk attention launch on strided views of the exchange buffer deterministic, and equal to the un-chunked launch's columns?
(round 4: the pipelined exchange's 12-head / long-key case showed two consecutive forwards differing.)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fastvideo_amd import ops
torch.manual_seed(0)
D = 128
for (N, Sq, hg, S) in ((3256, 3256, 6, 3255), (3256, 1628, 6, 3255), (4096, 4096, 6, 4096), (3256, 3256, 3, 3255)):
    recv = torch.randn((N, 3, hg, D), device="cuda").bfloat16()
    q, k, v = recv[:Sq, 2], recv[:, 0], recv[:, 1]
    def run(qh, kh, vh, kernel=0):
        vt = ops.v_transpose(vh[:S].unsqueeze(0))
        return ops.attn_dense(qh.unsqueeze(0), kh[:S].unsqueeze(0), vt=vt, scale=D**-0.5, layout="bshd", kernel=kernel, key_splits=1)[0]
    full = [run(q, k, v) for _ in range(3)]
    print(f"N={N} Sq={Sq} hg={hg} S={S}: full launch repeatable: {all(torch.equal(full[0], f) for f in full[1:])}")
    for a, b in ((0, hg // 2), (hg // 2, hg)):
        # (1) chunk as a strided sub-view of the SAME buffer; (2) chunk in a buffer of its own, the layout the pipelined exchange receives
        sub = [run(q[:, a:b], k[:, a:b], v[:, a:b]) for _ in range(3)]
        own = recv[:, :, a:b].contiguous()
        ownr = [run(own[:Sq, 2], own[:, 0], own[:, 1]) for _ in range(3)]
        for name, outs in (("sub-view", sub), ("own buffer", ownr)):
            rep = all(torch.equal(outs[0], o) for o in outs[1:])
            eq = torch.equal(outs[0], full[0][:, a:b])
            d = (outs[0].float() - full[0][:, a:b].float()).abs()
            print(f"   heads [{a},{b}) {name}: repeatable {rep}, equals the full launch's columns {eq} (max diff {d.max().item():.3g}, {int((d > 0).sum())} elements)")
