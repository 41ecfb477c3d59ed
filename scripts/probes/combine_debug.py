"""Where does the fused sparse + combine differ from the two-pass form?  (debug, round 6)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from fastvideo_amd import ops
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "oracle"))
DEV = "cuda"
import vsa_oracle as V
for (nq, nk, H, with_gate) in ((8, 8, 1, False), (8, 8, 1, True), (8, 8, 2, True), (520, 40, 2, True)):
    B = 1
    g = torch.Generator().manual_seed(nq)
    mk = lambda n: torch.randn((B, H, n * 64, 128), generator=g).to(torch.bfloat16).to(DEV)
    q, k, v = mk(nq), mk(nk), mk(nk)
    rng = np.random.default_rng(nq)
    bm = rng.random((B, H, nq, nk)) < 0.5
    bm[..., 0] = True
    vbs = np.full(nk, 64, dtype=np.int32)
    idx, num = V.map_to_index(bm)
    dv = lambda t: torch.from_numpy(t).to(DEV)
    out_c = torch.randn((B, H, nq, 128), generator=g).to(torch.bfloat16).to(DEV)
    gate = torch.randn((B, H, nq * 64, 128), generator=g).to(torch.bfloat16).to(DEV) if with_gate else None
    o_s = ops.attn_block_sparse(q, k, v, dv(idx), dv(num), dv(vbs), layout="bhsd")
    two = ops.vsa_combine(out_c, o_s, gate, 64, layout="bhsd")
    one = ops.vsa_sparse_combine(q, k, v, dv(idx), dv(num), dv(vbs), out_c, gate, layout="bhsd")
    d = (one.float() - two.float()).abs()
    bad = (d > 0)
    print(f"nq={nq} nk={nk} H={H} gate={with_gate}: equal={torch.equal(one, two)} n_bad={int(bad.sum())} max={d.max().item():.4g}")
    if bad.any():
        rows = bad.any(-1)[0]  # [H, S]
        for h in range(H):
            r = torch.nonzero(rows[h]).flatten().cpu().numpy()
            print(f"  head {h}: {len(r)} bad rows; first {r[:12]}; blocks {sorted(set((r // 64).tolist()))[:20]}")
        h_, r_ = [int(x) for x in torch.nonzero(rows)[0]]
        cols = torch.nonzero(bad[0, h_, r_]).flatten().cpu().numpy()
        print("  first bad row", h_, r_, "bad cols", cols[:32], "one", one[0, h_, r_, cols[:6]].float().cpu().numpy(), "two", two[0, h_, r_, cols[:6]].float().cpu().numpy(),
              "o_s", o_s[0, h_, r_, cols[:6]].float().cpu().numpy(), "oc", out_c[0, h_, r_ // 64, cols[:6]].float().cpu().numpy(),
              "g", None if gate is None else gate[0, h_, r_, cols[:6]].float().cpu().numpy())
