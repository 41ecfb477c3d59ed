"""What do the pipelined exchange's two head-chunk attention launches cost on ONE GPU at a rank's SP shapes (compute only, no exchange)?
Per P: the rank's attention as ONE launch (the plain exchange), as two chunk launches in stream order (FVK_SP_OVERLAP=1) and as two chunk
launches on two HIP streams (FVK_SP_OVERLAP=2).  Key runs (split-KV) decided from the whole head group in every form, as the model does.
usage: python scripts/sp_chunk_launch_ab.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fastvideo_amd import ops
dev = torch.device("cuda")
S, H, D = 32760, 12, 128
g = torch.Generator(device=dev).manual_seed(0)
side = torch.cuda.Stream()
for P, G, U in ((2, 2, 1), (4, 4, 1), (8, 4, 2)):
    hg = H // G
    Sq = -(-S // P) * G            # a rank's query rows: its query-block row of the G x U grid
    q = torch.randn((1, Sq, hg, D), generator=g, device=dev).bfloat16()
    k = torch.randn((1, S, hg, D), generator=g, device=dev).bfloat16()
    v = torch.randn((1, S, hg, D), generator=g, device=dev).bfloat16()
    vt = ops.v_transpose(v)
    splits = ops.attn_key_splits(-(-Sq // 256) * hg, -(-S // 128))
    ha = (hg + 1) // 2
    att = lambda a, b: ops.attn_dense(q[:, :, a:b], k[:, :, a:b], vt=vt[:, a:b], scale=D**-0.5, layout="bshd", key_splits=splits)

    def one():
        return att(0, hg)

    def two_in_order():
        return att(0, ha), att(ha, hg)

    def two_streams():
        main = torch.cuda.current_stream()
        side.wait_stream(main)
        oa = att(0, ha)
        with torch.cuda.stream(side):
            ob = att(ha, hg)
        main.wait_stream(side)
        return oa, ob

    res = {}
    for name, f in (("one launch", one), ("two chunks, stream order", two_in_order), ("two chunks, two streams", two_streams)):
        for _ in range(3): f()
        torch.cuda.synchronize()
        t = []
        for rep in range(3):
            s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s_.record()
            for _ in range(10): o = f()
            e_.record(); torch.cuda.synchronize()
            t.append(s_.elapsed_time(e_) / 10 * 1e3)
        res[name] = sorted(t)[1]
    full = one()
    oa, ob = two_streams()
    torch.cuda.synchronize()
    same = torch.equal(full[:, :, :ha], oa) and torch.equal(full[:, :, ha:], ob)
    print(f"P={P} (G{G}xU{U}): {hg} heads x {Sq} query rows, {splits} key run(s), chunks {ha}+{hg - ha} heads: " +
          ", ".join(f"{n} {v_:.0f} us" for n, v_ in res.items()) + f"; chunks == full launch bit for bit: {same}", flush=True)
