"""Do packed-fp32 VALU results go wrong when MFMA kernels run on the same CUs — in ONE process?  Stream A loops a large bf16 GEMM (torch /
vendor kernel, or this library's), stream B loops the QK-norm + RoPE + pack pass (whose RoPE arithmetic hipcc compiles to v_pk_mul_f32 /
v_pk_add_f32 with op_sel / neg modifiers) on FIXED inputs; every pack output is compared with the first.
usage: python scripts/coresidency/pk_f32_mfma_probe.py [iters=600]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from fastvideo_amd import ops
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 600
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(5)
Sl, d, D = 338, 768, 128
qkv = torch.randn((Sl, 3 * d), generator=g).bfloat16().to(dev)
wq, wk = (1 + 0.1 * torch.randn(d, generator=g)).bfloat16().to(dev), (1 + 0.1 * torch.randn(d, generator=g)).bfloat16().to(dev)
ang = torch.rand((2 * Sl, D), generator=g) * 6.28
cos, sin = torch.cos(ang).float().to(dev), torch.sin(ang).float().to(dev)
A = torch.randn((8192, 4096), generator=g).bfloat16().to(dev)
B = torch.randn((4096, 4096), generator=g).bfloat16().to(dev)
xe = torch.randn((64 << 20,), generator=g).to(dev)
pack = lambda: ops.qkv_norm_rope_pack(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], wq, wk, cos, sin, 2, 1, head_dim=D, seq_len=2 * Sl)
q4 = torch.randn((1, 8192, 12, 128), generator=g).bfloat16().to(dev)
xc = torch.randn((10, 240, 416, 96), generator=g).bfloat16().to(dev)
wc = (torch.randn((96, 27 * 96), generator=g) * (27 * 96)**-0.5).bfloat16().to(dev)
bc = torch.zeros(96).bfloat16().to(dev)
xv = torch.randn((1 << 20,), generator=g).to(dev)
yv = torch.randn((1 << 20,), generator=g).to(dev)
lnx = torch.randn((4096, 1536), generator=g).bfloat16().to(dev)
mul = torch.randn((1, 1536), generator=g).to(dev)
victims_all = (("fvk norm+rope+pack", lambda: ops.qkv_norm_rope_pack(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], wq, wk, cos, sin, 2, 1, head_dim=D, seq_len=2 * Sl)),
           ("torch fp32 x*y+x (1 M elements)", lambda: xv * yv + xv),
           ("torch layer_norm fp32 [4096,1536]", lambda: torch.nn.functional.layer_norm(lnx.float(), (1536,))),
           ("fvk ln_modulate [4096,1536]", lambda: ops.ln_modulate(lnx, mul=mul, add=mul, rows_per_batch=4096)))
q1 = torch.randn((1, 1024, 48, 128), generator=g).bfloat16().to(dev)
A2 = torch.randn((8192, 4160), generator=g).bfloat16().to(dev)
B2 = torch.randn((4096, 4160), generator=g).bfloat16().to(dev)
nb = 128
bmask = torch.zeros((1, 12, nb, nb), dtype=torch.bool)
bmask.scatter_(-1, torch.rand((1, 12, nb, nb), generator=g).topk(32, dim=-1).indices, True)
bidx, bnum = ops.map_to_index(bmask.to(dev))
bvbs = torch.full((nb,), 64, dtype=torch.int32, device=dev)
loads = (("nothing", None), ("fvk attention, 1024 keys (attn_pp2: 8 waves, 32x32x16)", lambda: ops.attn_dense(q1, q1, q1, layout="bshd")),
         ("fvk block-sparse attention (attn_fwd: 32x32x16, loader waves)", lambda: ops.attn_block_sparse(q4, q4, q4, bidx, bnum, bvbs, layout="bshd")),
         ("fvk gemm K = 4160 (gemm_ph: 2 waves per SIMD, 32x32x16)", lambda: ops.gemm(A2, B2)), ("torch bf16 matmul", lambda: A @ B), ("fvk gemm (gemm_w1)", lambda: ops.gemm(A, B)),
         ("fvk dense attention (8192 tokens, 12 heads)", lambda: ops.attn_dense(q4, q4, q4, layout="bshd")),
         ("fvk dense attention, attn_w64 (488 registers, 32x32x16)", lambda: ops.attn_dense(q4, q4, q4, layout="bshd", kernel=ops.ATTN_KERNEL_W64, key_splits=1)),
         ("fvk vae conv 96->96 (conv3w)", lambda: ops.vae_conv(xc, wc, bc, T=8, H=240, W=416, kt=3, ks=3)),
         ("torch elementwise fp32 (256 MB)", lambda: xe * 1.5 + 2.0))
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
victims = victims_all if os.environ.get("ALL_VICTIMS") else victims_all[:1]
for vname, vic in victims:
    ref = vic().clone()
    torch.cuda.synchronize()
    for name, load in loads:
        bad = 0
        outs = []
        for i in range(iters):
            if load is not None:
                with torch.cuda.stream(sa):
                    keep = load()
            with torch.cuda.stream(sb):
                outs.append(vic())
            if len(outs) == 25 or i == iters - 1:
                torch.cuda.synchronize()
                bad += sum(0 if torch.equal(o, ref) else 1 for o in outs)
                outs = []
        torch.cuda.synchronize()
        print(f"victim {vname:36s} | other stream: {name:46s}: {bad} of {iters} outputs differ from the first", flush=True)
