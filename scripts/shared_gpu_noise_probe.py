"""Is the wrong-value noise of the multi-process one-GPU harness a property of OUR kernels or of several processes time-slicing a GPU?
N processes on cuda:0 run, concurrently and without any communication, loops of (a) plain PyTorch ops (layer_norm -> matmul -> softmax ->
matmul: vendor / ATen kernels only), (b) this library's QK-norm + RoPE + pack pass, (c) its GEMM, (d) its dense attention — each on fixed inputs,
every output compared with the loop's first.  Prints mismatching iterations per op per process.
usage: python scripts/shared_gpu_noise_probe.py [procs=3] [iters=400]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.multiprocessing as mp


def worker(rank, procs, iters, bar, q):
    if os.environ.get("GEMM_IMPL"):
        os.environ["FVK_PROBE_LIB"] = "1"
    from fastvideo_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(100 + rank)
    Sl, d, H, D = 338, 768, 6, 128
    x = torch.randn((Sl, d), generator=g).bfloat16().to(dev)
    w1 = (torch.randn((3 * d, d), generator=g) * d**-0.5).bfloat16().to(dev)
    b1 = torch.zeros(3 * d).bfloat16().to(dev)
    wq, wk = (1 + 0.1 * torch.randn(d, generator=g)).bfloat16().to(dev), (1 + 0.1 * torch.randn(d, generator=g)).bfloat16().to(dev)
    ang = torch.rand((2 * Sl, D), generator=g) * 6.28
    cos, sin = torch.cos(ang).float().to(dev), torch.sin(ang).float().to(dev)
    qa = torch.randn((1, 2 * Sl, H, D), generator=g).bfloat16().to(dev)
    ka = torch.randn((1, 2 * Sl, H, D), generator=g).bfloat16().to(dev)
    va = torch.randn((1, 2 * Sl, H, D), generator=g).bfloat16().to(dev)
    wt = (torch.randn((d, d), generator=g) * d**-0.5).to(dev)
    xt = torch.randn((Sl, d), generator=g).to(dev)

    def op_torch():
        h = torch.nn.functional.layer_norm(xt, (d,))
        a = torch.softmax(h @ wt, dim=-1)
        return (a @ wt.t()).bfloat16()

    def op_gemm():
        return ops.gemm(x, w1, b1)

    qkv = ops.gemm(x, w1, b1)

    def op_pack():
        return ops.qkv_norm_rope_pack(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], wq, wk, cos, sin, 2, 1, head_dim=D, seq_len=2 * Sl, pos_offset=rank % 2 * Sl)

    def op_attn():
        return ops.attn_dense(qa, ka, va, layout="bshd")

    def op_chain():   # the sequence of the model: GEMM -> pack -> attention on the packed views, no synchronisation in between
        z = ops.gemm(x, w1, b1)
        s_ = ops.qkv_norm_rope_pack(z[:, :d], z[:, d:2 * d], z[:, 2 * d:], wq, wk, cos, sin, 2, 1, head_dim=D, seq_len=2 * Sl, pos_offset=0)
        r4 = s_.view(2 * Sl, 3, d // 2 // D, D)
        return ops.attn_dense(r4[None, :, 2], r4[None, :, 0], r4[None, :, 1], layout="bshd")

    def pack_of(z):
        return ops.qkv_norm_rope_pack(z[:, :d], z[:, d:2 * d], z[:, 2 * d:], wq, wk, cos, sin, 2, 1, head_dim=D, seq_len=2 * Sl, pos_offset=0)

    def attn_of(s_):
        r4 = s_.view(2 * Sl, 3, d // 2 // D, D)
        return ops.attn_dense(r4[None, :, 2], r4[None, :, 0], r4[None, :, 1], layout="bshd")

    s_fixed = pack_of(qkv)
    r4f = s_fixed.view(2 * Sl, 3, d // 2 // D, D)
    vt_fixed = ops.v_transpose(r4f[None, :, 1])
    zbuf = torch.empty((Sl, 3 * d), dtype=torch.bfloat16, device=dev)

    def gemm_poisoned(M=None):   # the GEMM into a NaN-filled buffer, copied at once: NaN left = never written; wrong finite values = computed wrong
        zbuf.fill_(float("nan"))
        ops.gemm(x, w1, b1, out=zbuf)
        return zbuf.clone()

    xl = torch.randn((4095, 1536), generator=g).bfloat16().to(dev)   # a P = 8 rank's out-projection: gemm_w1n's own shape (192 tiles of 256 x 128)
    wl = (torch.randn((1536, 1536), generator=g) * 1536**-0.5).bfloat16().to(dev)
    zl = torch.empty((4095, 1536), dtype=torch.bfloat16, device=dev)

    def gemm_poisoned_large():
        zl.fill_(float("nan"))
        ops.gemm(xl, wl, None, out=zl)
        return zl.clone()

    def gemm_pack_z():   # the pack output and, behind it, a copy of the z it read, side by side: [s_ flattened | z flattened]
        z = ops.gemm(x, w1, b1)
        s_ = pack_of(z)
        return torch.cat([s_.reshape(Sl, -1), z.clone()], dim=1)   # [Sl, 2*3*384 + 3*768]

    impl = os.environ.get("GEMM_IMPL")
    if impl:
        ops.set_tunable("gemm_impl", int(impl))   # measurement build: 14 forbids gemm_w1n, 6 forces it
    subs = (("gemm into NaN buffer -> copy", gemm_poisoned), ("[4095,1536,1536] gemm into NaN buffer -> copy", gemm_poisoned_large),
            ("gemm -> pack", lambda: pack_of(ops.gemm(x, w1, b1))), ("gemm -> pack, then z copied", gemm_pack_z),
            ("gemm -> copy -> pack of the copy", lambda: pack_of(ops.gemm(x, w1, b1).clone())),
            ("gemm -> pack without RoPE tables", lambda: (lambda z: ops.qkv_norm_rope_pack(z[:, :d], z[:, d:2 * d], z[:, 2 * d:], wq, wk, None, None, 2, 1, head_dim=D, seq_len=2 * Sl))(ops.gemm(x, w1, b1))),
            ("torch matmul -> pack", lambda: pack_of((x @ w1.t()).contiguous())), ("pack -> attention", lambda: attn_of(pack_of(qkv))),
            ("attention on fixed packed views", lambda: attn_of(s_fixed)), ("v_transpose of a fixed packed view", lambda: ops.v_transpose(r4f[None, :, 1])),
            ("attention kernel alone (fixed q, k, V^T)", lambda: ops.attn_dense(r4f[None, :, 2], r4f[None, :, 0], vt=vt_fixed, scale=D**-0.5, layout="bshd")),
            ("torch chain (3 dependent ops)", lambda: (torch.softmax(torch.nn.functional.layer_norm(xt @ wt, (d,)), -1) @ wt.t()).bfloat16()))
    only = os.environ.get("ONLY")
    res = {}
    for name, f in (("torch ops", op_torch), ("fvk gemm", op_gemm), ("fvk norm+rope+pack", op_pack), ("fvk attention", op_attn), ("fvk chain", op_chain)) + subs:
        if only and name not in only.split(","):
            continue
        ref = f().clone()
        torch.cuda.synchronize()
        bar.wait()
        bad = []
        outs = []
        for i in range(iters):
            outs.append(f())
            if len(outs) == 20 or i == iters - 1:   # compare in batches: the loop itself stays free of synchronisations
                for j, o in enumerate(outs):
                    if not torch.equal(o, ref):
                        dm = (o != ref) & ~(torch.isnan(o.float()) & torch.isnan(ref.float()))
                        idx = dm.nonzero()
                        where = f"rows {sorted(set(idx[:, 0].tolist()))[:6]} cols {sorted(set(idx[:, 1].tolist()))[:10]}" if o.dim() == 2 and len(idx) else ""
                        bad.append((i - len(outs) + 1 + j, int(dm.sum()), f"nan {int(torch.isnan(o.float()).sum())}", where))
                outs = []
        torch.cuda.synchronize()
        res[name] = bad
        bar.wait()
    q.put((rank, res))


if __name__ == "__main__":
    procs = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 400
    ctx = mp.get_context("spawn")
    bar, q = ctx.Barrier(procs), ctx.Queue()
    ps = [ctx.Process(target=worker, args=(r, procs, iters, bar, q)) for r in range(procs)]
    [p.start() for p in ps]
    got = sorted(q.get(timeout=900) for _ in ps)
    [p.join() for p in ps]
    print(f"{procs} process(es) x {iters} iterations per op, every output compared with the loop's first:")
    for rank, res in got:
        for name, bad in res.items():
            print(f"  process {rank}  {name:22s}: {len(bad)} mismatching iterations" + (f"  (iteration, differing elements, NaNs, where) {bad[:4]}" if bad else ""))
