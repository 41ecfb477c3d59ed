"""Which tensor of an SP forward first differs between two consecutive forwards (and from SP = 1)?  2 or 3 ranks share cuda:0 over gloo.
usage: python scripts/sp_forward_determinism.py <world> <heads> <overlap 0|1>"""
import os, sys, socket
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist, torch.multiprocessing as mp


def forward_traces(heads, lat_shape, n=3):
    from fastvideo_amd import wan_config as WC
    from fastvideo_amd.wan_dit import WanTransformer3DModelHip
    cfg = WC.WanConfig("sp-det", heads, 128, 768, 2, text_dim=64)
    sd = WC.random_state_dict(cfg, seed=4, device="cpu")
    if os.environ.get("GEMM_IMPL"):
        from fastvideo_amd import ops
        ops.set_tunable("gemm_impl", int(os.environ["GEMM_IMPL"]))   # measurement build: 14 = gemm_w1n forbidden
    model = WanTransformer3DModelHip(sd, cfg.num_heads, device="cuda:0")
    g = torch.Generator().manual_seed(12)
    lat = torch.randn(lat_shape, generator=g).bfloat16()
    ctx = torch.randn((1, 40, cfg.text_dim), generator=g).bfloat16()
    outs = []
    rec = None
    if os.environ.get("HOOK"):   # record the tensors of every packed exchange (send buffer, received buffer, attention output, returned shard)
        sp_ = model.sp
        orig = sp_.attention_packed

        def hooked(send, S, attn_fn, head_dim=128):
            Sl = send.shape[1]
            recv = sp_.exchange_rows(send)
            q_blk, k_all, v_all = sp_.views_of(recv, head_dim)
            o_blk = attn_fn(q_blk, k_all, v_all, S)
            out = sp_.scatter_seq_gather_heads(o_blk, Sl)
            i = len(rec) // 4
            rec.update({f"L{i}.send": send.cpu(), f"L{i}.recv": recv.cpu(), f"L{i}.o_blk": o_blk.cpu(), f"L{i}.out": out.cpu()})
            return out
        sp_.attention_packed = hooked
    calls = None
    if os.environ.get("OPS"):   # record the output of every kernel front-end call, in call order
        from fastvideo_amd import ops as _ops
        def wrap(name):
            f = getattr(_ops, name)
            def g(*a, **k):
                r = f(*a, **k)
                flat = []
                def add(x):
                    if isinstance(x, torch.Tensor): flat.append(x)
                    elif isinstance(x, (tuple, list)):
                        for y in x: add(y)
                add(r)
                torch.cuda.synchronize()
                for j, t in enumerate(flat):
                    calls.append((f"{len(calls):04d} {name}[{j}] {tuple(t.shape)}", t.detach().float().cpu() if t.dtype != torch.uint8 else t.cpu()))
                return r
            setattr(_ops, name, g)
        for nm in ("ln_modulate", "gemm", "qkv_norm_rope_pack", "attn_dense", "rmsnorm_rope", "v_transpose", "patchify", "unpatchify", "timestep_embedding", "silu"):
            wrap(nm)
    for it in range(n):
        calls = [] if os.environ.get("OPS") else None
        rec = {} if os.environ.get("HOOK") else None
        if os.environ.get("POISON"):
            # every block the caching allocator hands out during the forward comes out of this freed NaN-filled block: a kernel that reads memory no
            # kernel wrote shows up as NaN (or as a forward that differs from the next)
            torch.cuda.synchronize()
            torch.full((int(os.environ["POISON"]) << 20,), float("nan") if it % 2 == 0 else 7.0, dtype=torch.bfloat16, device="cuda")
            torch.cuda.synchronize()
        tr = {}
        y = model(lat.cuda(), ctx.cuda(), torch.tensor([333.0]).cuda(), trace=tr)
        tr["y"] = y
        if rec is not None:
            tr.update(rec)
        if calls is not None:
            tr.update(dict(calls))
        outs.append({k: v.cpu() for k, v in tr.items()})
    return outs, model.sp


def worker(rank, world, port, heads, lat_shape, overlap, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), FVK_SP_OVERLAP=str(overlap))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    outs, sp = forward_traces(heads, lat_shape)
    if rank == int(os.environ.get("RANK_SAVE", "0")):   # whose traces are compared (the final output y is the same tensor on every rank)
        torch.save((outs, sp.overlap), f"/tmp/sp_det_{port}.pt")
        q.put("done")
    dist.barrier(); dist.destroy_process_group()


if __name__ == "__main__":
    world, heads, overlap = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    lat_shape = (1, 16, 5, 42, 62) if heads == 12 else (1, 16, 5, 18, 30)
    if world == 1:   # single process (no exchange): do consecutive forwards differ when the allocator's free memory is poisoned differently?
        outs, _ = forward_traces(heads, lat_shape, 4)
        for k in outs[0]:
            print(f"  {k:28s}", [f"fwd{i} vs fwd0: {int((outs[i][k] != outs[0][k]).sum())} differ, non-finite {int((~torch.isfinite(outs[i][k].float())).sum())}" for i in (1, 2, 3)])
        sys.exit(0)
    ref, _ = forward_traces(heads, lat_shape, 1)
    ctx = mp.get_context("spawn"); q = ctx.Queue()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    ps = [ctx.Process(target=worker, args=(r, world, port, heads, lat_shape, overlap, q)) for r in range(world)]
    [p.start() for p in ps]
    q.get(timeout=600)
    [p.join() for p in ps]
    outs, ov = torch.load(f"/tmp/sp_det_{port}.pt", weights_only=False)
    print(f"world {world} heads {heads} overlap requested {overlap} kept {ov} GEMM_IMPL={os.environ.get('GEMM_IMPL')}")
    if os.environ.get("HOOK"):
        print("  y vs SP = 1:", [f"fwd{i}: {int((outs[i]['y'] != ref[0]['y']).sum())} differ" for i in range(3)], "(rank", os.environ.get("RANK_SAVE", "0"), "traces)")
        for k in outs[0]:
            print(f"  {k:28s}", [f"fwd{i} vs fwd0: {int((outs[i][k] != outs[0][k]).sum())} of {outs[0][k].numel()} differ" for i in (1, 2)])
            for i in (1, 2):
                d = (outs[i][k] != outs[0][k])
                if d.any() and (".send" in k or ".recv" in k or ".o_blk" in k or ".out" in k):
                    idx = d.nonzero()
                    print(f"      fwd{i}: shape {tuple(d.shape)}; first differing indices {idx[:3].tolist()} ... last {idx[-1].tolist()}; distinct dim-0 {idx[:, 0].unique().tolist()[:8]}, dim-1 "
                          f"{idx[:, 1].unique().tolist()[:12]} ({idx[:, 1].unique().numel()} rows), dim-2 {idx[:, 2].unique().tolist()[:8] if d.dim() > 2 else ''}; values "
                          f"{outs[0][k][d][:4].tolist()} vs {outs[i][k][d][:4].tolist()}")
        sys.exit(0)
    if os.environ.get("BRIEF"):
        bad = [k for k in outs[0] if any(not torch.equal(outs[i][k], outs[0][k]) for i in (1, 2))]
        print("   forwards differ in:", bad if bad else "nothing", "| non-finite:", [k for k in outs[0] if any(not torch.isfinite(outs[i][k].float()).all() for i in range(3))])
        sys.exit(0)
    Sl = outs[0]["blocks.0.out"].shape[1]
    for k in outs[0]:
        a = outs[0][k]
        r = ref[0][k]
        r = r[:, :Sl] if (r.dim() == 3 and r.shape[1] != a.shape[1]) else r      # rank 0's shard of the SP = 1 residual stream
        vs_ref = "n/a" if r.shape != a.shape else f"{int((a != r).sum())} differ (max {(a.float() - r.float()).abs().max().item():.3g})"
        print(f"  {k:28s} fwd0 vs SP=1: {vs_ref:34s} fwd1 vs fwd0: {int((outs[1][k] != a).sum())} differ, fwd2 vs fwd0: {int((outs[2][k] != a).sum())} differ")
