"""Host issue time of one DiT forward (Wan2.1-1.3B, 30 layers, cfg2 latent), eager vs HIP graphs (WanTransformer3DModelHip.capture), at SP = 1 in
this process and at SP = WORLD (default 8) on processes that SHARE the one GPU and exchange through gloo (host-staged: the exchange TIMES mean
nothing here, the HOST SIDE of everything between the exchanges is what is measured).
  eager:  host issue = wall time of forward() minus the time spent inside the collectives (a host-staged collective also waits for the GPU to
          drain, so everything else is the host issuing ~20 launches per layer);
  graphs: host issue = the time spent in the graph launches of GraphSegments.replay() (2 segments per layer + 1).
usage: python scripts/graph_host_issue.py [world=8] -> one JSON object (profiles/r06_host_issue_sp8.json)"""
import json, os, socket, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def build():
    from fastvideo_amd import wan_config as WC
    from fastvideo_amd.wan_dit import WanTransformer3DModelHip
    dev = torch.device("cuda", 0)
    cfg = WC.WAN21_T2V_1_3B
    sd = WC.random_state_dict(cfg, seed=0, device=dev)
    model = WanTransformer3DModelHip(sd, cfg.num_heads, cfg.head_dim, cfg.patch_size, cfg.eps, cfg.freq_dim, device=dev)
    g = torch.Generator(device=dev).manual_seed(1)
    lat = torch.randn(WC.LATENT_81F_480P, generator=g, device=dev).bfloat16()
    ctx = torch.randn((1, 512, cfg.text_dim), generator=g, device=dev).bfloat16()
    return model, (lat, ctx, torch.tensor([500.0], device=dev))


def measure(model, inp, iters=4):
    sp = model.sp
    coll = {"s": 0.0}
    if sp.lay.P > 1:   # time spent inside the eager collectives (they block the host under gloo staging)
        for name in ("_a2a", "all_gather_unpad"):
            fn = getattr(sp, name)
            def wrap(*a, _fn=fn, **k):
                t0 = time.perf_counter(); r = _fn(*a, **k); coll["s"] += time.perf_counter() - t0; return r
            setattr(sp, name, wrap)
    for _ in range(2): y = model(*inp)
    torch.cuda.synchronize()
    issue, total = [], []
    for _ in range(iters):
        coll["s"] = 0.0
        t0 = time.perf_counter(); y = model(*inp); t1 = time.perf_counter()
        torch.cuda.synchronize(); t2 = time.perf_counter()
        issue.append((t1 - t0 - coll["s"]) * 1e3); total.append((t2 - t0) * 1e3)
    y_eager = y.clone()
    if sp.lay.P > 1:
        for name in ("_a2a", "all_gather_unpad"):
            delattr(sp, name)   # back to the class methods (capture replaces the collectives by cut points)
    replay = model.capture(*inp)
    seg = replay.segments
    for _ in range(2): yg = replay(*inp)
    torch.cuda.synchronize()
    g_issue, g_total = [], []
    for _ in range(iters):
        seg.host_s["graph_launch"] = seg.host_s["collective"] = 0.0
        t0 = time.perf_counter(); yg = replay(*inp); torch.cuda.synchronize(); t2 = time.perf_counter()
        g_issue.append(seg.host_s["graph_launch"] * 1e3); g_total.append((t2 - t0) * 1e3)
    return {"eager_host_issue_ms": round(min(issue), 3), "eager_forward_ms": round(min(total), 2),
            "graph_host_issue_ms": round(min(g_issue), 3), "graph_forward_ms": round(min(g_total), 2),
            "graphs_per_forward": seg.n_graphs, "collectives_per_forward": sum(1 for k, _ in seg.items if k == "call"),
            "graph_equals_eager": bool(torch.equal(yg, y_eager))}


def worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        model, inp = build()
        r = measure(model, inp, iters=3)
        if rank == 0:
            r["layout"] = f"G{model.sp.lay.G}xU{model.sp.lay.U}"
            q.put(r); q.close(); q.join_thread()
        dist.barrier()
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    out = {"what": __doc__.split("usage")[0].strip()}
    model, inp = build()
    out["sp1"] = measure(model, inp)
    del model, inp
    torch.cuda.empty_cache()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0)); port = s_.getsockname()[1]
    procs = [ctx.Process(target=worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs: p.start()
    out[f"sp{world}_shared_gpu"] = q.get(timeout=1200)
    for p in procs: p.join(timeout=300)
    print(json.dumps(out, indent=1))
