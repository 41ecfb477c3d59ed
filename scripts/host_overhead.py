"""How long does the HOST need to issue one DiT forward (30 layers of ctypes launches + torch glue), and does the forward capture into a
HIP graph?  At SP = 8 a rank has ~30 ms of GPU work per forward, so the issue time bounds the scaling.  Not the contract bench.
usage: python scripts/host_overhead.py [--w 16] (latent [1,16,21,60,w]: a small token count makes the GPU side negligible)"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

ap = argparse.ArgumentParser()
ap.add_argument("--w", type=int, default=16)
ap.add_argument("--iters", type=int, default=5)
args = ap.parse_args()

import __graft_entry__ as G
G.build()
from fastvideo_amd import wan_config as WC
from fastvideo_amd.wan_dit import WanTransformer3DModelHip

dev = torch.device("cuda", 0)
cfg = WC.WAN21_T2V_1_3B
sd = WC.random_state_dict(cfg, seed=0, device=dev)
model = WanTransformer3DModelHip(sd, cfg.num_heads, cfg.head_dim, cfg.patch_size, cfg.eps, cfg.freq_dim, device=dev)
del sd
g = torch.Generator(device=dev).manual_seed(1)
out = {}
for name, shape in (("small", (1, 16, 21, 60, args.w)), ("cfg2", (1, 16, 21, 60, 104))):
    latent = torch.randn(shape, generator=g, device=dev).bfloat16()
    ctx = torch.randn((1, 512, cfg.text_dim), generator=g, device=dev).bfloat16()
    ts = torch.tensor([500.0], device=dev)
    for _ in range(2):
        y = model(latent, ctx, ts)
    torch.cuda.synchronize()
    issue, total = [], []
    for _ in range(args.iters):
        t0 = time.perf_counter()
        y = model(latent, ctx, ts)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        issue.append((t1 - t0) * 1e3)
        total.append((t2 - t0) * 1e3)
    rec = {"tokens": shape[2] * (shape[3] // 2) * (shape[4] // 2), "host_issue_ms": round(min(issue), 2), "forward_ms": round(min(total), 2)}
    # HIP graph capture of the whole forward (torch allocations inside the capture go to the graph's private pool)
    try:
        static_in = latent.clone()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(2):
                model(static_in, ctx, ts)
        torch.cuda.current_stream().wait_stream(s)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            y_g = model(static_in, ctx, ts)
        torch.cuda.synchronize()
        graph.replay()
        torch.cuda.synchronize()
        rec["graph_equals_eager"] = bool(torch.equal(y_g, y))
        rep = []
        for _ in range(args.iters):
            t0 = time.perf_counter()
            graph.replay()
            torch.cuda.synchronize()
            rep.append((time.perf_counter() - t0) * 1e3)
        rec["graph_replay_ms"] = round(min(rep), 2)
    except Exception as e:  # noqa: BLE001
        rec["graph_error"] = repr(e)[:300]
    out[name] = rec
print(json.dumps(out))
