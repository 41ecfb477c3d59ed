"""HIP-graph capture of the DiT forward (round 6): ``WanTransformer3DModelHip.capture`` — one graph at SP = 1, graph segments
cut at the exchanges under sequence parallelism (distributed.GraphSegments) — must replay the eager forward bit for bit, for every attention
mode, on NEW input values, and under SP on processes sharing the GPU (gloo, host-staged exchanges between the graph segments)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

import _guard  # noqa: E402
import _mp  # noqa: E402
_guard.maybe_install()


def _model(attention, heads=6, layers=2, seed=4):
    from fastvideo_amd import wan_config as WC
    from fastvideo_amd.wan_dit import WanTransformer3DModelHip
    cfg = WC.WanConfig("graph", heads, 128, 768, layers, text_dim=64)
    sd = WC.random_state_dict(cfg, seed=seed, device="cpu", with_vsa_gate=(attention == "vsa"))
    return cfg, WanTransformer3DModelHip(sd, cfg.num_heads, attention=attention, device="cuda:0")


def _inputs(cfg, shape, seed):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g).bfloat16().cuda(), torch.randn((1, 40, cfg.text_dim), generator=g).bfloat16().cuda(),
            torch.tensor([float(100 + seed)]).cuda())


@pytest.mark.parametrize("attention,shape", [("dense", (1, 16, 5, 18, 30)), ("dense", (1, 16, 9, 32, 32)), ("vsa", (1, 16, 5, 18, 30)),
                                             ("sta", (1, 16, 6, 16, 16))])
def test_graph_replay_equals_eager(attention, shape):
    cfg, model = _model(attention)
    a, b = _inputs(cfg, shape, 12), _inputs(cfg, shape, 13)
    ya, yb = model(*a).clone(), model(*b).clone()
    replay = model.capture(*a)
    assert replay.segments.n_graphs == 1
    assert torch.equal(replay(*a), ya), "graph replay differs from the eager forward on the captured inputs"
    assert torch.equal(replay(*b), yb), "graph replay differs from the eager forward on new inputs"
    assert torch.equal(replay(*a), ya)
    assert torch.equal(model(*b), yb), "the eager path changed after a capture"
    with pytest.raises(ValueError):
        replay(a[0][..., :-2], a[1], a[2])


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, attention, shape, out_q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cfg, model = _model(attention)
        a, b = _inputs(cfg, shape, 12), _inputs(cfg, shape, 13)
        ya, yb = model(*a).clone(), model(*b).clone()
        replay = model.capture(*a)
        ra, rb = replay(*a).clone(), replay(*b).clone()
        if rank == 0:
            out_q.put(_mp.ship((ya.cpu(), yb.cpu(), ra.cpu(), rb.cpu(), replay.segments.n_graphs, sum(1 for k, _ in replay.segments.items if k == "call"))))
            out_q.close(); out_q.join_thread()
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,attention", [(2, "dense"), (3, "dense"), (2, "vsa")])
def test_graph_segments_under_sequence_parallelism(world, attention):
    """2 layers: per layer two exchanges (dense) cut the capture, plus the final all-gather -> 2 * 2 + 1 collectives between 2 * 2 + 2 graphs
    (vsa: one more exchange per layer where the plan is uneven).  Replay == eager == the SP = 1 forward."""
    shape = (1, 16, 5, 18, 30)
    cfg, model = _model(attention)
    a, b = _inputs(cfg, shape, 12), _inputs(cfg, shape, 13)
    ref_a, ref_b = model(*a).cpu(), model(*b).cpu()
    del model
    ctx = mp.get_context("spawn")
    out_q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, attention, shape, out_q)) for r in range(world)]
    for p in procs:
        p.start()
    ya, yb, ra, rb, n_graphs, n_calls = _mp.unship(out_q.get(timeout=300))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert n_calls >= 5 and n_graphs == n_calls + 1, (n_graphs, n_calls)
    assert torch.equal(ya, ref_a) and torch.equal(yb, ref_b), "eager SP forward differs from SP = 1"
    assert torch.equal(ra, ya) and torch.equal(rb, yb), "graph-segment replay differs from the eager SP forward"
