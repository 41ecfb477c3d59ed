"""N > 1 path on CPU: world_size 2 and 4 over gloo.  The 2-D Ulysses exchange (fastvideo_amd/distributed.py) must
reproduce single-process attention exactly (the layout code moves bytes, it does no arithmetic), including ragged
sequences (zero padding), head counts not divisible by the world size (U > 1) and the final all-gather+unpad —
the reference's own SP test asserts SP=2 == SP=1 (fastvideo/tests/distributed/test_sp_wan.py:198-281)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import wan_oracle as W


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _attn_fn(q, k, v, kv_len):
    """oracle attention on [S,h,D] with key padding beyond kv_len"""
    o = W.attention_fp32_ref(q.transpose(0, 1)[None], k[:kv_len].transpose(0, 1)[None], v[:kv_len].transpose(0, 1)[None],
                             q.shape[-1]**-0.5)
    return o[0].transpose(0, 1).to(q.dtype)


def _worker(rank, world, port, H, S, D, q, k, v, out_q, overlap=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), FVK_SP_OVERLAP="1" if overlap else "0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from fastvideo_amd.distributed import SequenceParallel
        sp = SequenceParallel(H)
        assert sp.overlap == overlap
        assert sp.lay.G * sp.lay.U == world and H % sp.lay.G == 0
        ql, kl, vl = (sp.shard(t[None], dim=1)[0] for t in (q, k, v))
        ol = sp.attention(ql, kl, vl, S, _attn_fn)
        if overlap and sp.lay.heads_per_group >= 2:  # the first pipelined call checked itself against the plain exchange and kept the mode
            assert sp._overlap_checked and sp.overlap
            assert torch.equal(sp.attention(ql, kl, vl, S, _attn_fn), ol)
        full = sp.all_gather_unpad(ol[None], S, dim=1)[0]
        # shard + gather round trip of a [B,S,d] activation (ragged S -> zero padded)
        x = torch.arange(2 * S * 6, dtype=torch.float32).view(2, S, 6)
        back = sp.all_gather_unpad(sp.shard(x, dim=1), S, dim=1)
        if rank == 0:
            out_q.put((full, back.equal(x), (sp.lay.G, sp.lay.U)))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("overlap", [False, True])
@pytest.mark.parametrize("world,H,S", [(2, 2, 37), (2, 3, 40), (4, 2, 45), (4, 12, 64), (2, 12, 33)])
def test_sp_attention_equals_single_process(world, H, S, overlap):
    """overlap=True: FVK_SP_OVERLAP — the rank's head group split in two chunks with asynchronous exchanges (odd head counts per group
    included: 12 heads on 4 ranks = 3 per group -> chunks of 2 and 1; one head per group falls back to the plain exchange)."""
    D = 16
    g = torch.Generator().manual_seed(world * 100 + H)
    q, k, v = (torch.randn((S, H, D), generator=g) for _ in range(3))
    ref = _attn_fn(q, k, v, S)
    ctx = mp.get_context("spawn")
    out_q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, H, S, D, q, k, v, out_q, overlap)) for r in range(world)]
    for p in procs:
        p.start()
    full, roundtrip_ok, (G, U) = out_q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert G * U == world
    assert roundtrip_ok
    assert full.shape == ref.shape
    # same arithmetic on the same values, only partitioned differently over heads / query blocks: tight tolerance
    torch.testing.assert_close(full, ref, atol=1e-5, rtol=1e-5)


def test_layout_choice():
    import math
    for H, P, G, U in [(12, 1, 1, 1), (12, 2, 2, 1), (12, 4, 4, 1), (12, 8, 4, 2), (40, 8, 8, 1), (12, 6, 6, 1), (12, 16, 4, 4)]:
        assert math.gcd(H, P) == G and P // G == U
