"""Mi355xCommunicator (fastvideo_amd/distributed.py) against the REFERENCE's DeviceCommunicatorBase semantics, world_size 2 over gloo.
The expected tensors are computed with a restatement of the reference's AllToAll4D.forward (base_device_communicator.py:137-183:
transpose / all_to_all_single / split+cat) so both all-to-all modes, all_gather, slice, all_reduce, gather, send/recv are pinned
value-for-value; when /root/reference is importable the reference class itself is run in the same processes."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _ref_a2a(x, world, scatter_dim, gather_dim, group=None):
    """restatement of DistributedAutograd.AllToAll4D.forward"""
    if scatter_dim == 2 and gather_dim == 1:
        bs, shard_seqlen, hn, hd = x.shape
        shard_hn = hn // world
        inp = x.transpose(0, 2).contiguous()
        out = torch.empty_like(inp)
        dist.all_to_all_single(out, inp, group=group)
        out = torch.cat(out.split(shard_hn), dim=1)
        return out.transpose(0, 2).contiguous()
    bs, seqlen, shard_hn, hd = x.shape
    shard_seqlen = seqlen // world
    inp = x.transpose(0, 2).contiguous()
    inp = inp.reshape(shard_hn, world, shard_seqlen, bs, hd).transpose(0, 1).reshape(shard_hn * world, shard_seqlen, bs, hd).contiguous()
    out = torch.empty_like(inp)
    dist.all_to_all_single(out, inp, group=group)
    return out.transpose(0, 2).contiguous()


def _worker(rank, world, port, out_q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from fastvideo_amd.distributed import Mi355xCommunicator
        comm = Mi355xCommunicator(dist.group.WORLD)
        g = torch.Generator().manual_seed(10 + rank)
        ok = {}
        x = torch.randn((2, 5, 4 * world, 8), generator=g)                       # [bs, s/P, hn, hd]
        y = comm.all_to_all_4D(x, 2, 1)
        ok["a2a_21"] = torch.equal(y, _ref_a2a(x, world, 2, 1)) and y.shape == (2, 5 * world, 4, 8)
        z = comm.all_to_all_4D(y, 1, 2)
        ok["a2a_12"] = torch.equal(z, _ref_a2a(y, world, 1, 2))
        ok["a2a_roundtrip"] = torch.equal(z, x)
        full = torch.arange(3 * 4 * world * 2, dtype=torch.float32).view(3, 4 * world, 2)
        sl = comm.slice(full, dim=1, scale_grad=False)
        ok["slice"] = torch.equal(sl, full[:, rank * 4:(rank + 1) * 4])
        ok["all_gather"] = torch.equal(comm.all_gather(sl, dim=1), full)
        ok["all_reduce"] = torch.equal(comm.all_reduce(torch.full((4,), float(rank + 1))), torch.full((4,), float(sum(range(1, world + 1)))))
        gat = comm.gather(torch.full((2, 2), float(rank)), dst=0, dim=0)
        ok["gather"] = (gat is None) if rank != 0 else torch.equal(gat, torch.cat([torch.full((2, 2), float(r)) for r in range(world)], 0))
        if rank == 0:
            comm.send(torch.arange(6.0))
            ok["sendrecv"] = True
        elif rank == 1:
            ok["sendrecv"] = torch.equal(comm.recv(torch.Size([6]), torch.float32), torch.arange(6.0))
        else:
            ok["sendrecv"] = True
        try:
            comm.all_to_all_4D(x, 3, 1)
            ok["bad_mode_raises"] = False
        except RuntimeError:
            ok["bad_mode_raises"] = True
        # the reference class itself, when its checkout is present
        from oracle import ref_loader as R
        if R.available():
            R.install()
            from fastvideo.distributed.device_communicators.base_device_communicator import DeviceCommunicatorBase
            ref = DeviceCommunicatorBase(dist.group.WORLD, device_group=dist.group.WORLD)
            ok["vs_reference_class"] = torch.equal(ref.all_to_all_4D(x, 2, 1), y) and torch.equal(ref.all_to_all_4D(y, 1, 2), z) \
                and torch.equal(ref.all_gather(sl, dim=1), full)
        out_q.put((rank, ok))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_communicator_matches_reference_semantics(world):
    ctx = mp.get_context("spawn")
    out_q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, out_q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [out_q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok in res:
        assert all(ok.values()), f"rank {rank}: {ok}"
