"""Sequence parallelism for the Wan DiT on one MI355X node: one process per GPU, ``torch.distributed``
(backend "nccl" = RCCL over xGMI on ROCm; "gloo" on CPU for the world_size-2 tests).

Reference being replaced: Ulysses SP — ``sequence_model_parallel_shard / all_to_all_4D / all_gather_with_unpad``
(fastvideo/distributed/communication_op.py:28-91, device_communicators/base_device_communicator.py:123-193,
fastvideo/distributed/utils.py:63-146) driven by ``DistributedAttention.forward`` (fastvideo/attention/layer.py:82-164).
The reference needs ``num_heads % sp_size == 0`` (wanvideo.py:606-607), so Wan2.1-1.3B (12 heads) cannot use 8 GPUs.

MI355X-first redesign — **2-D Ulysses**: the P ranks form a G x U grid, G = gcd(H, P) head groups x U = P/G query
blocks.  xGMI is a full point-to-point mesh (every pair of GPUs owns a link), so an all-to-all with per-peer
messages is not ring-bound; we use exactly that:
    exchange #1  K, V : every rank sends its token shard's head-group g slice to all U ranks of column g
                 Q    : ... only to the rank (g, u) whose query block u contains the shard
    attention    rank (g, u): queries of block u (G shards), heads of group g, ALL keys
    exchange #2  O    : rank (g, u) returns each shard's rows to the shard's owner
U = 1 is plain Ulysses (P | H).  For 12 heads on 8 GPUs: G = 4, U = 2 — perfectly balanced (3 heads x half the
queries per GPU) at 1.5x the K/V traffic, instead of the reference's hard failure.
RoPE and QK-norm are token-local, so they are applied *before* exchange #1 with global positions (the reference
applies RoPE after its all-to-all, layer.py:130-132 — same values, one fewer pass over the gathered tensor).

Exchange #1 is ONE ``all_to_all_single`` with equal per-peer blocks in a ROW-INTERLEAVED layout: the message for rank r' is
``[Sl, 3, W]`` = per token ``[K | V | Q]`` of r'`s head group (W = heads/G * head_dim).  The received buffer is then
``[P*Sl, 3, heads/G, head_dim]`` — the shape of a fused QKV projection — and attention reads K, V^T and this rank's query rows out of
it IN PLACE through strides: no unpack copy.  On the device the send buffer is written directly by the norm/RoPE kernel
(``ops.qkv_norm_rope_pack``): no pack copy either (the reference: ``torch.cat`` + two ``transpose().contiguous()`` per exchange,
layer.py:117-124, base_device_communicator.py:147-183).  Q travels to all U ranks of a column although only one needs it: xGMI
links are point-to-point, the exchange lasts as long as its fullest link, and the links to this rank's own query-block row carry
K+V+Q anyway — uniform blocks cost no wall time and keep the exchange a single equal-split collective.

All functions work on CPU tensors too (layout code is plain torch), which is how the gloo tests exercise them;
the attention itself is injected (``attn_fn``): the product passes the HIP kernel, tests pass the oracle.
"""
from __future__ import annotations

import math
import os
from dataclasses import dataclass

import torch
import torch.distributed as dist


@dataclass
class SPLayout:
    P: int      # SP world size
    rank: int   # rank in the SP group
    H: int      # attention heads
    G: int      # head groups  (gcd(H, P))
    U: int      # query blocks (P // G)

    @property
    def g(self):
        return self.rank % self.G

    @property
    def u(self):
        return self.rank // self.G

    @property
    def heads_per_group(self):
        return self.H // self.G


@dataclass
class BlockPlan:
    """Uneven exchange #2 of the tile-major sparse modes on a G x U grid (``SequenceParallel.block_plan``): rank (g, u) owns the
    tile-major rows [r0, r1) — whole query blocks, the u-th of U contiguous runs of the block list — of every head of group g."""
    r0: int
    r1: int
    send_tokens: torch.Tensor   # int32 [n_send]: this rank's real tokens, ascending = grouped by destination shard
    in_splits: list             # rows sent to each rank (the owner of each token's shard)
    out_splits: list            # rows received from each rank
    asm_idx: torch.Tensor       # int32 [Sl * G]: row of the received buffer holding (local token p, head group g'); pad tokens -> row 0
    n_recv: int


class GraphSegments:
    """One forward captured as HIP graphs CUT AT THE COLLECTIVES (round 6): everything a rank launches between two exchanges — ~10 kernels per
    segment, two segments per layer — becomes one graph launch, the exchanges themselves stay eager calls on the same buffers.  Built by
    ``WanTransformer3DModelHip.capture``; ``SequenceParallel`` cuts the running capture wherever it would call a collective (``cut``).  All
    segments share one memory pool and are replayed in capture order, so a tensor produced in one segment is still there for the next.
    P = 1: no cut, the whole forward is one graph.  Why: a rank of an SP = 8 run has ~34 ms of GPU work per forward against 8 ms of host issue
    (profiles/r02_host_issue_and_hipgraph.json); with the launches in graphs the host only issues 2 graphs + 2 collectives per layer."""

    def __init__(self):
        self.items = []     # ("graph", torch.cuda.CUDAGraph) | ("call", thunk re-running one collective on its capture-time buffers)
        self.pool = torch.cuda.graph_pool_handle()
        self._g = None
        self.host_s = {"graph_launch": 0.0, "collective": 0.0}   # host seconds spent in replay(), by kind (bench / scripts read it)

    def begin(self):
        self._g = torch.cuda.CUDAGraph()
        self._g.capture_begin(pool=self.pool)

    def cut(self, thunk):
        """End the running segment, run the collective now (eagerly, on the capture stream), remember it, open the next segment."""
        self._g.capture_end()
        self.items.append(("graph", self._g))
        thunk()
        self.items.append(("call", thunk))
        self.begin()

    def end(self):
        self._g.capture_end()
        self.items.append(("graph", self._g))
        self._g = None

    def replay(self):
        import time
        for kind, it in self.items:
            t0 = time.perf_counter()
            if kind == "graph":
                it.replay()
                self.host_s["graph_launch"] += time.perf_counter() - t0
            else:
                it()
                self.host_s["collective"] += time.perf_counter() - t0

    @property
    def n_graphs(self):
        return sum(1 for k, _ in self.items if k == "graph")


class SequenceParallel:

    def __init__(self, num_heads: int, group=None):
        self.group = group
        if dist.is_available() and dist.is_initialized():
            P, rank = dist.get_world_size(group), dist.get_rank(group)
        else:
            P, rank = 1, 0
        G = math.gcd(num_heads, P)
        self.lay = SPLayout(P=P, rank=rank, H=num_heads, G=G, U=P // G)
        # gloo cannot move device memory: stage through the host (used by the 2-process single-GPU parity test;
        # production runs use backend "nccl" = RCCL, which takes device pointers directly)
        self._stage_host = P > 1 and dist.get_backend(group) == "gloo"
        # Pipelined exchange (attention_packed_pipelined): a rank's head group is split in two chunks; both input exchanges are issued
        # asynchronously up front, so chunk B's all-to-all rides under chunk A's attention and chunk A's output exchange under chunk B's.
        # What it can save inside ONE batch-1 forward is bounded: every op of a DiT layer depends on the previous one, so the critical path
        # is still exchange #1 (all of it) -> attention -> exchange #2 of the LAST chunk; only the first chunk's share of exchange #2 (~1/6 of
        # a layer's exchange time) is hidden, and two attention launches in stream order split the grid (192 workgroups at SP = 8 become
        # 128 + 64; FVK_SP_OVERLAP=2 runs the second chunk on a second HIP stream instead).  Heads are independent in attention, so the result
        # is the plain exchange's bit for bit — checked on the first call (all-reduced verdict); any rank seeing a difference switches every
        # rank back to the plain exchange.
        # DEFAULT (round 6): the PLAIN exchange at every P.  Round 5 switched P >= 4 to the pipelined form on the strength of a compute-side
        # A/B only (profiles/r04z_sp_chunk_launch_ab.log: two chunk launches are free at P = 4 / 8, cost 18 % of the attention at P = 2);
        # what it HIDES needs xGMI links and has never been measured, so it stays opt-in until one 4 / 8-GPU RCCL run has
        # validated it for time and for hangs.  FVK_SP_OVERLAP = 0 / 1 / 2 forces plain / one stream / two streams; the mode in use is
        # logged once per process and printed in bench.py's line.
        mode = {"0": 0, "1": 1, "2": 2}.get(os.environ.get("FVK_SP_OVERLAP", "auto"), 0)
        if P > 1 and rank == 0:
            import logging
            logging.getLogger("fastvideo_amd").info("sequence-parallel exchange mode: %s (FVK_SP_OVERLAP=%s)",
                                                     ("plain", "pipelined, one stream", "pipelined, two streams")[mode],
                                                     os.environ.get("FVK_SP_OVERLAP", "unset"))
        self.overlap = P > 1 and mode > 0
        self.overlap_streams = 2 if mode == 2 else 1   # 2: chunk B's attention on a second HIP stream
        self._side_stream = None
        self._overlap_checked = False
        # bench.py's exchange accounting: set ``stats`` to a dict to collect, per exchange kind, the bytes this rank sends to OTHER
        # ranks and HIP-event pairs around the collective on the caller's stream (None = nothing recorded, nothing extra on the stream)
        self.stats = None
        # set by WanTransformer3DModelHip.capture while a forward is being captured: collectives cut the graph (GraphSegments)
        self.segmenter = None

    def _tick(self):
        if self.stats is None or self._stage_host:
            return None
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    def _tock(self, e0, kind: str, remote_bytes: int) -> None:
        if self.stats is None:
            return
        rec = self.stats.setdefault(kind, {"calls": 0, "remote_bytes": 0, "events": []})
        rec["calls"] += 1
        rec["remote_bytes"] += int(remote_bytes)
        if e0 is not None:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            rec["events"].append((e0, e1))

    def sum_over_ranks(self, values, device=None):
        """Element-wise sum of a short list of floats over the group (every rank gets the same list back): lets the ranks take ONE decision from
        per-rank measurements (wan_dit's in-place attention kernel choice).  P == 1: the values themselves."""
        vals = [float(v) for v in values]
        if self.lay.P == 1:
            return vals
        t = torch.tensor(vals, dtype=torch.float64, device="cpu" if (self._stage_host or device is None) else device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return [float(v) for v in t.cpu()]

    # -- sharding with zero padding (ref: distributed/utils.py:63-123, communication_op.py:61-91) ---------------
    def padded_len(self, S: int) -> int:
        return (S + self.lay.P - 1) // self.lay.P * self.lay.P

    def shard(self, x: torch.Tensor, dim: int = 1) -> torch.Tensor:
        """Zero-pad ``dim`` to a multiple of P and return this rank's contiguous shard."""
        P, r = self.lay.P, self.lay.rank
        if P == 1:
            return x
        S = x.shape[dim]
        Sp = self.padded_len(S)
        if Sp != S:
            pad_shape = list(x.shape)
            pad_shape[dim] = Sp - S
            x = torch.cat([x, x.new_zeros(pad_shape)], dim=dim)
        Sl = Sp // P
        return x.narrow(dim, r * Sl, Sl).contiguous()

    def all_gather_unpad(self, x: torch.Tensor, S: int, dim: int = 1) -> torch.Tensor:
        """ref: sequence_model_parallel_all_gather_with_unpad (communication_op.py:75-91)."""
        P = self.lay.P
        if P == 1:
            return x
        xm = x.movedim(dim, 0).contiguous()
        dev = xm.device
        if self.segmenter is not None:
            out = xm.new_empty((P * xm.shape[0], *xm.shape[1:]))

            def run():
                if self._stage_host:
                    xh = xm.cpu()
                    oh = xh.new_empty(out.shape)
                    dist.all_gather_into_tensor(oh, xh, group=self.group)
                    out.copy_(oh)
                else:
                    dist.all_gather_into_tensor(out, xm, group=self.group)
            self.segmenter.cut(run)
            return out[:S].movedim(0, dim).contiguous()
        if self._stage_host:
            xm = xm.cpu()
        out = xm.new_empty((P * xm.shape[0], *xm.shape[1:]))
        dist.all_gather_into_tensor(out, xm, group=self.group)
        return out[:S].movedim(0, dim).contiguous().to(dev)

    # -- the exchanges ------------------------------------------------------------------------------------------
    def _a2a(self, send: torch.Tensor, in_splits, out_splits, out_rows: int) -> torch.Tensor:
        dev = send.device
        if self.segmenter is not None:
            # graph segments: the collective cuts the capture and is re-run at every replay on these very buffers
            send_c = send.contiguous()
            recv = send_c.new_empty((out_rows, *send_c.shape[1:]))

            def run():
                if self._stage_host:
                    sh = send_c.cpu()
                    rh = sh.new_empty(recv.shape)
                    dist.all_to_all_single(rh, sh, output_split_sizes=out_splits, input_split_sizes=in_splits, group=self.group)
                    recv.copy_(rh)
                else:
                    dist.all_to_all_single(recv, send_c, output_split_sizes=out_splits, input_split_sizes=in_splits, group=self.group)
            self.segmenter.cut(run)
            return recv
        if self._stage_host:
            send = send.cpu()
        recv = send.new_empty((out_rows, *send.shape[1:]))
        dist.all_to_all_single(recv, send.contiguous(), output_split_sizes=out_splits, input_split_sizes=in_splits,
                               group=self.group)
        return recv.to(dev)

    def _a2a_async(self, send: torch.Tensor, in_splits, out_splits, out_rows: int):
        """Issue the exchange and return a thunk that completes it (RCCL: the collective runs on the process group's own stream and
        ``wait()`` orders the caller's stream behind it without blocking the host; gloo: host-staged, completes at the thunk)."""
        if self._stage_host:
            dev, send_h = send.device, send.cpu().contiguous()
            recv = send_h.new_empty((out_rows, *send_h.shape[1:]))
            work = dist.all_to_all_single(recv, send_h, output_split_sizes=out_splits, input_split_sizes=in_splits, group=self.group, async_op=True)
            return lambda: (work.wait(), recv.to(dev))[1]
        send = send.contiguous()
        recv = send.new_empty((out_rows, *send.shape[1:]))
        work = dist.all_to_all_single(recv, send, output_split_sizes=out_splits, input_split_sizes=in_splits, group=self.group, async_op=True)
        return lambda: (work.wait(), recv, send)[1]  # `send` stays referenced until the exchange has been waited for

    # ---- exchange #1, row-interleaved uniform layout ------------------------------------------------------------------------
    def pack_rows(self, q, k, v, gate=None):
        """Layout restatement (plain torch, any device) of what ``ops.qkv_norm_rope_pack`` writes on the GPU: q, k, v [Sl, H, D]
        (already normed / rotated) -> send [P, Sl, 3, W], message row = [K | V | Q] of the destination's head group; with ``gate``
        (the VSA compress gate, a fourth per-token per-head tensor) [P, Sl, 4, W] = [K | V | Q | gate].  Used by the CPU (gloo)
        tests and as the checker of the kernel's layout; the device path never calls it."""
        L = self.lay
        Sl, H, D = q.shape
        W = (H // L.G) * D
        parts = (k, v, q) if gate is None else (k, v, q, gate)
        rows = torch.stack([t.reshape(Sl, L.G, W) for t in parts], dim=2)           # [Sl, G, NS, W]
        return rows.permute(1, 0, 2, 3).repeat(L.U, 1, 1, 1).contiguous()           # rank rp = g + G*u' gets group g

    def exchange_rows(self, send: torch.Tensor) -> torch.Tensor:
        """send [P, Sl, NS, W] -> recv [P*Sl, NS, W] (NS = 3 or 4 slots): row n = s*Sl + m is token m of source rank s = global token n."""
        L = self.lay
        P, Sl, NS, W = send.shape
        t0 = self._tick()
        recv = self._a2a(send.reshape(P * Sl, NS * W), None, None, P * Sl)
        self._tock(t0, "exchange1", send.numel() * send.element_size() * (P - 1) // P)
        return recv.view(P * Sl, NS, W)

    def views_of(self, recv: torch.Tensor, D: int):
        """(q_blk [G*Sl, hg, D], k_all [P*Sl, hg, D], v_all [P*Sl, hg, D]) as strided VIEWS of the received buffer (row stride NS*W)."""
        L = self.lay
        n, NS, W = recv.shape
        Sl = n // L.P
        r4 = recv.view(n, NS, W // D, D)
        return r4[L.u * L.G * Sl:(L.u + 1) * L.G * Sl, 2], r4[:, 0], r4[:, 1]

    # ---- tile-major sparse modes (video-sparse / sliding-tile attention) on the G x U grid -------------------------------------
    def block_plan(self, token_of_row: torch.Tensor, Sl: int, block: int = 64) -> BlockPlan:
        """Integer plan of the sparse modes' output exchange.  ``token_of_row`` (int [S_pad], -1 = padding row) maps the tile-major
        padded rows of the WHOLE sequence to global tokens.  The query blocks (``block`` rows each) are cut into U contiguous runs; rank
        (g, u) computes run u for head group g — against all keys, which every rank of column g holds after exchange #1, as it holds
        every query and (VSA) every gate row, because Q travels with K and V.  A run's tokens are scattered over all shards (tiles cut
        across the raster order), so exchange #2 is an all-to-all with UNEVEN splits: rank (g, u) sends each shard owner the rows of that
        shard's tokens in run u (ascending token order), and a shard owner assembles [Sl, G, hg*D] from G x U sources through
        ``asm_idx``.  U = 1 degenerates to the equal-split exchange of the dense path (every run is the whole sequence) and is served by it.
        Everything here is host-side integer work, identical on every rank, cached by the caller per (grid, layout)."""
        L = self.lay
        tor = token_of_row.detach().cpu().to(torch.int64)
        n_blocks = tor.numel() // block
        if tor.numel() % block:
            raise ValueError("block_plan: the tile-major row count must be a multiple of the block size")
        cuts = [n_blocks * u // L.U * block for u in range(L.U + 1)]
        own = []                                   # per run: its real tokens, ascending
        for u in range(L.U):
            t = tor[cuts[u]:cuts[u + 1]]
            own.append(torch.sort(t[t >= 0]).values)
        counts = torch.stack([torch.bincount(torch.div(t, Sl, rounding_mode="floor"), minlength=L.P)[:L.P] for t in own])  # [U, P shards]
        mine, me = own[L.u], L.rank
        in_splits = [int(c) for c in counts[L.u]]
        out_splits = [int(counts[rp // L.G, me]) for rp in range(L.P)]
        offs = [0]
        for c in out_splits:
            offs.append(offs[-1] + c)
        run_of_token = torch.full((L.P * Sl,), -1, dtype=torch.int64)
        idx_in_run = torch.zeros((L.P * Sl,), dtype=torch.int64)
        for u in range(L.U):
            t = own[u]
            run_of_token[t] = u
            shard = torch.div(t, Sl, rounding_mode="floor")
            first = torch.searchsorted(t, shard * Sl)            # index of the shard's first token inside the run
            idx_in_run[t] = torch.arange(t.numel()) - first
        asm = torch.zeros((Sl, L.G), dtype=torch.int64)
        loc = torch.arange(me * Sl, (me + 1) * Sl)
        real = run_of_token[loc] >= 0
        for g in range(L.G):
            src = g + L.G * run_of_token[loc].clamp_min(0)
            asm[:, g] = torch.where(real, torch.tensor(offs)[src] + idx_in_run[loc], torch.zeros_like(src))
        return BlockPlan(r0=cuts[L.u], r1=cuts[L.u + 1], send_tokens=mine.to(torch.int32), in_splits=in_splits, out_splits=out_splits,
                         asm_idx=asm.reshape(-1).to(torch.int32), n_recv=offs[-1])

    @staticmethod
    def _take_rows(t: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
        """rows t[idx] of a [n, ...] tensor: one gather kernel on the device, index_select on the CPU (gloo tests)."""
        if t.is_cuda:
            from . import ops
            return ops.gather_rows(t.reshape(1, t.shape[0], -1), idx.numel(), src_index=idx.to(t.device))[0].view(idx.numel(), *t.shape[1:])
        return t.index_select(0, idx.to(torch.int64))

    def attention_blocks(self, send: torch.Tensor, plan: BlockPlan, block_fn, head_dim: int = 128) -> torch.Tensor:
        """Sparse tile-major attention under sequence parallelism.  ``send`` [P, Sl, NS, W] (``ops.qkv_norm_rope_pack`` with or without
        the gate slot, or ``pack_rows``).  ``block_fn(recv4, plan)`` gets the received buffer as [P*Sl, NS, hg, D] (token-major: slot 0 K,
        1 V, 2 Q, 3 gate; rows >= S are padding) and returns o [P*Sl, hg, D] in TOKEN order in which the rows of this rank's own tokens
        (the real tokens of tile-major rows [plan.r0, plan.r1)) are valid.  Returns this rank's shard [Sl, H, D]."""
        L = self.lay
        Sl = send.shape[1]
        recv = self.exchange_rows(send)
        n, NS, W = recv.shape
        o_tok = block_fn(recv.view(n, NS, W // head_dim, head_dim), plan)
        if L.U == 1:
            return self.scatter_seq_gather_heads(o_tok, Sl)   # every rank computed all tokens of its head group: the dense path's exchange
        hg, D = o_tok.shape[1], o_tok.shape[2]
        packed = self._take_rows(o_tok, plan.send_tokens)
        t0 = self._tick()
        got = self._a2a(packed, plan.in_splits, plan.out_splits, plan.n_recv)
        self._tock(t0, "exchange2", (packed.shape[0] - plan.in_splits[L.rank]) * hg * D * packed.element_size())
        return self._take_rows(got, plan.asm_idx).reshape(Sl, L.G * hg, D)

    def scatter_heads_gather_seq(self, q, k, v):
        """q,k,v: this rank's shard [Sl, H, D] (batch 1).  Returns (q_blk [G*Sl, hg, D], k_all [P*Sl, hg, D], v_all) — views of ONE
        received buffer.  A layer costs two collectives (this one and the output exchange)."""
        return self.views_of(self.exchange_rows(self.pack_rows(q, k, v)), q.shape[-1])

    # ---- exchange #1, per-tensor blocks with uneven splits (the pipelined mode's packing) ----------------------------------
    def _pack_qkv(self, q, k, v):
        """q, k, v [Sl, G*hg, D] (heads group-major) -> (send buffer, splits, Sl) of exchange #1."""
        L = self.lay
        Sl, H, D = q.shape
        hg = H // L.G
        by_group = lambda t: t.reshape(Sl, L.G, hg, D).permute(1, 0, 2, 3)  # [G, Sl, hg, D]: chunk g' = heads of group g'
        qg, kg, vg = by_group(q), by_group(k), by_group(v)
        mine = [(rp // L.G) == L.u for rp in range(L.P)]        # as destination: gets my Q; as source: its Q comes to me
        send = torch.cat([t for rp in range(L.P) for t in ((kg[rp % L.G], vg[rp % L.G], qg[rp % L.G]) if mine[rp]
                                                           else (kg[rp % L.G], vg[rp % L.G]))], 0)
        splits = [(3 if m else 2) * Sl for m in mine]
        return send, splits, Sl

    def _unpack_qkv(self, send, splits, Sl, sync=False, done=None):
        L = self.lay
        mine = [(rp // L.G) == L.u for rp in range(L.P)]
        recv = self._a2a(send, splits, splits, sum(splits)) if sync else done()
        ks, vs, qs, off = [], [], [], 0
        for s_ in range(L.P):
            ks.append(recv[off:off + Sl]); vs.append(recv[off + Sl:off + 2 * Sl])
            if mine[s_]:
                qs.append(recv[off + 2 * Sl:off + 3 * Sl])
            off += splits[s_]
        return torch.cat(qs, 0), torch.cat(ks, 0), torch.cat(vs, 0)

    def scatter_seq_gather_heads(self, o_blk: torch.Tensor, Sl: int) -> torch.Tensor:
        """o_blk [G*Sl, hg, D] (query block u, head group g) -> this rank's shard [Sl, H, D]."""
        L = self.lay
        hg, D = o_blk.shape[1], o_blk.shape[2]
        o_in, o_out = self._o_splits(Sl)
        t0 = self._tick()
        recv = self._a2a(o_blk.contiguous(), o_in, o_out, L.G * Sl)      # [G(g), Sl, hg, D]
        self._tock(t0, "exchange2", (L.G - 1) * Sl * hg * D * o_blk.element_size())
        return recv.reshape(L.G, Sl, hg, D).permute(1, 0, 2, 3).reshape(Sl, L.G * hg, D).contiguous()

    def _o_splits(self, Sl: int):
        L = self.lay
        o_in = [Sl if (rp // L.G) == L.u else 0 for rp in range(L.P)]   # rows j*Sl.. go to shard owner u*G+j
        o_out = [Sl if (s // L.G) == L.u else 0 for s in range(L.P)]    # from (g, u) for every g
        return o_in, o_out

    def _attention_pipelined(self, q, k, v, S: int, attn_fn):
        """FVK_SP_OVERLAP: two head chunks per group, every exchange asynchronous (see __init__).  Heads are independent in attention, so
        the result is the un-chunked one bit for bit."""
        L = self.lay
        Sl, H, D = q.shape
        hg = L.heads_per_group
        cuts = [(0, (hg + 1) // 2), ((hg + 1) // 2, hg)]
        sub = lambda t, a, b: t.reshape(Sl, L.G, hg, D)[:, :, a:b].reshape(Sl, L.G * (b - a), D)
        pend = []
        for a, b in cuts:  # issue both input exchanges up front
            send, splits, _ = self._pack_qkv(sub(q, a, b), sub(k, a, b), sub(v, a, b))
            pend.append((send, splits, self._a2a_async(send, splits, splits, sum(splits))))
        outs = []
        for (a, b), (send, splits, done) in zip(cuts, pend):
            q_blk, k_all, v_all = self._unpack_qkv(send, splits, Sl, done=done)
            o_blk = attn_fn(q_blk, k_all, v_all, S).contiguous()
            o_in, o_out = self._o_splits(Sl)
            outs.append((b - a, self._a2a_async(o_blk, o_in, o_out, L.G * Sl)))  # rides under the next chunk's attention
        out = q.new_empty((Sl, L.G, hg, D))
        for (a, b), (hc, done) in zip(cuts, outs):
            out[:, :, a:b] = done().reshape(L.G, Sl, hc, D).permute(1, 0, 2, 3)
        return out.reshape(Sl, H, D)

    def attention_packed(self, send: torch.Tensor, S: int, attn_fn, head_dim: int = 128):
        """The device path of ``attention``: ``send`` [P, Sl, 3, W] was written by ``ops.qkv_norm_rope_pack`` (norm + RoPE + packing in one
        kernel).  exchange #1 -> attention on strided views of the received buffer -> exchange #2.  Returns this rank's [Sl, H, D]."""
        Sl = send.shape[1]
        q_blk, k_all, v_all = self.views_of(self.exchange_rows(send), head_dim)
        return self.scatter_seq_gather_heads(attn_fn(q_blk, k_all, v_all, S), Sl)

    def exchange_rows_async(self, send: torch.Tensor):
        """exchange_rows issued asynchronously: returns a thunk that completes it on the stream current at call time."""
        P, Sl, NS, W = send.shape
        done = self._a2a_async(send.reshape(P * Sl, NS * W), None, None, P * Sl)
        return lambda: done().view(P * Sl, NS, W)

    def attention_packed_pipelined(self, sends, S: int, attn_fn, head_dim: int = 128):
        """The pipelined form of ``attention_packed``: ``sends`` = the two head-chunk send buffers of ``ops.qkv_norm_rope_pack(heads_a=...)``
        ([P, Sl, 3, Wa], [P, Sl, 3, Wb]).  Both input exchanges are issued asynchronously up front (RCCL runs them on the process group's own
        stream; ``wait()`` only orders the compute stream), chunk B's exchange rides under chunk A's attention, each chunk's output exchange is
        issued as soon as its attention is and chunk A's rides under chunk B's attention; the compute stream then waits for both and assembles
        [Sl, H, D] (heads in the un-chunked order).  ``FVK_SP_OVERLAP=1``: ONE compute stream (the chunks' attention launches in stream
        order); ``=2``: chunk B's receive + attention + output exchange on a second HIP stream, so the two chunk launches share the chip
        instead of each running a half-empty grid.  (The two-stream form was withdrawn mid-round-4 when consecutive forwards differed on the
        one-GPU box; the cause — a small kernel's wave sharing a SIMD with a gemm_w1 wave, DESIGN §5 — is fixed, and the form is back, bit-identical
        in the GPU tests; which of the two pays is a question for real xGMI.)
        CPU / gloo: the same order of operations (tests)."""
        import contextlib
        L = self.lay
        Sl = sends[0].shape[1]
        two = self.overlap_streams == 2 and sends[0].is_cuda
        main = torch.cuda.current_stream() if two else None
        if two:
            if self._side_stream is None:
                self._side_stream = torch.cuda.Stream()
            self._side_stream.wait_stream(main)   # the side stream starts behind everything issued so far (the pack pass, the allocator's reuse)
        streams = [main, self._side_stream] if two else [None, None]
        pend = [self.exchange_rows_async(s_) for s_ in sends]
        outs = []
        o_in, o_out = self._o_splits(Sl)
        for st, send, done in zip(streams, sends, pend):
            with (torch.cuda.stream(st) if st is not None else contextlib.nullcontext()):
                t0 = self._tick()
                recv = done()
                self._tock(t0, "exchange1", send.numel() * send.element_size() * (L.P - 1) // L.P)
                if st is not None and st is not main:
                    recv.record_stream(st)
                q_blk, k_all, v_all = self.views_of(recv, head_dim)
                o_blk = attn_fn(q_blk, k_all, v_all, S).contiguous()
                outs.append((o_blk.shape[1], self._a2a_async(o_blk, o_in, o_out, L.G * Sl), o_blk, recv, st))
        hg = sum(o[0] for o in outs)
        out = sends[0].new_empty((Sl, L.G, hg, head_dim))
        a = 0
        for hc, done, o_blk, _recv, st in outs:
            t0 = self._tick()
            if st is not None and st is not main:
                main.wait_stream(st)      # the side stream's attention (and its output exchange's staging) before the caller's stream reads them
            res = done()
            self._tock(t0, "exchange2", (L.G - 1) * Sl * hc * head_dim * o_blk.element_size())
            if st is not None and st is not main:
                res.record_stream(main)
                o_blk.record_stream(main)
            out[:, :, a:a + hc] = res.reshape(L.G, Sl, hc, head_dim).permute(1, 0, 2, 3)
            a += hc
        return out.reshape(Sl, L.G * hg, head_dim)

    def pipelined_agrees(self, o: torch.Tensor, ref: torch.Tensor) -> bool:
        """First-call verdict of the pipelined exchange, identical on every rank: True iff EVERY rank's pipelined result equals its plain one bit
        for bit; on False the pipelined mode is switched off (with a warning) and callers use ``ref``."""
        self._overlap_checked = True
        same = torch.equal(o, ref)
        ok = torch.tensor([1 if same else 0], dtype=torch.int32, device="cpu" if self._stage_host else o.device)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=self.group)
        if int(ok.item()) != 1:
            import warnings
            d_ = (o.float() - ref.float()).abs()
            warnings.warn("fastvideo_amd: the pipelined sequence-parallel exchange disagreed with the plain exchange on its first call "
                          f"(rank {self.lay.rank}: {'equal here' if same else f'{int((d_ > 0).sum())} of {d_.numel()} elements differ, max {d_.max().item():.4g}, non-finite {int((~torch.isfinite(o.float())).sum())}'}); "
                          "falling back to the plain exchange (set FVK_SP_OVERLAP=0 to silence)")
            self.overlap = False
            return False
        return True

    def attention(self, q, k, v, S: int, attn_fn, extra=None):
        """Distributed self-attention for one batch element.
        q,k,v: [Sl, H, D] shards (already normed + rotated).  S = true (unpadded) global sequence length.
        attn_fn(q [Sq, h, D], k [Skv, h, D], v [Skv, h, D], kv_len) -> o [Sq, h, D]; keys >= kv_len are padding."""
        L = self.lay
        if L.P == 1:
            return attn_fn(q, k, v, S) if extra is None else attn_fn(q, k, v, S, extra)
        Sl = q.shape[0]
        if self.overlap and extra is None and L.heads_per_group >= 2:
            o = self._attention_pipelined(q, k, v, S, attn_fn)
            if self._overlap_checked:
                return o
            self._overlap_checked = True
            ref = self.scatter_seq_gather_heads(attn_fn(*self.scatter_heads_gather_seq(q, k, v), S), Sl)
            ok = torch.tensor([1 if torch.equal(o, ref) else 0], dtype=torch.int32, device="cpu" if self._stage_host else o.device)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=self.group)
            if int(ok.item()) != 1:
                import warnings
                warnings.warn("fastvideo_amd: the pipelined sequence-parallel exchange disagreed with the plain exchange on its first call; "
                              "falling back to the plain exchange (set FVK_SP_OVERLAP=0 to silence)")
                self.overlap = False
                return ref
            return o
        q_blk, k_all, v_all = self.scatter_heads_gather_seq(q, k, v)
        if extra is None:
            o_blk = attn_fn(q_blk, k_all, v_all, S)
        else:
            # a fourth per-token, per-head tensor (the VSA compress gate, layer.py:172-245): it must cover the same rows as q,
            # which only holds for plain Ulysses (U == 1: every rank sees all tokens of its head group)
            if L.U != 1:
                raise NotImplementedError("token-aligned extra tensors need U == 1 (num_heads divisible by the SP world size)")
            hg, D = L.heads_per_group, extra.shape[-1]
            e_all = self._a2a(extra.reshape(Sl, L.G, hg, D).permute(1, 0, 2, 3).reshape(L.P * Sl, hg, D), [Sl] * L.P, [Sl] * L.P, L.P * Sl)
            o_blk = attn_fn(q_blk, k_all, v_all, S, e_all)
        return self.scatter_seq_gather_heads(o_blk, Sl)


class Mi355xCommunicator:
    """Inference-side drop-in for the reference's ``DeviceCommunicatorBase`` (fastvideo/distributed/device_communicators/
    base_device_communicator.py:196-277; returned by ``Platform.get_device_communicator_cls()``, platforms/rocm.py:114-116): same
    constructor and the same ``all_reduce / all_gather(dim) / slice(dim) / all_to_all_4D(scatter_dim, gather_dim) / gather / send /
    recv / destroy`` methods with the same results, on RCCL (backend "nccl" on ROCm) or gloo.

    ``all_to_all_4D`` keeps the reference's two modes (``:147-183``) but packs with ONE permuted copy per side instead of
    ``transpose().contiguous()`` + ``cat`` + ``transpose().contiguous()``: per-peer messages are contiguous [bs, s, hn/P, hd] blocks,
    which on xGMI's point-to-point mesh travel on P-1 links in parallel.  No autograd (the denoising path runs under no_grad)."""

    def __init__(self, cpu_group, device=None, device_group=None, unique_name: str = ""):
        self.device = device or torch.device("cpu")
        self.cpu_group, self.device_group, self.unique_name = cpu_group, device_group or cpu_group, unique_name
        self.rank = dist.get_rank(cpu_group)
        self.world_size = dist.get_world_size(cpu_group)
        self.ranks = dist.get_process_group_ranks(cpu_group)
        self.global_rank, self.global_world_size = dist.get_rank(), dist.get_world_size()
        self.rank_in_group = dist.get_group_rank(cpu_group, self.global_rank)

    def all_reduce(self, input_, op=dist.ReduceOp.SUM):
        out = input_.clone()
        if self.world_size > 1:
            dist.all_reduce(out, op=op, group=self.device_group)
        return out

    def all_gather(self, input_, dim: int = -1):
        if self.world_size == 1:
            return input_
        dim = dim % input_.dim()
        x = input_.movedim(dim, 0).contiguous()
        out = x.new_empty((self.world_size * x.shape[0], *x.shape[1:]))
        dist.all_gather_into_tensor(out, x, group=self.device_group)
        return out.movedim(0, dim).contiguous()

    def slice(self, input_, dim: int = -1, *, scale_grad: bool = False):
        if self.world_size == 1:
            return input_.contiguous()
        dim = dim % input_.dim()
        n = input_.shape[dim] // self.world_size
        return input_.narrow(dim, self.rank_in_group * n, n).contiguous()

    def all_to_all_4D(self, input_, scatter_dim: int = 2, gather_dim: int = 1):
        P = self.world_size
        if P == 1:
            return input_
        if input_.dim() != 4:
            raise AssertionError(f"input must be 4D tensor, got {input_.dim()} and shape {input_.shape}")
        if scatter_dim == 2 and gather_dim == 1:      # [bs, s/P, hn, hd] -> [bs, s, hn/P, hd]
            bs, s, hn, hd = input_.shape
            send = input_.reshape(bs, s, P, hn // P, hd).permute(2, 0, 1, 3, 4).contiguous()   # [P(dst), bs, s, hn/P, hd]
            recv = torch.empty_like(send)
            dist.all_to_all_single(recv, send, group=self.device_group)                        # [P(src = seq shard), ...]
            return recv.permute(1, 0, 2, 3, 4).reshape(bs, P * s, hn // P, hd)
        if scatter_dim == 1 and gather_dim == 2:      # [bs, s, hn/P, hd] -> [bs, s/P, hn, hd]
            bs, s, shn, hd = input_.shape
            send = input_.reshape(bs, P, s // P, shn, hd).permute(1, 0, 2, 3, 4).contiguous()  # [P(dst = seq shard), bs, s/P, shn, hd]
            recv = torch.empty_like(send)
            dist.all_to_all_single(recv, send, group=self.device_group)                        # [P(src = head group), ...]
            return recv.permute(1, 2, 0, 3, 4).reshape(bs, s // P, P * shn, hd)
        raise RuntimeError(f"Invalid scatter_dim={scatter_dim}, gather_dim={gather_dim}. "
                           f"Only (scatter_dim=2, gather_dim=1) and (scatter_dim=1, gather_dim=2) are supported.")

    def gather(self, input_, dst: int = 0, dim: int = -1):
        dim = dim % input_.dim()
        lst = [torch.empty_like(input_) for _ in range(self.world_size)] if self.rank_in_group == dst else None
        dist.gather(input_, lst, dst=self.ranks[dst], group=self.device_group)
        return torch.cat(lst, dim=dim) if self.rank_in_group == dst else None

    def send(self, tensor, dst: int | None = None) -> None:
        dist.send(tensor, self.ranks[(self.rank_in_group + 1) % self.world_size if dst is None else dst], self.device_group)

    def recv(self, size, dtype, src: int | None = None):
        t = torch.empty(size, dtype=dtype, device=self.device)
        dist.recv(t, self.ranks[(self.rank_in_group - 1) % self.world_size if src is None else src], self.device_group)
        return t

    def destroy(self) -> None:
        pass
