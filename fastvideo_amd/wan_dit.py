"""Host of the per-step Wan video-DiT forward on MI355X: the reference's ``WanTransformer3DModel.forward``
(fastvideo/models/dits/wanvideo.py:656-766) and ``WanTransformerBlock.forward`` (``:361-434``; ``_VSA`` variant
``:520-582``) re-expressed as a short chain of fused HIP kernels (fastvideo_amd.ops -> libfvk_amd.so).

Same constructor inputs as the reference module needs at run time (a reference ``state_dict`` with the reference's
parameter names, heads / head_dim / patch size / eps) and the same call signature
``forward(hidden_states[B,C,T,H,W], encoder_hidden_states[B,L,text_dim], timestep[B]) -> [B,C_out,T,H,W]``.

Kernel chain per block (S = tokens on this rank, d = model dim), with the reference op each replaces:
    ln_modulate                 norm1 + (1+scale)·x+shift                        wanvideo.py:393
    gemm [S,d]x[d,3d]           to_q | to_k | to_v fused (one pass over the activations)   :394-396
    rmsnorm_rope (q,k)          RMSNorm across heads + 3-D RoPE                   :398-401, layer.py:130-132
    v_transpose                 V -> MFMA-ready V^T                               (layout op, no reference equivalent)
    attention                   dense / video-sparse / sliding-tile               layer.py:147, backends/*
    gemm to_out                 + bias                                            :411
    ln_modulate(residual,gate)  ScaleResidualLayerNormScaleShift (affine LN)      :414-421
    gemm to_q(cross) -> rmsnorm; text K/V are projected once per forward for all layers
    attention (512 text keys)   LocalAttention                                    :188-222
    gemm to_out(cross)
    ln_modulate(residual)       ScaleResidualLayerNormScaleShift (no affine)      :425-427
    gemm fc_in + GELU-tanh epilogue;  gemm fc_out + gated-residual epilogue       :430-431
No eager / CPU fallback exists: every tensor op on the token axis is a HIP kernel; torch is used only for the
[B, 6, d]-sized modulation vectors, allocation and collectives."""
from __future__ import annotations

import math

import torch

from . import kernel_api, ops, rope
from .distributed import SequenceParallel

BF16 = torch.bfloat16


class WanTransformer3DModelHip:

    def __init__(self, state_dict: dict, num_heads: int, head_dim: int = 128, patch_size=(1, 2, 2), eps: float = 1e-6,
                 freq_dim: int = 256, attention: str = "dense", vsa_sparsity: float = 0.8, sta_window=(3, 3, 3),
                 sta_tile=(6, 8, 8), sp_group=None, device="cuda", quantization: str | None = None, attn_autotune: bool = False):
        if head_dim != 128:
            raise ValueError("the gfx950 attention kernels are specialised for head_dim 128 (Wan2.1 / Wan2.2)")
        if attention not in ("dense", "vsa", "sta"):
            raise ValueError(f"unknown attention mode {attention!r}")
        self.H, self.D, self.d = num_heads, head_dim, num_heads * head_dim
        self.patch, self.eps, self.freq_dim = tuple(patch_size), eps, freq_dim
        self.attention, self.vsa_sparsity = attention, vsa_sparsity
        self.sta_window, self.sta_tile = tuple(sta_window), tuple(sta_tile)
        self.device = torch.device(device)
        # quantization: None (bf16) | "fp8" (per-tensor scales) | "fp8_channel" (per-output-channel weights, per-token activations):
        # FP8Config(granularity=...) of fastvideo/layers/quantization/fp8_config.py applied to to_q/k/v/to_out (both attentions) and ffn
        if quantization not in (None, "fp8", "fp8_channel"):
            raise ValueError(f"unknown quantization {quantization!r}")
        self.quant = quantization
        self.sp = SequenceParallel(num_heads, sp_group)
        if attention == "sta" and self.sp.lay.U != 1:
            # sliding-tile queries are packed by window class into 256-row groups that share a KV list: cutting that list over U query
            # runs is not built; video-sparse attention (block lists per 64-row block) and dense attention run on any G x U grid
            raise NotImplementedError(f"sta attention needs num_heads divisible by the SP world size (heads {num_heads}, world {self.sp.lay.P}); "
                                      "vsa and dense attention run on any G x U grid")
        self.num_layers = 1 + max(int(k.split(".")[1]) for k in state_dict if k.startswith("blocks."))
        self._load(state_dict)
        self._vsa_cache = {}
        # sliding-tile list form: "grouped" (shipped) = queries packed by window class on 256-row workgroups; "tile" = one list per 384-token
        # tile (256-row workgroups + a 128-row remainder); "block128" = one list per 128-row query block on the 4-wave kernel (A/B, tests)
        self.sta_lists = "grouped"
        self.vsa_fold = True  # single GPU: tile(q), tile(k), tile(gate), untile(out) folded into the neighbouring kernels
        self.vsa_fold_v = True  # ... and tile(v) into V's block means and V^T pass (False: a gathered copy of V, the form until round 6)
        self.sta_fold = True  # single GPU: no gather passes at all (q / k scattered by the norm pass, V^T gathered, output scattered)
        self.attn_events = None  # set to a list to collect (start, end, Sq, Skv, heads) HIP-event pairs
        # Dense self-attention on long key axes has two kernels that agree to rounding (fvk_attn_dense_kernel_bf16: attn_w16 / attn_w64); which
        # is faster depends on the clock the device reaches between THIS model's other kernels (profiles/r03_attn_context.md).  attn_autotune:
        # the SECOND forward (the first one also pays module loads and allocator growth, and runs on a cold chip) alternates them layer by
        # layer, times every launch with HIP events, and keeps the faster one from then on (one synchronisation, once).  Off (default): the
        # library default for every launch, bit-identical forwards from the first one on.
        self.attn_kernel = ops.ATTN_KERNEL_DEFAULT
        self.attn_autotune = bool(attn_autotune)
        self.attn_tune_report = None
        self.dense_kernel_ran = None  # name of the kernel the LAST dense self-attention launch ran (bench.py reports it)
        self._tune = None
        self._forwards = 0
        self.vsa_trace = None    # set to a list to collect every layer's VSA block mask (tests)
        self.fuse_cross_residual = True  # the cross-attention residual add in the out-projection's epilogue (False: in the norm pass; A/B, tests)
        self.vt_gemm = True      # dense, single GPU, bf16: V projection written as V^T by its own GEMM (ops.gemm_vt); False = fused QKV + layout pass (A/B, tests)

    # ------------------------------------------------------------------ weights
    def _load(self, sd):
        dev = self.device
        b16 = lambda t: t.detach().to(device=dev, dtype=BF16).contiguous()
        f32 = lambda t: t.detach().to(device=dev, dtype=torch.float32).contiguous()
        g = lambda k: sd[k]
        w = {}
        pe = g("patch_embedding.proj.weight")
        w["pe_w"], w["pe_b"] = b16(pe.reshape(pe.shape[0], -1)), b16(g("patch_embedding.proj.bias"))
        for n in ("time_embedder.mlp.fc_in", "time_embedder.mlp.fc_out", "time_modulation.linear", "text_embedder.fc_in",
                  "text_embedder.fc_out"):
            w[n + ".w"], w[n + ".b"] = b16(g(f"condition_embedder.{n}.weight")), b16(g(f"condition_embedder.{n}.bias"))
        w["proj_out.w"], w["proj_out.b"] = b16(g("proj_out.weight")), b16(g("proj_out.bias"))
        self.out_table = b16(g("scale_shift_table"))  # [1,2,d] kept in the parameter dtype (bf16 math, wanvideo.py:754)
        self.w = w
        self.vsa_gate = any(k.endswith("to_gate_compress.weight") for k in sd)
        L = self.num_layers
        blocks = []
        tables = []
        kv_w, kv_b = [], []
        for i in range(L):
            p = f"blocks.{i}."
            b = {}
            qkv_names = ["to_q", "to_k", "to_v"] + (["to_gate_compress"] if self.vsa_gate and self.attention == "vsa" else [])
            b["qkv_w"] = b16(torch.cat([g(p + n + ".weight") for n in qkv_names], 0))
            b["qkv_b"] = b16(torch.cat([g(p + n + ".bias") for n in qkv_names], 0))
            b["n_qkv"] = len(qkv_names)
            b["nq_w"], b["nk_w"] = b16(g(p + "norm_q.weight")), b16(g(p + "norm_k.weight"))
            b["o_w"], b["o_b"] = b16(g(p + "to_out.weight")), b16(g(p + "to_out.bias"))
            b["ln2_w"], b["ln2_b"] = f32(g(p + "self_attn_residual_norm.norm.weight")), f32(g(p + "self_attn_residual_norm.norm.bias"))
            b["cq_w"], b["cq_b"] = b16(g(p + "attn2.to_q.weight")), b16(g(p + "attn2.to_q.bias"))
            b["cnq_w"], b["cnk_w"] = b16(g(p + "attn2.norm_q.weight")), b16(g(p + "attn2.norm_k.weight"))
            b["co_w"], b["co_b"] = b16(g(p + "attn2.to_out.weight")), b16(g(p + "attn2.to_out.bias"))
            b["f1_w"], b["f1_b"] = b16(g(p + "ffn.fc_in.weight")), b16(g(p + "ffn.fc_in.bias"))
            b["f2_w"], b["f2_b"] = b16(g(p + "ffn.fc_out.weight")), b16(g(p + "ffn.fc_out.bias"))
            kv_w += [g(p + "attn2.to_k.weight"), g(p + "attn2.to_v.weight")]
            kv_b += [g(p + "attn2.to_k.bias"), g(p + "attn2.to_v.bias")]
            tables.append(g(p + "scale_shift_table"))
            blocks.append(b)
        self.blocks = blocks
        # all layers' text K/V projections as ONE GEMM over the 512 text tokens: [L*2d, d]
        self.ckv_w, self.ckv_b = b16(torch.cat(kv_w, 0)), b16(torch.cat(kv_b, 0))
        if self.quant:
            # convert_model_to_fp8 (fp8_config.py:211-245): every tagged linear is quantised on its own (own scale); a fused weight
            # [sum N_i, K] therefore carries a per-row scale vector that is constant inside each original matrix (tensor granularity)
            row = self.quant == "fp8_channel"

            def q8(wcat, n_parts):
                parts = wcat.chunk(n_parts, 0)
                qs, ss = zip(*[ops.fp8_quantize(p_.contiguous(), rowwise=row) for p_ in parts])
                scales = [s_.view(-1) if row else s_.expand(p_.shape[0]) for s_, p_ in zip(ss, parts)]
                return torch.cat(qs, 0).contiguous(), torch.cat(scales, 0).contiguous()

            for b in blocks:
                if b["n_qkv"] == 4:  # to_gate_compress is not an fp8-tagged layer (fp8_config.py:31-44): it stays a bf16 GEMM
                    b["gate_w"], b["gate_b"] = b["qkv_w"][3 * self.d:].contiguous(), b["qkv_b"][3 * self.d:].contiguous()
                    b["qkv_w"], b["qkv_b"] = b["qkv_w"][:3 * self.d].contiguous(), b["qkv_b"][:3 * self.d].contiguous()
                b["qkv_q"], b["qkv_s"] = q8(b["qkv_w"], 3)
                for k_ in ("o", "cq", "co", "f1", "f2"):
                    b[k_ + "_q"], b[k_ + "_s"] = q8(b[k_ + "_w"], 1)
                    del b[k_ + "_w"]
                del b["qkv_w"]
            self.ckv_q, self.ckv_s = q8(self.ckv_w, 2 * L)
            del self.ckv_w
        self.tables = b16(torch.stack(tables, 0))  # [L,1,6,d] in parameter dtype (bf16 + fp32 temb -> fp32, :386-390)

    # ------------------------------------------------------------------ attention variants
    def _vsa_meta(self, grid):
        m = self._vsa_cache.get(grid)
        if m is None:
            h = ops.vsa_build_metadata_host(grid)
            m = {k: (v.to(self.device) if isinstance(v, torch.Tensor) else v) for k, v in h.items()}
            m["S_pad"] = math.prod(m["num_tiles"]) * 64
            m["topk"] = max(1, min(math.ceil((1 - self.vsa_sparsity) * m["variable_block_sizes"].numel()),
                                   m["variable_block_sizes"].numel()))
            # row maps of the gather-free path (_vsa_fused): token -> tile-major padded row, and back (-1 = padding row)
            n_tok = h["tile_partition_indices"].numel()
            row = torch.empty(n_tok, dtype=torch.int32)
            row[h["tile_partition_indices"].long()] = h["non_pad_index"].to(torch.int32)
            tok = torch.full((m["S_pad"],), -1, dtype=torch.int32)
            tok[row.long()] = torch.arange(n_tok, dtype=torch.int32)
            m["row_of_token"], m["token_of_row"] = row.to(self.device), tok.to(self.device)
            tok128 = torch.full(((m["S_pad"] + 127) // 128 * 128,), -1, dtype=torch.int32)  # v_transpose walks whole 128-key tiles
            tok128[:m["S_pad"]] = tok
            m["token_of_row_128"] = tok128.to(self.device)
            self._vsa_cache[grid] = m
        return m

    def _sta_meta(self, grid, n_heads):
        """Tile permutation + per-(head, query block) KV block lists of the sliding-tile window.  Pure integer work on the host
        (cached); every head of this model uses the same window ``self.sta_window`` (the reference API takes one per head)."""
        key = (grid, n_heads)
        m = self._vsa_cache.get(("sta",) + key)
        if m is None:
            h = kernel_api.sliding_tile_block_lists(grid, self.sta_tile, self.sta_window)
            dev = self.device
            # row maps of the gather-free path (_sta_fused): token -> K row (tile-major, padded), V^T key position -> token, grouped query
            # row -> token
            n_tok = h["tile_partition_indices"].numel()
            k_row = torch.empty(n_tok, dtype=torch.int32)
            k_row[h["tile_partition_indices"].long()] = h["non_pad_index"].to(torch.int32)
            v_src = torch.full(((h["S_pad"] + 127) // 128 * 128,), -1, dtype=torch.int32)
            v_src[k_row.long()] = torch.arange(n_tok, dtype=torch.int32)
            g_tok = torch.full((h["group_rows"],), -1, dtype=torch.int32)
            g_tok[h["group_dst"].long()] = h["group_src"]
            m = dict(k_row_of_token=k_row.to(dev), v_src_rows=v_src.to(dev), group_token_of_row=g_tok.to(dev),
                     S_pad=h["S_pad"], perm=h["tile_partition_indices"].to(dev), non_pad=h["non_pad_index"].to(dev),
                     untile=h["untile_combined_index"].to(dev), block_sizes=h["block_sizes"].to(dev),
                     q2k_idx=h["q2k_idx"].to(dev)[None, None].expand(1, n_heads, -1, -1).contiguous(),
                     q2k_num=h["q2k_num"].to(dev)[None, None].expand(1, n_heads, -1).contiguous(),
                     q_block=h["q_block"], density=h["density"], tile_tokens=h["tile_tokens"],
                     tile_q2k_idx=h["tile_q2k_idx"].to(dev)[None, None].expand(1, n_heads, -1, -1).contiguous(),
                     tile_q2k_num=h["tile_q2k_num"].to(dev)[None, None].expand(1, n_heads, -1).contiguous(),
                     tile_rows_valid=h["tile_rows_valid"].to(dev), group_rows=h["group_rows"],
                     group_src=h["group_src"].to(dev), group_dst=h["group_dst"].to(dev), group_untile=h["group_untile"].to(dev),
                     group_q2k_idx=h["group_q2k_idx"].to(dev)[None, None].expand(1, n_heads, -1, -1).contiguous(),
                     group_q2k_num=h["group_q2k_num"].to(dev)[None, None].expand(1, n_heads, -1).contiguous())
            self._vsa_cache[("sta",) + key] = m
        return m

    def _tile_bufs(self, S_pad, h, n, grid, mode):
        # keyed by the GRID, not by S_pad: two grids can share S_pad with their pad rows in different places ((8,8,8) and (7,8,8)
        # both pad to 512), and pad rows are zeroed only at allocation — block means divide the sum over all 64 rows
        key = ("tilebuf", mode, tuple(grid), S_pad, h)
        bufs = self._vsa_cache.get(key)
        if bufs is None or len(bufs) < n:
            bufs = [torch.zeros((1, S_pad, h, self.D), dtype=BF16, device=self.device) for _ in range(n)]
            self._vsa_cache[key] = bufs
        return bufs

    def _finish_attn_tune(self):
        """Close the timing forward: the faster long-key kernel (median launch time, the first launch of each left out) is kept."""
        ev, self._tune = self._tune, None
        self.attn_autotune = False
        if len(ev) < 6:  # too few long-key launches to compare (short sequences take the 8-wave kernel anyway); the same count on every rank
            return
        torch.cuda.synchronize()
        ms = {ops.ATTN_KERNEL_W16: [], ops.ATTN_KERNEL_W64: []}
        for kern, e0, e1 in ev[2:]:
            ms[kern].append(e0.elapsed_time(e1))
        med = {kk: sorted(v)[len(v) // 2] for kk, v in ms.items()}
        if self.sp.lay.P > 1:  # sequence parallel: ONE decision for all ranks, from the sum of their medians
            tot = self.sp.sum_over_ranks([med[ops.ATTN_KERNEL_W16], med[ops.ATTN_KERNEL_W64]], device=self.device)
            med = {ops.ATTN_KERNEL_W16: tot[0] / self.sp.lay.P, ops.ATTN_KERNEL_W64: tot[1] / self.sp.lay.P}
        self.attn_kernel = min(med, key=med.get)
        self.attn_tune_report = {"attn_w16_ms": round(med[ops.ATTN_KERNEL_W16], 4), "attn_w64_ms": round(med[ops.ATTN_KERNEL_W64], 4),
                                 "launches_timed": len(ev) - 2, "kept": "attn_w16" if self.attn_kernel == ops.ATTN_KERNEL_W16 else "attn_w64"}

    def _dense_attn(self, q4, k4, v4, vt=None):
        """Dense self-attention of ALL batch elements in ONE launch: q4 [B,Sq,h,D], k4 / v4 [B,Skv,h,D] (strided views ok) -> o [B,Sq,h,D].
        ``vt``: a ready V^T [B,h,D,Skv_pad] (the V projection written in that layout by ops.gemm_vt; then v4 is ignored).
        The reference runs the classifier-free-guidance pair as two forwards (denoising.py:497-560); batched here (DenoisingLoopHip(cfg_batch=
        True)) the pair shares every launch — per sample the arithmetic is the same (rows and batch elements are independent in every kernel)."""
        if vt is None:
            vt = ops.v_transpose(v4)
        kern, tune = self.attn_kernel, self._tune
        # key runs are decided for the rank's WHOLE head group (the pipelined exchange launches it as two head chunks on two streams: they
        # fill the chip together, and must run the arithmetic of the un-chunked launch)
        heads = max(q4.shape[2], self.sp.lay.heads_per_group if self.sp.lay.P > 1 else q4.shape[2])
        # ... and from ONE sample's grid: the split count sets the merge rounding, so a batch-2 (cfg_batch) launch must take the count its
        # samples take alone, or its halves would not equal their stand-alone forwards bit for bit at under-filled geometries
        splits = ops.attn_key_splits(-(-q4.shape[1] // 256) * heads, -(-k4.shape[1] // 128)) if q4.shape[1] >= 256 else 1
        long_keys = q4.shape[1] >= 256 and k4.shape[1] >= 2048
        if splits > 1 or not long_keys:
            # short key axes take the 8-wave kernel and split-KV grids always run attn_w16 (fvk_attn_dense_split_bf16): nothing to choose
            # between, nothing to time — and the kernel a caller may report is the one that ran
            kern, tune = ops.ATTN_KERNEL_DEFAULT, None
        elif tune is not None:
            kern = (ops.ATTN_KERNEL_W16, ops.ATTN_KERNEL_W64)[len(tune) % 2]
        self.dense_kernel_ran = (f"attn_w16 split-KV x{splits}" if splits > 1 else "attn_pp2" if not long_keys else
                                 "attn_w64" if kern == ops.ATTN_KERNEL_W64 else "attn_w16")
        if self.attn_events is None and tune is None:
            return ops.attn_dense(q4, k4, vt=vt, scale=self.D**-0.5, layout="bshd", kernel=kern, key_splits=splits)
        # bench.py roofline leg / the in-place kernel choice: HIP events on the launch stream around the dominant kernel only
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        o = ops.attn_dense(q4, k4, vt=vt, scale=self.D**-0.5, layout="bshd", kernel=kern, key_splits=splits)
        e1.record()
        if self.attn_events is not None:
            self.attn_events.append((e0, e1, q4.shape[1], k4.shape[1], q4.shape[2] * q4.shape[0]))
        if tune is not None:
            tune.append((kern, e0, e1))
        return o

    def _attn_local(self, q, k, v, kv_len, grid, gate=None):
        if self.attn_events is None or self.attention == "dense":
            return self._attn_local_impl(q, k, v, kv_len, grid, gate)
        # bench.py roofline leg for the sparse modes: HIP events around the whole attention (tile + kernels + untile)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        o = self._attn_local_impl(q, k, v, kv_len, grid, gate)
        e1.record()
        self.attn_events.append((e0, e1, kv_len, kv_len, q.shape[1]))
        return o

    def _attn_local_impl(self, q, k, v, kv_len, grid, gate=None):
        """q [Sq,h,D], k/v [Skv,h,D] (strided views ok) -> o [Sq,h,D] contiguous."""
        q4, k4, v4 = q.unsqueeze(0), k[:kv_len].unsqueeze(0), v[:kv_len].unsqueeze(0)
        if self.attention == "dense":
            return self._dense_attn(q4, k4, v4)[0]
        if self.attention == "vsa":
            # ref: VideoSparseAttentionImpl.preprocess_qkv / forward / postprocess_output (video_sparse_attn.py:254-342)
            # Everything stays [1, S_pad, h, D] ("bshd"): the kernels take strides, so there is no transpose / contiguous copy; the four
            # tile buffers are allocated (zeroed) once per shape and reused by every layer — pad rows are never written, so they stay
            # zero (the reference keeps one such `tile_buf` per step too, video_sparse_attn.py:254-264)
            m = self._vsa_meta(grid)
            S = kv_len
            bufs = self._tile_bufs(m["S_pad"], q.shape[1], 4 if gate is not None else 3, grid, "vsa")
            tile = lambda t, j: ops.gather_rows(t[:, :S], m["S_pad"], m["tile_partition_indices"], m["non_pad_index"], out=bufs[j])
            tq, tk, tv = tile(q4, 0), tile(k4, 1), tile(v4, 2)
            tg = tile(gate.unsqueeze(0), 3) if gate is not None else None
            vbs = m["variable_block_sizes"]
            if self.vsa_trace is None:
                o = kernel_api.video_sparse_attn_bshd(tq, tk, tv, vbs, vbs, m["topk"], 64, tg)
            else:  # tests: keep every layer's block selection so that an oracle can be evaluated with the SAME selection
                o, inter = kernel_api._vsa_forward(tq, tk, tv, vbs, vbs, m["topk"], tg, "bshd", True)
                self.vsa_trace.append(inter["mask"])
            o = ops.gather_rows(o, S, m["untile_combined_index"], None)
            if q.shape[0] != S:
                o = torch.cat([o, o.new_zeros((1, q.shape[0] - S, *o.shape[2:]))], 1)
            return o[0]
        # sliding-tile attention (ref kernel API: fastvideo_kernel.sliding_tile_attention, ops.py:21-62; mask semantics
        # fastvideo-kernel/tests/support_flex_sta.py:29-59).  The token grid is padded to whole tiles; tokens are gathered tile-major
        # with each tile's real tokens first, and the window rule is expressed as KV 64-blocks with variable sizes for the
        # block-sparse kernel — which serves any canvas (the reference kernels hard-code three).
        m = self._sta_meta(grid, q.shape[1])
        S = kv_len
        bufs = self._tile_bufs(m["S_pad"], q.shape[1], 3, grid, "sta")
        tile = lambda t, j: ops.gather_rows(t[:, :S], m["S_pad"], m["perm"], m["non_pad"], out=bufs[j])  # [1,S_pad,h,D]
        if self.sta_lists == "grouped":
            # queries packed by WINDOW CLASS (tiles whose clamped windows coincide share one KV list): no per-tile query padding, every
            # query row on a 256-row workgroup of the dense kernel's schedule; K / V stay tile-major
            qg = self._tile_bufs(m["group_rows"], q.shape[1], 1, grid, "sta_q")[0]
            ops.gather_rows(q4[:, :S], m["group_rows"], m["group_src"], m["group_dst"], out=qg)
            o = ops.attn_tile_lists(qg, tile(k4, 1), tile(v4, 2), m["group_q2k_idx"], m["group_q2k_num"], m["block_sizes"], 256, None,
                                    scale=self.D**-0.5, layout="bshd")
            o = ops.gather_rows(o, S, m["group_untile"], None)
            if q.shape[0] != S:
                o = torch.cat([o, o.new_zeros((1, q.shape[0] - S, *o.shape[2:]))], 1)
            return o[0]
        if m["tile_tokens"] >= 256 and m["tile_tokens"] % 128 == 0 and self.sta_lists == "tile":
            # one list per tile: 256 of a tile's query rows share a workgroup on the dense kernel's schedule (fvk_attn_tile_lists_bf16)
            o = ops.attn_tile_lists(tile(q4, 0), tile(k4, 1), tile(v4, 2), m["tile_q2k_idx"], m["tile_q2k_num"], m["block_sizes"],
                                    m["tile_tokens"], m["tile_rows_valid"], scale=self.D**-0.5, layout="bshd")
        else:
            o = ops.attn_block_sparse(tile(q4, 0), tile(k4, 1), tile(v4, 2), m["q2k_idx"], m["q2k_num"], m["block_sizes"], scale=self.D**-0.5,
                                      layout="bshd", q_block=m["q_block"])
        o = ops.gather_rows(o, S, m["untile"], None)
        if q.shape[0] != S:
            o = torch.cat([o, o.new_zeros((1, q.shape[0] - S, *o.shape[2:]))], 1)
        return o[0]

    def _vsa_fused(self, rows, b, cos, sin, S, grid):
        """Video-sparse self-attention of one sample (single GPU) with NO gather pass: the QK-norm / RoPE pass scatters q and k into the
        tile-major padded layout, V's block means and its V^T layout pass read the token-order rows through the tile map (round 6; a gathered
        copy of V until then), the combine pass reads the compress gate in token order and writes the result in token order (tile(q), tile(k),
        tile(v), tile(gate) and untile(out) folded away; ref video_sparse_attn.py:254-342).  rows [S, 4d] -> o [S, H, D]."""
        d, H, D = self.d, self.H, self.D
        m = self._vsa_meta(grid)
        has_gate = rows.shape[1] >= 4 * d
        tq, tk, tv = self._tile_bufs(m["S_pad"], H, 3, grid, "vsa")[:3]
        ops.rmsnorm_rope([rows[:, :d], rows[:, d:2 * d]], [b["nq_w"], b["nk_w"]], cos, sin, head_dim=D, seq_len=S, eps=self.eps,
                         outs=[tq.view(-1, d), tk.view(-1, d)], row_maps=[m["row_of_token"], m["row_of_token"]])
        ev = None
        if self.attn_events is not None:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        v_tok = rows[:, 2 * d:3 * d].view(1, S, H, D)
        v_map = m["token_of_row_128"] if self.vsa_fold_v else None
        if v_map is None:  # A/B: the gathered copy
            ops.gather_rows(v_tok, m["S_pad"], m["tile_partition_indices"], m["non_pad_index"], out=tv)
        gate = rows[:, 3 * d:4 * d].view(1, S, H, D) if has_gate else None
        vbs = m["variable_block_sizes"]
        want = self.vsa_trace is not None
        res = kernel_api._vsa_forward(tq, tk, tv if v_map is None else v_tok, vbs, vbs, m["topk"], gate, "bshd", want, 64,
                                      token_of_row=m["token_of_row"], n_tokens=S, v_src_rows=v_map)
        if want:  # tests: keep every layer's block selection so that an oracle can be evaluated with the SAME selection
            res, inter = res
            self.vsa_trace.append(inter["mask"])
        if ev is not None:
            ev[1].record()
            self.attn_events.append((ev[0], ev[1], S, S, H))
        return res[0]

    def _sta_fused(self, rows, b, cos, sin, S, grid):
        """Sliding-tile self-attention of one sample with NO gather pass (single GPU): the QK-norm / RoPE pass scatters q into the
        window-class packed layout and k into the tile-major one (fvk_rmsnorm_rope_scatter_bf16), V goes from token order straight to the
        tile-major V^T (fvk_v_transpose_gather_bf16), and the attention epilogue stores every query row at its token's row (o_rows).
        rows [S, 3d(+d)] = the fused QKV GEMM output -> o [S, H, D] in token order."""
        d, H, D = self.d, self.H, self.D
        m = self._sta_meta(grid, H)
        qg = self._tile_bufs(m["group_rows"], H, 1, grid, "sta_q")[0]
        kt = self._tile_bufs(m["S_pad"], H, 1, grid, "sta_k")[0]
        ops.rmsnorm_rope([rows[:, :d], rows[:, d:2 * d]], [b["nq_w"], b["nk_w"]], cos, sin, head_dim=D, seq_len=S, eps=self.eps,
                         outs=[qg.view(-1, d), kt.view(-1, d)], row_maps=[m["group_untile"], m["k_row_of_token"]])
        ev = None
        if self.attn_events is not None:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        vt = ops.v_transpose(rows[:, 2 * d:3 * d].view(1, S, H, D), src_rows=m["v_src_rows"])
        o = ops.attn_tile_lists(qg, kt, None, m["group_q2k_idx"], m["group_q2k_num"], m["block_sizes"], 256, None, scale=D**-0.5,
                                layout="bshd", vt=vt, o_rows=m["group_token_of_row"], n_out_rows=S)
        if ev is not None:
            ev[1].record()
            self.attn_events.append((ev[0], ev[1], S, S, H))
        return o[0]

    # ------------------------------------------------------------------ sparse attention under sequence parallelism
    def _sp_plan(self, grid, Sl):
        """The uneven output-exchange plan of the tile-major sparse modes for this (grid, SP layout): integer host work, cached."""
        key = ("sp_plan", grid, Sl, self.attention)
        plan = self._vsa_cache.get(key)
        if plan is None:
            L = self.sp.lay
            if self.attention == "vsa":
                plan = self.sp.block_plan(self._vsa_meta(grid)["token_of_row"], Sl, 64)
            else:
                if L.U != 1:
                    raise NotImplementedError(f"sliding-tile attention under sequence parallelism needs num_heads % world == 0 (heads {self.H}, world "
                                              f"{L.P}: G{L.G} x U{L.U}); video-sparse and dense attention run on any G x U grid")
                plan = self.sp.block_plan(torch.arange(Sl * L.P), Sl, 1)  # U == 1: only the equal-split exchange is used
            plan.send_tokens, plan.asm_idx = plan.send_tokens.to(self.device), plan.asm_idx.to(self.device)
            self._vsa_cache[key] = plan
        return plan

    def _sp_sparse(self, rows, b, cos, sin, S, grid, pos0):
        """Video-sparse / sliding-tile self-attention of one sample under sequence parallelism (P > 1, any G x U grid for VSA).
        ONE kernel writes exchange #1's send buffer (QK-norm + RoPE + per-peer packing, the VSA compress gate as a fourth slot next to Q:
        fvk_qkv(g)_norm_rope_pack_bf16) — the reference sends q, k, v and the gate through all_to_all_4D with cat / transpose copies
        (attention/layer.py:172-245).  On the receive side the rank holds K, V, Q (and the gate) of ALL tokens for its head group, token-major;
        q and k are scattered into the tile-major layouts by row maps, V goes straight into the tile-major V^T (STA) / tile buffer (VSA),
        the kernels compute this rank's run of query blocks and write their output in token order, and the output exchange returns rows to
        their shard owners (equal split for U = 1, the uneven plan of SequenceParallel.block_plan for U > 1).  rows [Sl, 3d(+d)] -> [Sl, H*D]."""
        d, D, sp = self.d, self.D, self.sp
        L = sp.lay
        Sl = rows.shape[0]
        gate = rows[:, 3 * d:4 * d] if (self.attention == "vsa" and rows.shape[1] >= 4 * d) else None
        send = ops.qkv_norm_rope_pack(rows[:, :d], rows[:, d:2 * d], rows[:, 2 * d:3 * d], b["nq_w"], b["nk_w"], cos, sin, L.G, L.U, head_dim=D,
                                      seq_len=S, eps=self.eps, pos_offset=pos0, gate=gate)
        plan = self._sp_plan(grid, Sl)

        def vsa_fn(r4, plan):
            n, NS, hg, _ = r4.shape
            m = self._vsa_meta(grid)
            tq, tk, tv = self._tile_bufs(m["S_pad"], hg, 3, grid, "vsa")[:3]
            for slot, buf in ((2, tq), (0, tk), (1, tv)):
                ops.gather_rows(r4[:S, slot].unsqueeze(0), m["S_pad"], m["tile_partition_indices"], m["non_pad_index"], out=buf)
            vbs = m["variable_block_sizes"]
            b0, b1 = plan.r0 // 64, plan.r1 // 64
            g4 = r4[:, 3].unsqueeze(0) if NS == 4 else None
            want = self.vsa_trace is not None
            res = kernel_api._vsa_forward(tq[:, plan.r0:plan.r1], tk, tv, vbs, vbs[b0:b1], m["topk"], g4, "bshd", want, 64,
                                          token_of_row=m["token_of_row"][plan.r0:plan.r1], n_tokens=n)
            if want:
                res, inter = res
                self.vsa_trace.append(inter["mask"])
            return zero_pad_rows(res[0], n)

        def zero_pad_rows(o, n):
            # rows >= S are the shards' zero-padding tokens (S % P != 0): no tile-major row maps to them, so the scattering epilogues never
            # write them, and exchange #2 ships them to the last shard's owner, where they enter the o-projection — harmless for the
            # row-wise bf16 path (all_gather_unpad drops them), but a tensor-wise fp8 absmax over all rows would take its scale from
            # uninitialised memory.  The dense path's pad queries are finite by construction; make these finite (zero) too.
            if n > S:
                o[S:].zero_()
            return o

        def sta_fn(r4, plan):
            n, NS, hg, _ = r4.shape
            m = self._sta_meta(grid, hg)
            qg = self._tile_bufs(m["group_rows"], hg, 1, grid, "sta_q")[0]
            kt = self._tile_bufs(m["S_pad"], hg, 1, grid, "sta_k")[0]
            ops.gather_rows(r4[:S, 2].unsqueeze(0), m["group_rows"], m["group_src"], m["group_dst"], out=qg)
            ops.gather_rows(r4[:S, 0].unsqueeze(0), m["S_pad"], m["perm"], m["non_pad"], out=kt)
            vt = ops.v_transpose(r4[:S, 1].unsqueeze(0), src_rows=m["v_src_rows"])
            o = ops.attn_tile_lists(qg, kt, None, m["group_q2k_idx"], m["group_q2k_num"], m["block_sizes"], 256, None, scale=D**-0.5,
                                    layout="bshd", vt=vt, o_rows=m["group_token_of_row"], n_out_rows=n)
            return zero_pad_rows(o[0], n)

        fn = vsa_fn if self.attention == "vsa" else sta_fn
        if self.attn_events is not None:
            inner = fn

            def fn(r4, plan):  # bench.py roofline leg: HIP events around the whole local attention
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                o = inner(r4, plan)
                e1.record()
                self.attn_events.append((e0, e1, (plan.r1 - plan.r0) if self.attention == "vsa" else S, S, r4.shape[2]))
                return o
        return sp.attention_blocks(send, plan, fn, head_dim=D).reshape(Sl, self.H * D)

    def _lin(self, x, b, key, bias, **kw):
        """y = epilogue(x @ W^T + bias) through the bf16 GEMM or the fp8 path (dynamic activation quantisation + fp8 MFMA GEMM)."""
        if not self.quant:
            return ops.gemm(x, b[key + "_w"], bias, **kw)
        if isinstance(x, tuple):  # already quantised per token by the producing LayerNorm pass (ops.ln_modulate(fp8_rowwise=...))
            xq, xs = x
        else:
            xq, xs = ops.fp8_quantize(x, rowwise=(self.quant == "fp8_channel"))
        return ops.gemm_fp8(xq, xs, b[key + "_q"], b[key + "_s"], bias, **kw)

    # ------------------------------------------------------------------ forward
    @torch.no_grad()
    def forward(self, hidden_states, encoder_hidden_states, timestep, trace=None):
        dev, d, H, D = self.device, self.d, self.H, self.D
        if hidden_states.device.type != "cuda":
            raise RuntimeError("WanTransformer3DModelHip runs on a ROCm device only (no CPU fallback)")
        B, C, T, Hh, W = hidden_states.shape
        pt, ph, pw = self.patch
        grid = (T // pt, Hh // ph, W // pw)
        S = math.prod(grid)
        w = self.w
        sp = self.sp
        P, rank = sp.lay.P, sp.lay.rank
        cos, sin = rope.get_rotary_pos_embed(grid, D, device=dev)
        if self.attn_autotune and self.attention == "dense" and self._forwards >= 1:
            self._tune = []  # this forward times the two long-key attention kernels in place (see __init__)

        # patch embedding (Conv3d k=s=patch == GEMM over patch rows), then shard the token axis
        x = ops.gemm(ops.patchify(hidden_states.to(BF16), self.patch).view(B * S, -1), w["pe_w"], w["pe_b"]).view(B, S, d)
        x = sp.shard(x, dim=1)
        Sl = x.shape[1]
        pos0 = rank * Sl

        # modulation groups: row m of the local token block uses modulation row m // rpb (the kernels' rows_per_batch indexing).
        #   scalar timestep [B]      : one row per batch element, rpb = Sl
        #   per-token timesteps [B,S]: Wan2.2 TI2V (wanvideo.py:375-385, 690-712; built at denoising.py:441-446) and per-frame
        #                              (causal / diffusion-forcing) schedules.  The embedder MLPs are row-independent, so tokens that
        #                              share a timestep share a modulation row: one row per latent frame (rpb = tokens per frame) when
        #                              the timestep is constant inside every frame and the local shard is frame-aligned, else one
        #                              row per token (rpb = 1) — never the reference's [S, 6, d] fp32 tensor per layer unless needed.
        rpb = Sl
        if timestep.dim() == 2:
            if B != 1 or timestep.shape[1] != S:
                raise ValueError(f"per-token timestep must be [1, {S}], got {tuple(timestep.shape)}")
            ts = timestep.to(device=dev, dtype=torch.float32).reshape(-1)
            tpf = grid[1] * grid[2]
            tf = ts.view(grid[0], tpf)
            if Sl % tpf == 0 and bool((tf == tf[:, :1]).all()):
                rpb, ng = tpf, Sl // tpf
                tg = torch.zeros(P * ng, dtype=torch.float32, device=dev)
                tg[:grid[0]] = tf[:, 0]
                timestep = tg[rank * ng:(rank + 1) * ng]
            else:
                rpb = 1
                tg = torch.zeros(P * Sl, dtype=torch.float32, device=dev)
                tg[:S] = ts
                timestep = tg[pos0:pos0 + Sl]
        G = timestep.shape[0]

        # condition embedder (wanvideo.py:102-136)
        t_freq = ops.timestep_embedding(timestep.to(dev), self.freq_dim)
        h = ops.gemm(t_freq, w["time_embedder.mlp.fc_in.w"], w["time_embedder.mlp.fc_in.b"], epilogue=ops.EPI_SILU)
        temb = ops.gemm(h, w["time_embedder.mlp.fc_out.w"], w["time_embedder.mlp.fc_out.b"])
        tproj = ops.gemm(ops.silu(temb), w["time_modulation.linear.w"], w["time_modulation.linear.b"]).view(G, 6, d)
        ctx = encoder_hidden_states.to(device=dev, dtype=BF16)
        Lc = ctx.shape[1]
        c = ops.gemm(ctx.reshape(B * Lc, -1), w["text_embedder.fc_in.w"], w["text_embedder.fc_in.b"], epilogue=ops.EPI_GELU_TANH)
        c = ops.gemm(c, w["text_embedder.fc_out.w"], w["text_embedder.fc_out.b"])
        ckv = self._lin(c, {"ckv_w": getattr(self, "ckv_w", None), "ckv_q": getattr(self, "ckv_q", None), "ckv_s": getattr(self, "ckv_s", None)}, "ckv", self.ckv_b)  # [B*Lc, L*2d]: every layer's text K and V

        # AdaLN vectors e = table + temb.float() (wanvideo.py:386-390): [L,G,6,d] fp32 for all layers at once while that is small
        # (G = batch or latent frames), per layer when there is one row per token
        tproj32 = tproj.float()
        if len(self.blocks) * G * 6 * d * 4 <= (1 << 30):
            e_all = self.tables + tproj32.unsqueeze(0)
            parts = [e_all[:, :, j].contiguous() for j in range(6)]
            parts[1], parts[4] = 1 + parts[1], 1.0 + parts[4]
            mods = lambda i: tuple(p_[i] for p_ in parts)
        else:
            def mods(i):
                e_i = self.tables[i] + tproj32
                sh, sc, ga, csh, csc, cga = (e_i[:, j].contiguous() for j in range(6))
                return sh, 1 + sc, ga, csh, 1.0 + csc, cga

        x = x.reshape(B * Sl, d)
        for i, b in enumerate(self.blocks):
            shift_i, mul_i, gate_i, c_shift_i, mul_c_i, c_gate_i = mods(i)
            # fp8_channel: the per-token quantisation of a LayerNorm output that only feeds linears is written by the LayerNorm pass itself
            # (byte-identical to the stand-alone quantiser; no bf16 copy, no absmax + quantise passes)
            fq = "only" if self.quant == "fp8_channel" else None
            nh = ops.ln_modulate(x, mul=mul_i, add=shift_i, eps=self.eps, rows_per_batch=rpb, fp8_rowwise=fq if b["n_qkv"] != 4 else None)
            # single GPU, dense attention, bf16 linears (the contract step): the V projection is a GEMM of its own that writes V^T in the
            # attention kernels' layout (ops.gemm_vt: the token rows as the GEMM's w operand, key permutation in its staging addresses) —
            # bit-identical to the fused QKV GEMM + the V^T layout pass, without writing V, reading it back and a launch of that pass
            vt_i = None
            if (self.vt_gemm and P == 1 and self.attention == "dense" and not self.quant and b["n_qkv"] == 3
                    and ops.gemm_vt_eligible(nh.view(B, Sl, d), b["qkv_w"][2 * d:], b["qkv_b"][2 * d:])):
                qkv = ops.gemm(nh, b["qkv_w"][:2 * d], b["qkv_b"][:2 * d])       # [B*S, 2d]: q | k
                vt_i = ops.gemm_vt(nh.view(B, Sl, d), b["qkv_w"][2 * d:], b["qkv_b"][2 * d:])
            elif self.quant and b["n_qkv"] == 4:
                qkv = torch.empty((B * Sl, 4 * d), dtype=BF16, device=dev)
                self._lin(nh, b, "qkv", b["qkv_b"], out=qkv[:, :3 * d])
                ops.gemm(nh, b["gate_w"], b["gate_b"], out=qkv[:, 3 * d:])
            else:
                qkv = self._lin(nh, b, "qkv", b["qkv_b"])  # [B*Sl, 3d (+d gate)]; fp8: nh quantised once for q, k and v (wants_prequantized_input)
            nq = b["n_qkv"]
            batched = (B > 1 or vt_i is not None) and P == 1 and self.attention == "dense"
            attn = torch.empty((B * Sl, d), dtype=BF16, device=dev) if (B > 1 and not batched) else None
            if batched:
                # all batch elements (the classifier-free-guidance pair) in one QK-norm / RoPE pass, one V^T pass and ONE attention launch:
                # positions are row % S in the norm pass, the attention kernels take the batch as a grid dimension
                qn, kn = ops.rmsnorm_rope([qkv[:, :d], qkv[:, d:2 * d]], [b["nq_w"], b["nk_w"]], cos, sin, head_dim=D, seq_len=S, eps=self.eps)
                attn = self._dense_attn(qn.view(B, S, H, D), kn.view(B, S, H, D), None if vt_i is not None else qkv[:, 2 * d:3 * d].view(B, S, H, D),
                                        vt=vt_i).view(B * S, d)
            for bi in range(0 if batched else B):
                rows = qkv[bi * Sl:(bi + 1) * Sl]
                gate = rows[:, 3 * d:4 * d].view(Sl, H, D) if nq == 4 else None
                fn = lambda q_, k_, v_, kv_len, g_=None: self._attn_local(q_, k_, v_, kv_len, grid, g_)
                if P > 1 and self.attention in ("vsa", "sta"):
                    o = self._sp_sparse(rows, b, cos, sin, S, grid, pos0)
                elif P > 1 and gate is None:
                    # sequence parallel: QK-norm + RoPE + the per-peer packing of exchange #1 in ONE kernel (no torch.cat, no unpack —
                    # attention reads K, V and its query rows out of the received buffer through strides)
                    pack = lambda **kw: ops.qkv_norm_rope_pack(rows[:, :d], rows[:, d:2 * d], rows[:, 2 * d:3 * d], b["nq_w"], b["nk_w"], cos, sin,
                                                               sp.lay.G, sp.lay.U, head_dim=D, seq_len=S, eps=self.eps, pos_offset=pos0, **kw)
                    hg = sp.lay.heads_per_group
                    if sp.overlap and hg >= 2:
                        # FVK_SP_OVERLAP=1: two head chunks (the larger first), their exchanges asynchronous
                        # (distributed.py: attention_packed_pipelined); the first call is checked against the plain exchange
                        o = sp.attention_packed_pipelined(pack(heads_a=(hg + 1) // 2), S, fn, head_dim=D).reshape(Sl, d)
                        if not sp._overlap_checked:
                            ref = sp.attention_packed(pack(), S, fn, head_dim=D).reshape(Sl, d)
                            if not sp.pipelined_agrees(o, ref):
                                o = ref
                    else:
                        o = sp.attention_packed(pack(), S, fn, head_dim=D).reshape(Sl, d)
                elif P == 1 and self.attention == "sta" and self.sta_lists == "grouped" and self.sta_fold:
                    o = self._sta_fused(rows, b, cos, sin, S, grid).reshape(Sl, d)
                elif P == 1 and self.attention == "vsa" and self.vsa_fold:
                    o = self._vsa_fused(rows, b, cos, sin, S, grid).reshape(Sl, d)
                else:
                    q, k = ops.rmsnorm_rope([rows[:, :d], rows[:, d:2 * d]], [b["nq_w"], b["nk_w"]], cos, sin, head_dim=D, seq_len=S,
                                            eps=self.eps, pos_offset=pos0)
                    v = rows[:, 2 * d:3 * d]
                    o = sp.attention(q.view(Sl, H, D), k.view(Sl, H, D), v.view(Sl, H, D), S, fn, extra=gate).reshape(Sl, d)
                if B == 1:
                    attn = o
                else:
                    attn[bi * Sl:(bi + 1) * Sl] = o
            if trace is not None:   # the self-attention output in token order, before the out-projection (tests/test_gpu_bigseq.py)
                trace[f"blocks.{i}.attn"] = attn.view(B, Sl, d).clone()
            a_out = self._lin(attn, b, "o", b["o_b"])
            nh, x = ops.ln_modulate(a_out, residual=x, gate=gate_i, ln_w=b["ln2_w"], ln_b=b["ln2_b"], eps=self.eps,
                                    want_residual=True, rows_per_batch=rpb, fp8_rowwise=fq)
            if trace is not None:
                trace[f"blocks.{i}.after_self_attn"] = x.view(B, Sl, d).clone()
            # cross attention over the text tokens (WanT2VCrossAttention, wanvideo.py:188-222)
            cq = self._lin(nh, b, "cq", b["cq_b"])
            cq = ops.rmsnorm_rope([cq], [b["cnq_w"]], head_dim=D, seq_len=Sl, eps=self.eps)[0]
            kv = ckv[:, i * 2 * d:(i + 1) * 2 * d]
            ck = ops.rmsnorm_rope([kv[:, :d]], [b["cnk_w"]], head_dim=D, seq_len=Lc, eps=self.eps)[0]
            co = ops.attn_dense(cq.view(B, Sl, H, D), ck.view(B, Lc, H, D), kv[:, d:].view(B, Lc, H, D), scale=D**-0.5,
                                layout="bshd")
            # cross_attn_residual_norm is called with gate = 1 (wanvideo.py:425): `residual + x` is then a bf16 + bf16 sum, ROUNDED to bf16
            # before the norm (layernorm.py:193-199) — so the add rides in the out-projection's epilogue (bf16(x + y), the same one rounding)
            # and the norm pass reads one tensor instead of two and writes one instead of two (round 5; bit-identical)
            if self.fuse_cross_residual:
                x = self._lin(co.view(B * Sl, d), b, "co", b["co_b"], epilogue=ops.EPI_RESIDUAL_GATE, residual=x, gate=None, rows_per_batch=rpb)
                nh = ops.ln_modulate(x, mul=mul_c_i, add=c_shift_i, eps=self.eps, round_norm=True, rows_per_batch=rpb, fp8_rowwise=fq)
            else:
                c_out = self._lin(co.view(B * Sl, d), b, "co", b["co_b"])
                nh, x = ops.ln_modulate(c_out, residual=x, mul=mul_c_i, add=c_shift_i, eps=self.eps, round_residual=True,
                                        round_norm=True, want_residual=True, rows_per_batch=rpb, fp8_rowwise=fq)
            f = self._lin(nh, b, "f1", b["f1_b"], epilogue=ops.EPI_GELU_TANH)
            x = self._lin(f, b, "f2", b["f2_b"], epilogue=ops.EPI_RESIDUAL_GATE, residual=x, gate=c_gate_i, rows_per_batch=rpb)
            if trace is not None:
                trace[f"blocks.{i}.out"] = x.view(B, Sl, d).clone()

        # output norm (bf16 modulation vectors, wanvideo.py:746-756), gather, projection, unpatchify
        ss = self.out_table + temb.unsqueeze(1)           # [G,2,d] bf16
        shift, scale = ss[:, 0], ss[:, 1]
        x = ops.ln_modulate(x, mul=(1.0 + scale).float(), add=shift.float(), eps=self.eps, round_norm=True, rows_per_batch=rpb)
        x = sp.all_gather_unpad(x.view(B, Sl, d), S, dim=1)
        if trace is not None:
            trace["norm_out"] = x.clone()
        y = ops.gemm(x.reshape(B * S, d), w["proj_out.w"], w["proj_out.b"])
        c_out = y.shape[-1] // (pt * ph * pw)
        self._forwards += 1
        if self._tune is not None:
            self._finish_attn_tune()
        return ops.unpatchify(y.view(B, S, -1), (B, c_out, T, Hh, W), self.patch)

    __call__ = forward

    def capture(self, hidden_states, encoder_hidden_states, timestep, warmup: int = 2):
        """The forward for inputs of THESE shapes as HIP graphs (round 6; ``distributed.GraphSegments``): one graph at SP = 1, graph segments
        cut at the exchanges under sequence parallelism.  Returns ``replay(hidden_states, encoder_hidden_states, timestep) -> output``: the
        inputs are copied into the captured buffers, the graphs replayed, and the SAME output tensor returned every time (clone it to keep
        it).  Bit-identical to ``forward`` (the same kernels on the same data: tests/test_gpu_graph.py).  Every rank of an SP group must call
        it (the capture runs the forward: its collectives are real).  Not served: the pipelined exchange (FVK_SP_OVERLAP), per-launch
        HIP-event hooks (attn_events), the in-place kernel choice (attn_autotune must have finished or be off), ``trace``."""
        from .distributed import GraphSegments
        if self.sp.overlap:
            raise NotImplementedError("capture: the pipelined exchange (FVK_SP_OVERLAP) issues its collectives asynchronously; capture the plain exchange")
        if self.attn_events is not None or self._tune is not None or (self.attn_autotune and self.attention == "dense" and self.attn_tune_report is None):
            raise RuntimeError("capture: HIP-event hooks / the in-place attention kernel choice synchronise the stream; finish them first")
        static = [hidden_states.clone(), encoder_hidden_states.clone(), timestep.clone()]
        cur = torch.cuda.current_stream()
        side = torch.cuda.Stream()
        side.wait_stream(cur)
        seg = GraphSegments()
        with torch.cuda.stream(side):
            for _ in range(warmup):    # module loads, LDS attributes, metadata caches, allocator growth: none of it inside the capture
                self.forward(*static)
            torch.cuda.synchronize()
            self.sp.segmenter = seg
            try:
                seg.begin()
                out = self.forward(*static)
                seg.end()
            finally:
                self.sp.segmenter = None
        cur.wait_stream(side)
        torch.cuda.synchronize()

        def replay(hidden_states, encoder_hidden_states, timestep):
            for dst, src in zip(static, (hidden_states, encoder_hidden_states, timestep)):
                if dst.shape != src.shape:
                    raise ValueError(f"captured for {tuple(dst.shape)}, got {tuple(src.shape)}")
                dst.copy_(src)
            seg.replay()
            return out
        replay.segments = seg
        return replay
