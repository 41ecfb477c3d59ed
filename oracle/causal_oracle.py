"""TEST INFRASTRUCTURE — CPU restatement of the reference's causal (autoregressive, KV-cached) Wan DiT inference forward.

Follows ``CausalWanTransformer3DModel._forward_inference`` (fastvideo/models/dits/causal_wanvideo.py:545-654),
``CausalWanTransformerBlock.forward`` (:261-342) and the KV-cache branch of ``CausalWanSelfAttention.forward`` (:72-191) under
bf16 autocast, every rounding point explicit.  Differences from the bidirectional model (oracle/wan_oracle.py) that matter:
  * one modulation row per latent FRAME group: ``timestep`` is [B, F'] and ``temb`` [B, F', 6, d]; ``e = scale_shift_table + temb``
    is NOT promoted to fp32 (:281), so with bf16 parameters ``1 + scale``, ``x * gate`` and the residual sums round to bf16;
  * ``norm1`` and the two residual norms are plain ``nn.LayerNorm`` modules (:207, :239-251), i.e. they follow the AUTOCAST POLICY of the
    device: ``layer_norm`` is on CUDA/ROCm autocast's fp32 list but not on the CPU one.  ``ln_policy="cpu"`` (what this container's
    reference run does, and what the fixtures hold) keeps them in bf16 — output rounded to bf16, modulation in bf16;
    ``ln_policy="cuda"`` restates the GPU eager path (fp32 norm, fp32 modulation, rounded once by the next linear's autocast);
  * RoPE tables stay float64 (no ``.float()``; :589-598) with the temporal positions shifted by ``start_frame``
    (rotary_embedding.py:387-388); q and k are rotated separately and cast with ``type_as(v)`` (:100-101);
  * the roped keys and the values are written into a per-layer cache [B, cache_tokens, H, D]; when a local attention window is
    configured and the cache is full, the oldest non-sink tokens are evicted by a left shift (:145-160); attention runs over
    ``cache[max(0, end - max_attention_size):end]`` (:172-173, :181);
  * ``rope_cache_policy="relativistic"`` caches RAW keys and rotates query / key window with a position-0 table at attention time
    (:96-98, :174-180, _relative_rope.py:10-24);
  * the text context is zero-padded to ``text_len`` tokens before the text embedder (:607-612).
Pinned bit-exactly against the real reference on the same host (tests/test_causal_oracle.py, live when /root/reference exists) and
against tests/golden/wan_causal.pt (generator: oracle/make_golden_causal.py).  Only tests / smoke / bench may import this module."""
from __future__ import annotations

import torch
import torch.nn.functional as F

from oracle import wan_oracle as W

BF16 = torch.bfloat16
GLOBAL_ATTN_COMPAT_MAX_LATENT_FRAMES = 21  # causal_wanvideo.py:37


def rope_tables_f64(grid_thw, head_dim: int, start_frame: int = 0, theta: float = 10000.0):
    """``get_rotary_pos_embed(..., dtype=float64, start_frame=)`` — rotary_embedding.py:468-564, 349-450 (``full_grid[0] += start_frame``
    on the fp32 meshgrid, :387-388), 290-346.  Returns float64 [S, head_dim] tables (the causal model never casts them)."""
    dims = W.rope_dim_list(head_dim)
    axes = [torch.linspace(0, n, n + 1, dtype=torch.float32)[:n] for n in grid_thw]
    grid = torch.stack(torch.meshgrid(*axes, indexing="ij"), dim=0)
    if start_frame > 0:
        grid[0] += start_frame
    cos_l, sin_l = [], []
    for i, dim in enumerate(dims):
        pos = grid[i].reshape(-1)
        freqs = 1.0 / (theta**(torch.arange(0, dim, 2)[:(dim // 2)].to(torch.float64) / dim))
        fr = torch.outer(pos * 1.0, freqs)
        cos_l.append(fr.cos().repeat_interleave(2, dim=-1))
        sin_l.append(fr.sin().repeat_interleave(2, dim=-1))
    return torch.cat(cos_l, dim=1), torch.cat(sin_l, dim=1)


def cache_update_plan(local_attn_size: int, sink_size: int, frame_seqlen: int, cache_tokens: int, num_new: int, current_start: int,
                      global_end: int, local_end_prev: int):
    """Pure integer restatement of the cache bookkeeping of causal_wanvideo.py:123-173.
    Returns dict(evict=(src0, dst0, n) or None, write=(lo, hi), window=(lo, hi), global_end, local_end)."""
    current_end = current_start + num_new
    sink_tokens = sink_size * frame_seqlen
    max_att = (GLOBAL_ATTN_COMPAT_MAX_LATENT_FRAMES if local_attn_size == -1 else local_attn_size) * frame_seqlen
    if local_attn_size == -1 and current_end > max_att:
        raise ValueError("Causal Wan local_attn_size=-1 keeps the previous 21-latent-frame KV window")
    evict = None
    if local_attn_size != -1 and current_end > global_end and num_new + local_end_prev > cache_tokens:
        num_evicted = num_new + local_end_prev - cache_tokens
        num_rolled = local_end_prev - num_evicted - sink_tokens
        evict = (sink_tokens + num_evicted, sink_tokens, num_rolled)
        local_end = local_end_prev + current_end - global_end - num_evicted
    else:
        local_end = local_end_prev + current_end - global_end
    return dict(evict=evict, write=(local_end - num_new, local_end), window=(max(0, local_end - max_att), local_end),
                global_end=current_end, local_end=local_end, max_attention_size=max_att)


class CausalWanOracle(W.WanOracle):
    def __init__(self, sd: dict, num_heads: int, head_dim: int = 128, patch=(1, 2, 2), eps: float = 1e-6, freq_dim: int = 256,
                 local_attn_size: int = -1, sink_size: int = 0, text_len: int = 512, rope_cache_policy: str = "absolute",
                 ln_policy: str = "cpu"):
        super().__init__(sd, num_heads, head_dim, patch, eps, freq_dim)
        assert ln_policy in ("cpu", "cuda")
        self.ln_fp32 = ln_policy == "cuda"
        self.local_attn_size, self.sink_size, self.text_len = local_attn_size, sink_size, text_len
        self.rope_cache_policy = rope_cache_policy

    def init_kv_cache(self, batch: int, cache_tokens: int, dtype=BF16):
        """causal_denoising.py:358-386: zeros [B, cache_tokens, H, D] per layer, both end indices 0."""
        return [dict(k=torch.zeros(batch, cache_tokens, self.H, self.D, dtype=dtype), v=torch.zeros(batch, cache_tokens, self.H, self.D, dtype=dtype),
                     global_end_index=0, local_end_index=0) for _ in range(self.num_layers)]

    def ln_ac(self, x, weight=None, bias=None):
        """nn.LayerNorm under bf16 autocast (see the module docstring for the device policy)."""
        d = x.shape[-1]
        if self.ln_fp32:
            return F.layer_norm(x.float(), (d, ), None if weight is None else weight.float(), None if bias is None else bias.float(), self.eps)
        return F.layer_norm(x, (d, ), weight, bias, self.eps)

    @staticmethod
    def lin_ac(x, w, b=None):
        """F.linear under bf16 autocast: operands cast to bf16, bf16 result."""
        return F.linear(x.to(BF16), w.to(BF16), None if b is None else b.to(BF16))

    def _lin(self, x, prefix):
        return self.lin_ac(x, self.sd[prefix + ".weight"], self.sd.get(prefix + ".bias"))

    # -- CausalWanSelfAttention.forward, kv_cache branch (causal_wanvideo.py:72-191) --
    def causal_self_attn(self, q, k, v, cos, sin, kv, current_start: int, frame_seqlen: int):
        relativistic = self.rope_cache_policy == "relativistic"
        if not relativistic:
            roped_q = W.apply_rotary_emb(q, cos, sin).type_as(v)
            roped_k = W.apply_rotary_emb(k, cos, sin).type_as(v)
        plan = cache_update_plan(self.local_attn_size, self.sink_size, frame_seqlen, kv["k"].shape[1], q.shape[1], current_start,
                                 int(kv["global_end_index"]), int(kv["local_end_index"]))
        stored_key = k if relativistic else roped_k
        if plan["evict"] is not None:
            src, dst, n = plan["evict"]
            kv["k"][:, dst:dst + n] = kv["k"][:, src:src + n].clone()
            kv["v"][:, dst:dst + n] = kv["v"][:, src:src + n].clone()
        lo, hi = plan["write"]
        kv["k"][:, lo:hi] = stored_key
        kv["v"][:, lo:hi] = v
        w0, w1 = plan["window"]
        key_window, value_window = kv["k"][:, w0:w1], kv["v"][:, w0:w1]
        if relativistic:
            window_len = min(plan["local_end"], plan["max_attention_size"])  # _relative_rope.py:23-24
            q_lo, q_hi = window_len - q.shape[1], window_len
            roped_q = W.apply_rotary_emb(q, cos[q_lo:q_hi], sin[q_lo:q_hi]).type_as(v)
            key_window = W.apply_rotary_emb(key_window, cos[:window_len], sin[:window_len]).type_as(v)
        o = W.sdpa_bshd(roped_q, key_window, value_window, self.D**-0.5)
        kv["global_end_index"], kv["local_end_index"] = plan["global_end"], plan["local_end"]
        return o

    # -- CausalWanTransformerBlock.forward (causal_wanvideo.py:261-342) --
    def causal_block(self, i, x, ctx, temb, cos, sin, kv, current_start, frame_seqlen, trace=None):
        p = f"blocks.{i}"
        odt = x.dtype
        B, S, d = x.shape
        Fp = temb.shape[1]
        tpt = S // Fp
        e = self.w(p + ".scale_shift_table") + temb  # [B, F', 6, d]; NOT promoted to fp32 (:281)
        shift_msa, scale_msa, gate_msa, c_shift, c_scale, c_gate = e.chunk(6, dim=2)
        n1 = self.ln_ac(x)
        nh = (n1.unflatten(1, (Fp, tpt)) * (1 + scale_msa) + shift_msa).flatten(1, 2)
        q = W.rms_norm(self._lin(nh, p + ".to_q"), self.w(p + ".norm_q.weight"), self.eps)
        k = W.rms_norm(self._lin(nh, p + ".to_k"), self.w(p + ".norm_k.weight"), self.eps)
        v = self._lin(nh, p + ".to_v")
        q, k, v = (t.unflatten(2, (self.H, -1)) for t in (q, k, v))
        a = self.causal_self_attn(q, k, v, cos, sin, kv, current_start, frame_seqlen).flatten(2)
        a = self._lin(a, p + ".to_out")
        # self_attn_residual_norm: ScaleResidualLayerNormScaleShift with a plain nn.LayerNorm (affine), 4-D gate (layernorm.py:173-207)
        res = x + (a.unflatten(1, (Fp, tpt)) * gate_msa).flatten(1, 2)
        ln = self.ln_ac(res, self.w(p + ".self_attn_residual_norm.norm.weight"), self.w(p + ".self_attn_residual_norm.norm.bias"))
        null = torch.tensor([0])
        nh = (ln * (1.0 + null) + null).to(odt)
        x = res.to(odt)
        if trace is not None:
            trace[f"{p}.after_self_attn"] = x.clone()
        a = self.cross_attn_ac(nh, ctx, p + ".attn2")
        res = x + a
        ln = self.ln_ac(res)
        nh = (ln.unflatten(1, (Fp, tpt)) * (1.0 + c_scale) + c_shift).flatten(1, 2)  # not cast: the FFN linear's autocast rounds it
        x = res
        f = self._lin(nh, p + ".ffn.fc_in")
        f = self._lin(W.gelu_tanh(f), p + ".ffn.fc_out")
        x = x + (f.unflatten(1, (Fp, tpt)) * c_gate).flatten(1, 2)
        if trace is not None:
            trace[f"{p}.out"] = x.clone()
        return x

    def cross_attn_ac(self, x, ctx, p):
        """``WanT2VCrossAttention.forward`` (wanvideo.py:188-222) under autocast."""
        B = x.shape[0]
        q = W.rms_norm(self._lin(x, p + ".to_q"), self.w(p + ".norm_q.weight"), self.eps).view(B, -1, self.H, self.D)
        k = W.rms_norm(self._lin(ctx, p + ".to_k"), self.w(p + ".norm_k.weight"), self.eps).view(B, -1, self.H, self.D)
        v = self._lin(ctx, p + ".to_v").view(B, -1, self.H, self.D)
        o = W.sdpa_bshd(q, k, v, self.D**-0.5)
        return self._lin(o.flatten(2), p + ".to_out")

    # -- CausalWanTransformer3DModel._forward_inference (causal_wanvideo.py:545-654) --
    def forward_inference(self, latent, ctx, timestep, kv_cache, current_start: int = 0, start_frame: int = 0, trace=None):
        B, C, T, Hh, Wd = latent.shape
        pt, ph, pw = self.patch
        grid = (T // pt, Hh // ph, Wd // pw)
        if self.rope_cache_policy == "relativistic":
            max_frames = GLOBAL_ATTN_COMPAT_MAX_LATENT_FRAMES if self.local_attn_size == -1 else self.local_attn_size
            cos, sin = rope_tables_f64((max_frames, grid[1], grid[2]), self.D, 0)
        else:
            cos, sin = rope_tables_f64(grid, self.D, start_frame)
        x = F.conv3d(latent, self.w("patch_embedding.proj.weight"), self.w("patch_embedding.proj.bias"), stride=self.patch)
        x = x.flatten(2).transpose(1, 2)
        ctx = torch.cat([ctx, ctx.new_zeros(1, self.text_len - ctx.size(1), ctx.size(2))], dim=1)
        # condition embedder on timestep.flatten() (visual_embedding.py:105-181; wanvideo.py:100-136)
        wdt = self.w("condition_embedder.time_embedder.mlp.fc_in.weight").dtype
        t_freq = W.timestep_embedding(timestep.flatten(), self.freq_dim).to(wdt)
        h = self._lin(t_freq, "condition_embedder.time_embedder.mlp.fc_in")
        temb = self._lin(F.silu(h), "condition_embedder.time_embedder.mlp.fc_out")
        tproj = self._lin(F.silu(temb), "condition_embedder.time_modulation.linear")
        c = self._lin(ctx, "condition_embedder.text_embedder.fc_in")
        c = self._lin(W.gelu_tanh(c), "condition_embedder.text_embedder.fc_out")
        tproj = tproj.unflatten(1, (6, self.d)).unflatten(0, tuple(timestep.shape))  # [B, F', 6, d]
        frame_seqlen = grid[1] * grid[2]
        for i in range(self.num_layers):
            x = self.causal_block(i, x, c, tproj, cos, sin, kv_cache[i], current_start, frame_seqlen, trace)
        temb4 = temb.unflatten(0, tuple(timestep.shape)).unsqueeze(2)  # [B, F', 1, d]
        shift, scale = (self.w("scale_shift_table").unsqueeze(1) + temb4).chunk(2, dim=2)
        Fp = timestep.shape[1]
        # norm_out: LayerNormScaleShift WITHOUT compute_dtype (causal_wanvideo.py:399-403) = plain nn.LayerNorm under the autocast policy,
        # 4-D scale/shift branch (layernorm.py:259-264), no trailing cast (the proj_out linear's autocast rounds)
        normalized = self.ln_ac(x)
        x = (normalized.unflatten(1, (Fp, x.shape[1] // Fp)) * (1.0 + scale) + shift).flatten(1, 2)
        if trace is not None:
            trace["norm_out"] = x.clone()
        x = self._lin(x, "proj_out")
        return W.unpatchify(x, grid, self.patch, x.shape[-1] // (pt * ph * pw))


def causal_dmd_rollout(oracle: CausalWanOracle, latents, ctx, dmd_steps, noise_list, num_frames_per_block: int, cache_frames: int,
                       shift: float = 8.0, context_noise: int = 0):
    """The block loop of ``CausalDMDDenosingStage.forward`` (causal_denoising.py:205-349, T2V single-expert path, no warp) on the oracle
    model + oracle/dmd_oracle.py.  ``noise_list``: the re-noising draws in call order, each [1, nfb, C, H, W] bf16."""
    from oracle import dmd_oracle as D
    ts_tab, sg_tab = D.tables(shift)
    latents = latents.clone()
    B, C, T, Hh, Wd = latents.shape
    fs = (Hh // oracle.patch[1]) * (Wd // oracle.patch[2])
    kv = oracle.init_kv_cache(1, cache_frames * fs)
    timesteps = torch.tensor(list(dmd_steps), dtype=torch.long)
    noise_it = iter(noise_list)
    start = 0
    for _ in range(T // num_frames_per_block):
        n = num_frames_per_block
        cur = latents[:, :, start:start + n]
        noise_btchw = cur.permute(0, 2, 1, 3, 4)
        for i, t_cur in enumerate(timesteps):
            noise_latents = noise_btchw.clone()
            t_expand = t_cur.repeat(1)
            pred = oracle.forward_inference(cur.to(BF16), ctx, t_cur * torch.ones((1, 1), dtype=torch.long), kv, current_start=start * fs,
                                            start_frame=start).permute(0, 2, 1, 3, 4)
            video = D.pred_noise_to_pred_video(pred.flatten(0, 1), noise_latents.flatten(0, 1), t_expand, ts_tab, sg_tab).unflatten(0, pred.shape[:2])
            if i < len(timesteps) - 1:
                nxt_t = timesteps[i + 1] * torch.ones([1], dtype=torch.long)
                noise = next(noise_it)
                noise_btchw = D.add_noise(video.flatten(0, 1), noise.flatten(0, 1), nxt_t, ts_tab, sg_tab).unflatten(0, video.shape[:2])
                cur = noise_btchw.permute(0, 2, 1, 3, 4)
            else:
                cur = video.permute(0, 2, 1, 3, 4)
        latents[:, :, start:start + n] = cur
        oracle.forward_inference(cur.to(BF16), ctx, torch.ones((1, 1), dtype=torch.long) * int(context_noise), kv, current_start=start * fs,
                                 start_frame=start)
        start += n
    return latents
