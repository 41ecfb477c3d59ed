"""Host of the causal (autoregressive, KV-cached) Wan DiT inference forward on MI355X: the reference's
``CausalWanTransformer3DModel._forward_inference`` (fastvideo/models/dits/causal_wanvideo.py:545-654),
``CausalWanTransformerBlock.forward`` (``:261-342``) and the KV-cache branch of ``CausalWanSelfAttention.forward`` (``:72-191``), driven
block by block as ``CausalDMDDenosingStage`` does (fastvideo/pipelines/stages/causal_denoising.py:205-349).

Same weights, kernels and kernel chain as ``WanTransformer3DModelHip`` (wan_dit.py); what differs per block:
  * one modulation row per latent-frame group (``timestep`` [1, F'], the kernels' ``rows_per_batch`` indexing — no [S, 6, d] tensors);
  * 3-D RoPE positions start at ``start_frame`` (rotary_embedding.py:387-388);
  * the roped keys and the values of the block go into a per-layer cache [1, cache_tokens, H, D] (resident in HBM for the whole rollout:
    21 frames x 1560 tokens x 1536 x 2 B x 2 = 201 MB per layer at 480p) and the dense attention kernel runs the block's queries against
    the cache window ``[max(0, end - max_attention_size), end)`` — Sq != Skv, the same kernel entry as cross-attention;
  * cache bookkeeping (growth / rewrite of the same positions / eviction behind the sink frames by a left shift) is integer host logic,
    bit-identical to ``causal_wanvideo.py:123-173`` (oracle/causal_oracle.py: cache_update_plan is the checker of this module's own copy);
  * ``rope_cache_policy="relativistic"``: keys are cached normed but un-roped and the window is rotated at attention time with a
    position-0 table (``:96-98, :174-180``);
  * text context zero-padded to ``text_len`` (``:607-612``); text K/V per layer can be kept across calls (``crossattn_cache``).
Numerics follow the reference's GPU eager path (CUDA/ROCm autocast keeps ``layer_norm`` in fp32; the CPU run of the reference rounds the
three plain ``nn.LayerNorm`` outputs to bf16 — oracle/causal_oracle.py ``ln_policy``).  RoPE runs in fp32 on fp32 tables (the causal
reference keeps float64 tables); the modulation products the reference rounds to bf16 twice are rounded once here.
SP = 1 only: the reference's causal attention is a ``LocalAttention`` (no sequence-parallel exchange, ``:65-70``)."""
from __future__ import annotations

import math

import torch

from . import ops, rope
from .wan_dit import BF16, WanTransformer3DModelHip

GLOBAL_ATTN_COMPAT_MAX_LATENT_FRAMES = 21  # causal_wanvideo.py:37


def cache_update_plan(local_attn_size: int, sink_size: int, frame_seqlen: int, cache_tokens: int, num_new: int, current_start: int,
                      global_end: int, local_end_prev: int) -> dict:
    """Integer bookkeeping of one cache update (causal_wanvideo.py:123-173): what to shift, where to write, which window to attend."""
    current_end = current_start + num_new
    sink_tokens = sink_size * frame_seqlen
    max_att = (GLOBAL_ATTN_COMPAT_MAX_LATENT_FRAMES if local_attn_size == -1 else local_attn_size) * frame_seqlen
    if local_attn_size == -1 and current_end > max_att:
        raise ValueError("Causal Wan local_attn_size=-1 keeps the previous "
                         f"{GLOBAL_ATTN_COMPAT_MAX_LATENT_FRAMES}-latent-frame KV window for compatibility. Set local_attn_size for "
                         f"longer rollouts; got current_end={current_end} tokens with frame_seqlen={frame_seqlen}.")
    evict = None
    if local_attn_size != -1 and current_end > global_end and num_new + local_end_prev > cache_tokens:
        num_evicted = num_new + local_end_prev - cache_tokens
        num_rolled = local_end_prev - num_evicted - sink_tokens
        evict = (sink_tokens + num_evicted, sink_tokens, num_rolled)  # (src, dst, count) in tokens
        local_end = local_end_prev + current_end - global_end - num_evicted
    else:
        local_end = local_end_prev + current_end - global_end
    if local_end > cache_tokens or local_end - num_new < 0:
        raise ValueError(f"KV cache of {cache_tokens} tokens cannot take tokens [{local_end - num_new}, {local_end})")
    return dict(evict=evict, write=(local_end - num_new, local_end), window=(max(0, local_end - max_att), local_end),
                global_end=current_end, local_end=local_end, max_attention_size=max_att)


class CausalWanTransformer3DModelHip(WanTransformer3DModelHip):

    def __init__(self, state_dict: dict, num_heads: int, head_dim: int = 128, patch_size=(1, 2, 2), eps: float = 1e-6,
                 freq_dim: int = 256, local_attn_size: int = -1, sink_size: int = 0, text_len: int = 512,
                 rope_cache_policy: str = "absolute", num_frames_per_block: int = 3, device="cuda"):
        super().__init__(state_dict, num_heads, head_dim, patch_size, eps, freq_dim, attention="dense", device=device)
        if self.sp.lay.P != 1:
            raise NotImplementedError("the causal Wan path runs at SP = 1 (the reference's causal attention has no SP exchange)")
        if rope_cache_policy not in ("absolute", "relativistic"):
            raise ValueError(f"unknown rope_cache_policy {rope_cache_policy!r}")
        assert num_frames_per_block <= 3  # causal_wanvideo.py:415
        self.local_attn_size, self.sink_size, self.text_len = local_attn_size, sink_size, text_len
        self.rope_cache_policy, self.num_frames_per_block = rope_cache_policy, num_frames_per_block

    # ------------------------------------------------------------------ caches (causal_denoising.py:358-408)
    def init_kv_cache(self, cache_tokens: int, batch: int = 1) -> list[dict]:
        z = lambda: torch.zeros((batch, cache_tokens, self.H, self.D), dtype=BF16, device=self.device)
        return [dict(k=z(), v=z(), global_end_index=0, local_end_index=0) for _ in range(self.num_layers)]

    def init_crossattn_cache(self) -> list[dict]:
        return [dict(is_init=False, k=None, v=None) for _ in range(self.num_layers)]

    def _self_attention(self, qkv, b, kv, cos, sin, current_start, frame_seqlen):
        """qkv [S, 3d] fused projection of the block's tokens -> attention output [S, d]; updates the layer's cache in place."""
        d, H, D = self.d, self.H, self.D
        S = qkv.shape[0]
        relativistic = self.rope_cache_policy == "relativistic"
        if relativistic:
            q, k = ops.rmsnorm_rope([qkv[:, :d], qkv[:, d:2 * d]], [b["nq_w"], b["nk_w"]], None, None, head_dim=D, seq_len=S, eps=self.eps)
        else:
            q, k = ops.rmsnorm_rope([qkv[:, :d], qkv[:, d:2 * d]], [b["nq_w"], b["nk_w"]], cos, sin, head_dim=D, seq_len=S, eps=self.eps)
        plan = cache_update_plan(self.local_attn_size, self.sink_size, frame_seqlen, kv["k"].shape[1], S, current_start,
                                 int(kv["global_end_index"]), int(kv["local_end_index"]))
        if plan["evict"] is not None:
            src, dst, n = plan["evict"]
            for t in (kv["k"], kv["v"]):
                t[:, dst:dst + n] = t[:, src:src + n].clone()
        lo, hi = plan["write"]
        kv["k"][0, lo:hi] = k.view(S, H, D)
        kv["v"][0, lo:hi] = qkv[:, 2 * d:3 * d].view(S, H, D)
        w0, w1 = plan["window"]
        key_window, value_window = kv["k"][:, w0:w1], kv["v"][:, w0:w1]
        q4 = q.view(1, S, H, D)
        if relativistic:
            window_len = min(plan["local_end"], plan["max_attention_size"])  # _relative_rope.py:23-24
            q_lo = window_len - S
            q4 = ops.rmsnorm_rope([q], None, cos[q_lo:window_len], sin[q_lo:window_len], head_dim=D, seq_len=S)[0].view(1, S, H, D)
            key_window = ops.rmsnorm_rope([key_window.reshape(window_len, d)], None, cos[:window_len], sin[:window_len], head_dim=D,
                                          seq_len=window_len)[0].view(1, window_len, H, D)
        vt = ops.v_transpose(value_window)
        if self.attn_events is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        o = ops.attn_dense(q4, key_window, vt=vt, scale=D**-0.5, layout="bshd")
        if self.attn_events is not None:
            e1.record()
            self.attn_events.append((e0, e1, S, w1 - w0, H))
        kv["global_end_index"], kv["local_end_index"] = plan["global_end"], plan["local_end"]
        return o.view(S, d)

    # ------------------------------------------------------------------ forward
    @torch.no_grad()
    def forward_inference(self, hidden_states, encoder_hidden_states, timestep, kv_cache, crossattn_cache=None, current_start: int = 0,
                          start_frame: int = 0, trace=None):
        """hidden_states [1, C, F, H, W] (one block of latent frames), encoder_hidden_states [1, L <= text_len, text_dim],
        timestep [1, F'] (F' divides F: one modulation row per F/F' frames) -> [1, C_out, F, H, W]; kv_cache from ``init_kv_cache``."""
        dev, d, H, D = self.device, self.d, self.H, self.D
        if hidden_states.device.type != "cuda":
            raise RuntimeError("CausalWanTransformer3DModelHip runs on a ROCm device only (no CPU fallback)")
        B, C, T, Hh, Wd = hidden_states.shape
        if B != 1:
            raise ValueError("the causal path takes one sample per call (the reference pads the text with a batch-1 tensor, :607-612)")
        pt, ph, pw = self.patch
        grid = (T // pt, Hh // ph, Wd // pw)
        S = math.prod(grid)
        frame_seqlen = grid[1] * grid[2]
        w = self.w
        timestep = timestep.to(dev).reshape(1, -1)
        Fp = timestep.shape[1]
        if S % Fp:
            raise ValueError(f"{S} tokens do not split into {Fp} timestep groups")
        rpb = S // Fp
        if self.rope_cache_policy == "relativistic":
            max_frames = GLOBAL_ATTN_COMPAT_MAX_LATENT_FRAMES if self.local_attn_size == -1 else self.local_attn_size
            cos, sin = rope.get_rotary_pos_embed((max_frames, grid[1], grid[2]), D, device=dev)
        else:
            cos, sin = rope.get_rotary_pos_embed(grid, D, device=dev, start_frame=start_frame)

        x = ops.gemm(ops.patchify(hidden_states.to(device=dev, dtype=BF16), self.patch).view(S, -1), w["pe_w"], w["pe_b"])
        # condition embedder on timestep.flatten() (wanvideo.py:100-136); text padded to text_len with zero rows
        t_freq = ops.timestep_embedding(timestep.reshape(-1).float(), self.freq_dim)
        h = ops.gemm(t_freq, w["time_embedder.mlp.fc_in.w"], w["time_embedder.mlp.fc_in.b"], epilogue=ops.EPI_SILU)
        temb = ops.gemm(h, w["time_embedder.mlp.fc_out.w"], w["time_embedder.mlp.fc_out.b"])            # [F', d] bf16
        tproj = ops.gemm(ops.silu(temb), w["time_modulation.linear.w"], w["time_modulation.linear.b"]).view(Fp, 6, d)
        need_text = crossattn_cache is None or not all(c["is_init"] for c in crossattn_cache)
        ckv = None
        Lc = self.text_len
        if need_text:
            ctx = encoder_hidden_states.to(device=dev, dtype=BF16)
            if ctx.shape[0] != 1 or ctx.shape[1] > self.text_len:
                raise ValueError(f"text context must be [1, <= {self.text_len}, text_dim], got {tuple(ctx.shape)}")
            ctxp = torch.zeros((self.text_len, ctx.shape[2]), dtype=BF16, device=dev)
            ctxp[:ctx.shape[1]] = ctx[0]
            c = ops.gemm(ctxp, w["text_embedder.fc_in.w"], w["text_embedder.fc_in.b"], epilogue=ops.EPI_GELU_TANH)
            c = ops.gemm(c, w["text_embedder.fc_out.w"], w["text_embedder.fc_out.b"])
            ckv = ops.gemm(c, self.ckv_w, self.ckv_b)  # [text_len, L*2d]: every layer's text K and V

        # AdaLN vectors: e = scale_shift_table + temb in the PARAMETER dtype (causal_wanvideo.py:281: no fp32 promotion), (1 + scale) in
        # that dtype as the reference computes it, then fp32 rows for the kernels
        e_all = self.tables.view(self.num_layers, 1, 6, d) + tproj.unsqueeze(0)   # [L, F', 6, d] bf16
        part = lambda j: e_all[:, :, j].float().contiguous()
        shift_a, gate_a, cshift_a, cgate_a = part(0), part(2), part(3), part(5)
        mul_a, cmul_a = (1 + e_all[:, :, 1]).float().contiguous(), (1.0 + e_all[:, :, 4]).float().contiguous()

        for i, b in enumerate(self.blocks):
            nh = ops.ln_modulate(x, mul=mul_a[i], add=shift_a[i], eps=self.eps, rows_per_batch=rpb)
            qkv = ops.gemm(nh, b["qkv_w"], b["qkv_b"])
            a = self._self_attention(qkv, b, kv_cache[i], cos, sin, current_start, frame_seqlen)
            a_out = ops.gemm(a, b["o_w"], b["o_b"])
            # residual sums are bf16 + bf16 in this block (bf16 gate, causal_wanvideo.py:281): the LayerNorm sees the ROUNDED residual
            nh, x = ops.ln_modulate(a_out, residual=x, gate=gate_a[i], ln_w=b["ln2_w"], ln_b=b["ln2_b"], eps=self.eps,
                                    round_residual=True, want_residual=True, rows_per_batch=rpb)
            if trace is not None:
                trace[f"blocks.{i}.after_self_attn"] = x.view(1, S, d).clone()
            cq = ops.gemm(nh, b["cq_w"], b["cq_b"])
            cq = ops.rmsnorm_rope([cq], [b["cnq_w"]], head_dim=D, seq_len=S, eps=self.eps)[0]
            cc = None if crossattn_cache is None else crossattn_cache[i]
            if cc is not None and cc["is_init"]:
                ck, cvt = cc["k"], cc["v"]
            else:
                kv_i = ckv[:, i * 2 * d:(i + 1) * 2 * d]
                ck = ops.rmsnorm_rope([kv_i[:, :d]], [b["cnk_w"]], head_dim=D, seq_len=Lc, eps=self.eps)[0].view(1, Lc, H, D)
                cvt = ops.v_transpose(kv_i[:, d:].view(1, Lc, H, D))
                if cc is not None:  # wanvideo.py:202-214: text K / V computed once per prompt ("v" holds the MFMA-ready V^T here)
                    cc.update(is_init=True, k=ck, v=cvt)
            co = ops.attn_dense(cq.view(1, S, H, D), ck, vt=cvt, scale=D**-0.5, layout="bshd")
            c_out = ops.gemm(co.view(S, d), b["co_w"], b["co_b"])
            nh, x = ops.ln_modulate(c_out, residual=x, mul=cmul_a[i], add=cshift_a[i], eps=self.eps, round_residual=True,
                                    want_residual=True, rows_per_batch=rpb)
            f = ops.gemm(nh, b["f1_w"], b["f1_b"], epilogue=ops.EPI_GELU_TANH)
            x = ops.gemm(f, b["f2_w"], b["f2_b"], epilogue=ops.EPI_RESIDUAL_GATE, residual=x, gate=cgate_a[i], rows_per_batch=rpb)
            if trace is not None:
                trace[f"blocks.{i}.out"] = x.view(1, S, d).clone()

        # output norm: plain LayerNorm (fp32 on the GPU) modulated by bf16 (1 + scale), shift of table + temb (causal_wanvideo.py:648-651)
        ss = self.out_table.view(1, 2, d) + temb.unsqueeze(1)                    # [F', 2, d] bf16
        x = ops.ln_modulate(x, mul=(1.0 + ss[:, 1]).float(), add=ss[:, 0].float(), eps=self.eps, rows_per_batch=rpb)
        if trace is not None:
            trace["norm_out"] = x.view(1, S, d).clone()
        y = ops.gemm(x, w["proj_out.w"], w["proj_out.b"])
        c_out = y.shape[-1] // (pt * ph * pw)
        return ops.unpatchify(y.view(1, S, -1), (1, c_out, T, Hh, Wd), self.patch)

    def forward(self, hidden_states, encoder_hidden_states, timestep, kv_cache=None, **kw):
        if kv_cache is None:
            raise NotImplementedError("the teacher-forcing / training forward (flex-attention block masks) is outside this inference path")
        return self.forward_inference(hidden_states, encoder_hidden_states, timestep, kv_cache, **kw)

    __call__ = forward


class CausalDenoisingLoopHip:
    """Block-by-block DMD rollout of ``CausalDMDDenosingStage.forward`` (fastvideo/pipelines/stages/causal_denoising.py:62-349, the T2V
    single-expert path): for every block of ``num_frames_per_block`` latent frames, the DMD steps (model forward against the KV cache ->
    ``pred_noise_to_pred_video`` -> re-noise to the next step with fresh noise), then one forward at ``context_noise`` over the clean block
    to rewrite its K / V in the cache.  Step tails are one HIP kernel each (scheduler.DmdStepper, bit-identical to the eager ops); the
    latent never leaves the GPU.  ``noise_fn(shape_btchw, dtype) -> tensor`` supplies the re-noising draws (the reference draws
    ``torch.randn(shape, dtype=pred_video.dtype, generator=batch.generator)`` and moves it to the device, :296-300)."""

    def __init__(self, transformer: CausalWanTransformer3DModelHip, dmd_denoising_steps, flow_shift: float = 8.0, warp_denoising_step: bool = False,
                 context_noise: int = 0, sliding_window_num_frames: int = 21):
        from .scheduler import DmdStepper
        self.model = transformer
        self.stepper = DmdStepper(flow_shift)
        steps = torch.tensor(list(dmd_denoising_steps), dtype=torch.long)
        self.timesteps = self.stepper.tables.warp(steps) if warp_denoising_step else steps  # causal_denoising.py:79-84
        self.context_noise = int(context_noise)
        self.sliding_window_num_frames = sliding_window_num_frames

    def cache_tokens(self, frame_seqlen: int) -> int:
        """causal_denoising.py:365-368."""
        m = self.model
        return (m.local_attn_size if m.local_attn_size != -1 else self.sliding_window_num_frames) * frame_seqlen

    @torch.no_grad()
    def run(self, latents, prompt_embeds, noise_fn):
        """latents [1, C, T, H, W] (noise, fp32 or bf16) -> denoised latents, same shape and dtype."""
        m = self.model
        dev = m.device
        latents = latents.to(dev).clone()
        B, C, T, Hh, Wd = latents.shape
        nfb = m.num_frames_per_block
        if T % nfb:
            raise ValueError("num_frames must be divisible by num_frames_per_block for causal DMD denoising")
        fs = (Hh // m.patch[1]) * (Wd // m.patch[2])
        kv = m.init_kv_cache(self.cache_tokens(fs))
        cc = m.init_crossattn_cache()
        prompt_embeds = prompt_embeds.to(dev)
        start = 0
        nsteps = len(self.timesteps)
        for _ in range(T // nfb):
            cur = latents[:, :, start:start + nfb]                       # [1, C, nfb, H, W]
            noisy = cur.permute(0, 2, 1, 3, 4).flatten(0, 1).contiguous()  # frames first [nfb, C, H, W]
            for i in range(nsteps):
                t_cur = self.timesteps[i]
                x_in = noisy.view(1, nfb, C, Hh, Wd).permute(0, 2, 1, 3, 4).to(BF16)
                pred = m.forward_inference(x_in, prompt_embeds, t_cur.reshape(1, 1), kv, cc, current_start=start * fs, start_frame=start)
                pred_f = pred.permute(0, 2, 1, 3, 4).flatten(0, 1).contiguous()
                if i < nsteps - 1:
                    noise = noise_fn((1, nfb, C, Hh, Wd), BF16).to(dev).flatten(0, 1)
                    _, noisy = self.stepper.step(pred_f, noisy, t_cur, noise, self.timesteps[i + 1])
                else:
                    noisy, _ = self.stepper.step(pred_f, noisy, t_cur)
            clean = noisy.view(1, nfb, C, Hh, Wd).permute(0, 2, 1, 3, 4)
            latents[:, :, start:start + nfb] = clean
            # context re-run: rewrite this block's K / V from the clean latent (causal_denoising.py:318-349)
            t_ctx = torch.full((1, 1), self.context_noise, dtype=torch.long)
            m.forward_inference(clean.to(BF16), prompt_embeds, t_ctx, kv, cc, current_start=start * fs, start_frame=start)
            start += nfb
        return latents
