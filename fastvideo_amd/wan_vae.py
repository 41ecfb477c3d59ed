"""Host of the Wan causal-3D-conv VAE *decode* on MI355X: the reference's
``AutoencoderKLWan.decode`` (fastvideo/models/vaes/wanvae.py:1189-1215) -> ``WanDecoder3d.forward`` (``:955-993``)
re-expressed over three HIP kernels of libfvk_amd.so:

    fvk_vae_conv_bf16          every WanCausalConv3d / Conv2d (implicit GEMM on MFMA; causal history, zero padding, the 2x
                               nearest upsample, the residual add and the final clamp/layout are folded into it)
    fvk_vae_rmsnorm_silu_bf16  WanRMS_norm + SiLU, written straight into the consumer conv's input ring
    fvk_gemm_bf16 / fvk_attn_dense_bf16 (qk_dim 384)   1x1 convs and the mid block's single-head attention

MI355X-first layout instead of the reference's NCTHW tensors + ``torch.cat([cache_x, x])`` + ``F.pad``:
  * activations are channels-last bf16 ``[frames, H, W, C]`` so that a conv tap is a contiguous C-vector per pixel
    (16-B coalesced LDS-DMA pieces) and the conv is a GEMM with K = taps x C;
  * every 3x3x3 conv owns a persistent *ring* of (max chunk frames + 2) input frames in HBM: the producer (norm kernel)
    writes the chunk's frames behind the two most recent frames of the previous chunk, the conv kernel addresses frame
    slots modulo the ring — the reference's per-conv feature cache (``:426-431``) without a single cache copy or concat.
    Zero-initialised rings reproduce the first chunk's zero padding and the "Rep"/zeros rule of the temporal upsamplers
    (``:334-353``): the first chunk skips ``time_conv`` and leaves its history at zero.
Same chunking as the reference: one latent frame per decoder pass, 1 pixel frame for the first and 4 for every later one.

Beyond the cached decode, same kernels:
  * ``streaming_decode(z, cache, is_first_chunk)`` / ``get_streaming_cache()`` (wanvae.py:1247-1292): the rings ARE the cache, so a
    streaming cache is just the ring set kept alive between calls;
  * the cache-less family reached with ``use_feature_cache=False`` (``ParallelTiledVAE.decode``, common.py:76-92): ``_decode`` of a
    whole tile in one pass (T latent frames -> 4T pixel frames; zero rings = the 2-frame causal zero padding, ``time_conv`` on every
    frame), ``spatial_tiled_decode`` / ``tiled_decode`` / ``parallel_tiled_decode`` with the reference's in-place cross-fades done by
    ``fvk_vae_blend_f32`` (one launch per tile edge instead of 4 eager kernels per blended row) and the merged video assembled by
    strided copies into one preallocated buffer (no ``torch.cat`` chains).  Tile-parallel decode shards the tile list over the SP
    group exactly as the reference does, but every rank derives all tile shapes from the plan, so the exchange is ONE
    ``all_gather_into_tensor`` over RCCL (the reference adds a size all_gather and an ``all_gather_object`` of shapes).
  * ``postprocess_u8``: (x/2+0.5).clamp(0,1)*255 -> uint8 frames [T,H,W,3] in one pass (decoding.py:210, video_generator.py:912-913).

Constructor input: a reference ``state_dict`` (reference parameter names ``decoder.*``, ``post_quant_conv.*``).
No CPU / eager fallback: ROCm tensors only."""
from __future__ import annotations

import torch

from . import ops

BF16 = torch.bfloat16


def _pad32(c: int) -> int:
    return (c + 31) // 32 * 32


class _Conv:
    """bf16 weight [Cout, kt*kh*kw*Cin_padded] (tap-major, channel-minor) + bias, from a reference conv weight."""

    def __init__(self, sd, name, dev, cin_pad=None):
        w = sd[name + ".weight"].detach().float()
        if w.dim() == 4:
            w = w.unsqueeze(2)
        cout, cin, kt, kh, kw = w.shape
        cp = cin_pad or _pad32(cin)
        if cp != cin:
            w = torch.cat([w, w.new_zeros((cout, cp - cin, kt, kh, kw))], 1)
        self.w = w.permute(0, 2, 3, 4, 1).reshape(cout, -1).to(device=dev, dtype=BF16).contiguous()
        self.b = sd[name + ".bias"].detach().to(device=dev, dtype=BF16).contiguous()
        self.cin, self.cout, self.kt, self.ks = cp, cout, kt, kh


class _Site:
    """Input ring of one cached 3x3x3 conv: [ring, H, W, C] bf16; `start` = slot of the older history frame."""

    def __init__(self, max_t, H, W, C, dev):
        self.buf = torch.zeros((max_t + 2, H, W, C), dtype=BF16, device=dev)
        self.ring, self.start, self.H, self.W, self.C = max_t + 2, 0, H, W, C


class WanVaeDecoderHip:

    def __init__(self, state_dict: dict, dim_mult=(1, 2, 4, 4), num_res_blocks: int = 2, temperal_upsample=(True, True, False),
                 device="cuda", *, use_feature_cache: bool = True, tile_sample_min_height: int = 256, tile_sample_min_width: int = 256,
                 tile_sample_min_num_frames: int = 16, tile_sample_stride_height: int = 192, tile_sample_stride_width: int = 192,
                 tile_sample_stride_num_frames: int = 12, blend_num_frames: int | None = None, use_tiling: bool = False,
                 use_temporal_tiling: bool = False, use_parallel_tiling: bool = False, sp_group=None, fuse_norm: bool = True,
                 frames_per_pass: int = 4, precision: str = "bf16"):
        """``precision``: the reference's ``pipeline_config.vae_precision`` / ``vae_decode_precision`` (configs/pipelines/base.py;
        decoding.py:164-180 wraps ``vae.decode`` in ``torch.autocast(dtype=...)`` unless it is "fp32").  This decoder serves the
        **"bf16"** mode only: bf16 activations in HBM, bf16 MFMA operands, fp32 accumulation and fp32 norm / SiLU arithmetic — the
        same rounding points as the reference's bf16 autocast (conv outputs and residual sums bf16, norms fp32).  Its error against
        the reference's fp32 decode is bounded, quantile by quantile, by the error of the reference's OWN bf16-autocast decode at
        real frame sizes (tests/test_gpu_vae_real.py).  bf16 IS the Wan pipeline's decode default: ``vae_decode_precision = "bf16"``
        (configs/pipelines/wan.py:59) wins over ``vae_precision = "fp32"`` (wan.py:54, kept for the ENCODE) in decoding.py:165-167.
        "fp32" (a user override; the reference's own fp32 test tolerance is atol 1e-5, tests/vaes/test_wan_vae.py:87) is REFUSED
        rather than silently served at lower precision; "fp16" is refused too (no fp16 kernels)."""
        if precision != "bf16":
            raise ValueError(f"WanVaeDecoderHip serves vae_precision='bf16' only (got {precision!r}): the gfx950 decode keeps bf16 "
                             "activations with fp32 accumulation; set pipeline_config.vae_precision / vae_decode_precision to 'bf16' or keep "
                             "the reference decoder for fp32 pixels")
        self.precision = precision
        self.device = torch.device(device)
        # VAEConfig / WanVAEConfig fields (configs/models/vaes/base.py:29-46, wanvae.py:72-82)
        self.use_feature_cache = use_feature_cache
        # RMS-norm + SiLU of the 96-channel (full-resolution) stage fused into the producing conv's epilogue (False = separate norm kernels, A/B)
        self.fuse_norm = fuse_norm
        self.tile_sample_min_height, self.tile_sample_min_width = tile_sample_min_height, tile_sample_min_width
        self.tile_sample_min_num_frames = tile_sample_min_num_frames
        self.tile_sample_stride_height, self.tile_sample_stride_width = tile_sample_stride_height, tile_sample_stride_width
        self.tile_sample_stride_num_frames = tile_sample_stride_num_frames
        self.blend_num_frames = (tile_sample_min_num_frames - tile_sample_stride_num_frames) * 2 if blend_num_frames is None else blend_num_frames
        self.use_tiling, self.use_temporal_tiling, self.use_parallel_tiling = use_tiling, use_temporal_tiling, use_parallel_tiling
        self.sp_group = sp_group  # torch.distributed group of the sequence-parallel ranks (None = single rank)
        self.dim_mult, self.nres, self.t_up = tuple(dim_mult), num_res_blocks, tuple(temperal_upsample)
        sd, dev = state_dict, self.device
        self.f32 = lambda k: sd[k].detach().reshape(-1).to(device=dev, dtype=torch.float32).contiguous()
        # post_quant_conv (1x1x1, z_dim -> z_dim) as a GEMM whose K is padded to 64 and whose N is padded to 32, so that its
        # output IS conv_in's 32-channel (zero-padded) input row
        wq = sd["post_quant_conv.weight"].detach().float().reshape(sd["post_quant_conv.weight"].shape[0], -1)
        z_dim = wq.shape[1]
        self.z_dim = z_dim
        wq_p = torch.zeros((_pad32(wq.shape[0]), 64))
        wq_p[:wq.shape[0], :z_dim] = wq
        bq_p = torch.zeros(_pad32(wq.shape[0]))
        bq_p[:wq.shape[0]] = sd["post_quant_conv.bias"].detach().float()
        self.pq_w, self.pq_b = wq_p.to(dev, BF16), bq_p.to(dev, BF16)
        self.conv_in = _Conv(sd, "decoder.conv_in", dev)
        self.conv_out = _Conv(sd, "decoder.conv_out", dev)
        self.res = {}
        names = ["decoder.mid_block.resnets.0.", "decoder.mid_block.resnets.1."]
        for i in range(len(self.dim_mult)):
            names += [f"decoder.up_blocks.{i}.resnets.{j}." for j in range(num_res_blocks + 1)]
        for p in names:
            r = {"conv1": _Conv(sd, p + "conv1", dev), "conv2": _Conv(sd, p + "conv2", dev), "g1": self.f32(p + "norm1.gamma"),
                 "g2": self.f32(p + "norm2.gamma")}
            if (p + "conv_shortcut.weight") in sd:
                w = sd[p + "conv_shortcut.weight"].detach()
                r["sc_w"] = w.reshape(w.shape[0], -1).to(dev, BF16).contiguous()
                r["sc_b"] = sd[p + "conv_shortcut.bias"].detach().to(dev, BF16).contiguous()
            self.res[p] = r
        p = "decoder.mid_block.attentions.0."
        C = sd[p + "proj.weight"].shape[0]
        self.attn = {"g": self.f32(p + "norm.gamma"), "qkv_w": sd[p + "to_qkv.weight"].detach().reshape(3 * C, C).to(dev, BF16).contiguous(),
                     "qkv_b": sd[p + "to_qkv.bias"].detach().to(dev, BF16).contiguous(),
                     "proj_w": sd[p + "proj.weight"].detach().reshape(C, C).to(dev, BF16).contiguous(),
                     "proj_b": sd[p + "proj.bias"].detach().to(dev, BF16).contiguous()}
        if C not in (128, 384):
            raise ValueError(f"WanVaeDecoderHip: mid-block width {C} unsupported (the gfx950 attention kernels take 128 or 384)")
        self.ups = {}
        for i in range(len(self.dim_mult) - 1):
            p = f"decoder.up_blocks.{i}.upsamplers.0."
            u = {"resample": _Conv(sd, p + "resample.1", dev)}
            if self.t_up[i]:
                tc = _Conv(sd, p + "time_conv", dev)
                half = tc.cout // 2  # output channels [j*C, (j+1)*C) become output frame 2t + j (wanvae.py:354-356)
                u["tc"] = tc
                u["tc_w"] = [tc.w[:half].contiguous(), tc.w[half:].contiguous()]
                u["tc_b"] = [tc.b[:half].contiguous(), tc.b[half:].contiguous()]
            self.ups[i] = u
        self.g_out = self.f32("decoder.norm_out.gamma")
        # cached decode: latent frames per decoder pass after the first.  The reference walks ONE latent frame per pass (wanvae.py:1222-1233,
        # a memory-saving device); the causal convs read their history from rings, so F frames per pass are the same arithmetic per
        # output element (bit-identical, tests/test_gpu_vae.py) with F x the workgroups at the 60x104 stages and 1/F the launches.
        if frames_per_pass < 1:
            raise ValueError("frames_per_pass must be >= 1")
        self.frames_per_pass = int(frames_per_pass)
        self._sites = None
        self._tile_sites = (None, None)  # (geometry key, ring set) of the cache-less tile decode, reused while the geometry repeats
        self.t_ratio = 2**sum(1 for i in range(len(self.dim_mult) - 1) if self.t_up[i])
        self.s_ratio = 2**(len(self.dim_mult) - 1)

    # ------------------------------------------------------------------ per-decode state
    def _site_shapes(self, H, W, t0):
        """(name, frames per pass, H, W, C, is_time_conv_buffer) of every causal-history buffer of one decoder pass over t0 latent frames."""
        t, h, w = t0, H, W
        yield "conv_in", t, h, w, self.conv_in.cin, False
        for p in ("decoder.mid_block.resnets.0.", "decoder.mid_block.resnets.1."):
            yield p + "1", t, h, w, self.res[p]["conv1"].cin, False
            yield p + "2", t, h, w, self.res[p]["conv2"].cin, False
        for i in range(len(self.dim_mult)):
            for j in range(self.nres + 1):
                p = f"decoder.up_blocks.{i}.resnets.{j}."
                yield p + "1", t, h, w, self.res[p]["conv1"].cin, False
                yield p + "2", t, h, w, self.res[p]["conv2"].cin, False
            if i in self.ups:
                if "tc" in self.ups[i]:
                    yield f"tc{i}", t, h, w, self.ups[i]["tc"].cin, True
                    t *= 2
                h, w = 2 * h, 2 * w
        yield "conv_out", t, h, w, self.conv_out.cin, False

    RING_BYTES_MAX = 0xFFFFFF00  # the conv kernels address one ring through a 32-bit buffer descriptor (fvk_vae_conv_bf16 refuses more)

    def _fit_frames_per_pass(self, H, W):
        """Largest F <= frames_per_pass whose widest ring ((t_ratio F + 2) full-resolution frames) stays inside one buffer descriptor."""
        F = self.frames_per_pass
        while F > 1 and max((t + 2) * h * w * c * 2 for _, t, h, w, c, _ in self._site_shapes(H, W, F)) >= self.RING_BYTES_MAX:
            F -= 1
        return F

    def _make_sites(self, H, W, t0=1):
        """One ring per cached conv, zero-initialised (= the reference's empty feature cache); t0 = latent frames per decoder pass."""
        dev = self.device
        sites = {}
        for name, t, h, w, c, is_tc in self._site_shapes(H, W, t0):
            if is_tc:  # linear buffer [2 history frames + chunk frames]; the last resnet writes its output straight into it
                sites[name] = torch.zeros((t + 2, h, w, c), dtype=BF16, device=dev)
            else:
                sites[name] = _Site(t, h, w, c, dev)
        sites["frames_per_pass"] = t0
        return sites

    # ------------------------------------------------------------------ building blocks
    FUSE_NORM_C = (96, 192)  # channel counts whose RMS-norm + SiLU is fused into the producing conv's epilogue (fvk_vae_conv_norm_bf16)

    def _can_fuse(self, conv: _Conv, nxt) -> bool:
        return self.fuse_norm and nxt is not None and conv.w.shape[0] in self.FUSE_NORM_C and nxt[1].C == conv.w.shape[0]

    def _cached_conv(self, site: _Site, conv: _Conv, x, gamma, T, residual=None, out=None, out_f32=None, plane_stride=0, nxt=None,
                     want_raw=True):
        """norm+SiLU of x ([T,H,W,C] un-normed) into the ring (gamma None: the producer's epilogue already wrote it there), then the causal
        conv over [history | chunk].  nxt = (gamma', site') of the consumer: when fusable, this conv's epilogue also writes the consumer's
        normalised input (and only that when want_raw is False)."""
        HW = site.H * site.W
        slot0 = (site.start + 2) % site.ring
        if gamma is not None:
            ops.vae_rmsnorm_silu(x, gamma, site.buf, HW=HW, slot0=slot0, silu=True)
        if nxt is not None and out_f32 is None:
            g2, s2 = nxt
            y = ops.vae_conv_norm(site.buf, conv.w, conv.b, g2, s2.buf, T=T, H=site.H, W=site.W, kt=3, norm_slot0=(s2.start + 2) % s2.ring,
                                  ring_start=site.start, out=out, residual=residual, want_raw=want_raw)
        else:
            y = ops.vae_conv(site.buf, conv.w, conv.b, T=T, H=site.H, W=site.W, kt=3, ks=3, ring_start=site.start, out=out,
                             residual=residual, out_f32=out_f32, plane_stride=plane_stride)
        site.start = (site.start + T) % site.ring
        return y

    def _res_block(self, x, p, T, out=None, prenormed=False, nxt=None):
        """WanResidualBlock.  prenormed: norm1(x) already sits in conv1's ring (written by the producer of x); nxt = (gamma, site) of the
        consumer of this block's output, fused into conv2's epilogue when possible (returns whether it was)."""
        r, S = self.res[p], self._sites
        H, W = S[p + "1"].H, S[p + "1"].W
        if "sc_w" in r:
            h = ops.gemm(x.view(T * H * W, -1), r["sc_w"], r["sc_b"]).view(T, H, W, -1)
        else:
            h = x
        mid = (r["g2"], S[p + "2"])
        fuse_mid = self._can_fuse(r["conv1"], mid)
        y = self._cached_conv(S[p + "1"], r["conv1"], x, None if prenormed else r["g1"], T, nxt=mid if fuse_mid else None, want_raw=False)
        fuse_out = self._can_fuse(r["conv2"], nxt)
        y = self._cached_conv(S[p + "2"], r["conv2"], y, None if fuse_mid else r["g2"], T, residual=h, out=out, nxt=nxt if fuse_out else None)
        return y, fuse_out

    def _mid_attn(self, x):
        """WanAttentionBlock (wanvae.py:479-507): every frame attends over its own H*W pixels."""
        a = self.attn
        T, H, W, C = x.shape
        n = torch.empty((T, H * W, C), dtype=BF16, device=x.device)
        ops.vae_rmsnorm_silu(x, a["g"], n, HW=H * W, slot0=0, silu=False)
        qkv = ops.gemm(n.view(T * H * W, C), a["qkv_w"], a["qkv_b"]).view(T, H * W, 3 * C)
        q, k, v = qkv[:, :, :C], qkv[:, :, C:2 * C], qkv[:, :, 2 * C:]   # the frames are the batch: one launch per pass, not one per frame
        if C == 384:
            o = ops.attn_dense_wide(q, k, v)
        else:
            o = ops.attn_dense(q[:, :, None], k[:, :, None], v[:, :, None]).reshape(T, H * W, C)
        o = ops.gemm(o.view(T * H * W, C), a["proj_w"], a["proj_b"], epilogue=ops.EPI_RESIDUAL_GATE, residual=x.view(T * H * W, C))
        return o.view(T, H, W, C)

    def _decoder_pass(self, T0, skip_time_conv, keep_history, out_f32, plane_stride, trace=None):
        """One decoder pass over the T0 latent frames already sitting in conv_in's ring.
        cached decode : T0 = 1; the first chunk skips time_conv and leaves its history at zero ("Rep", wanvae.py:334-336); later
                        chunks convolve against the history and then shift it (keep_history).
        cache-less    : T0 = whole tile; time_conv on every frame against zero history that is never updated (wanvae.py:355-359)."""
        S = self._sites
        site = S["conv_in"]
        x = ops.vae_conv(site.buf, self.conv_in.w, self.conv_in.b, T=T0, H=site.H, W=site.W, kt=3, ks=3, ring_start=site.start)
        site.start = (site.start + T0) % site.ring
        x, _ = self._res_block(x, "decoder.mid_block.resnets.0.", T0)
        x = self._mid_attn(x)
        x, _ = self._res_block(x, "decoder.mid_block.resnets.1.", T0)
        if trace is not None:
            trace.append(("mid", x))
        T = T0
        n_up = len(self.dim_mult)
        prenormed = False  # the producer of x already wrote norm1(x) into the next residual block's conv1 ring
        for i in range(n_up):
            u = self.ups.get(i)
            for j in range(self.nres + 1):
                p = f"decoder.up_blocks.{i}.resnets.{j}."
                dst = None
                if j == self.nres and u is not None and "tc" in u:
                    dst = S[f"tc{i}"][2:2 + T]  # last resnet of the block writes into the time_conv buffer (frames 2..)
                if j < self.nres:  # consumer of this block's output: the next block's norm1 -> conv1, or norm_out -> conv_out
                    pn = f"decoder.up_blocks.{i}.resnets.{j + 1}."
                    nxt = (self.res[pn]["g1"], S[pn + "1"])
                elif u is None and i == n_up - 1:
                    nxt = (self.g_out, S["conv_out"])
                else:
                    nxt = None
                x, prenormed = self._res_block(x, p, T, out=dst, prenormed=prenormed, nxt=nxt)
            if u is not None:
                _, H, W, C = x.shape
                if "tc" in u and not skip_time_conv:
                    buf = S[f"tc{i}"]
                    y = torch.empty((2 * T, H, W, C), dtype=BF16, device=x.device)
                    for jj in range(2):  # frame interleave: output channel half jj -> output frame 2t + jj
                        ops.vae_conv(buf[:T + 2], u["tc_w"][jj], u["tc_b"][jj], T=T, H=H, W=W, kt=3, ks=1, ring_start=0,
                                     out=y[jj], out_frame_stride=2 * H * W * C)
                    if keep_history:  # history <- the last two input frames (wanvae.py:343-351)
                        if T == 1:
                            buf[0].copy_(buf[1]); buf[1].copy_(buf[2])
                        else:
                            buf[0:2].copy_(buf[T:T + 2])
                    x, T = y, 2 * T
                rs = u["resample"]
                pn = f"decoder.up_blocks.{i + 1}.resnets.0."
                nxt = (self.res[pn]["g1"], S[pn + "1"]) if pn in self.res else None
                if self._can_fuse(rs, nxt):  # the 2x-upsampling conv also writes norm1 of the next stage's first block
                    x = ops.vae_conv_norm(x.contiguous(), rs.w, rs.b, nxt[0], nxt[1].buf, T=T, H=2 * H, W=2 * W, kt=1,
                                          norm_slot0=(nxt[1].start + 2) % nxt[1].ring, upsample2x=True)
                    prenormed = True
                else:
                    x = ops.vae_conv(x.contiguous(), rs.w, rs.b, T=T, H=2 * H, W=2 * W, kt=1, ks=3, upsample2x=True)
            if trace is not None:
                trace.append((f"up{i}", x))
        self._cached_conv(S["conv_out"], self.conv_out, x, None if prenormed else self.g_out, T, out_f32=out_f32, plane_stride=plane_stride)
        return T

    def _latents_cl(self, z):
        """[1, z_dim, T, H, W] -> channels-last bf16 [T, H, W, 64] (zero-padded to the post-quant GEMM's K; layout plumbing)."""
        if z.device.type != "cuda":
            raise RuntimeError("WanVaeDecoderHip runs on a ROCm device only (no CPU fallback)")
        B, Cz, Tl, H, W = z.shape
        if B != 1 or Cz != self.z_dim:
            raise ValueError(f"decode: expected [1,{self.z_dim},T,H,W], got {tuple(z.shape)}")
        zc = torch.zeros((Tl, H, W, 64), dtype=BF16, device=self.device)
        zc[..., :Cz] = z[0].permute(1, 2, 3, 0).to(BF16)
        return zc

    def _out_geometry(self, H, W):
        return H * self.s_ratio, W * self.s_ratio

    def _chunked(self, zc, sites, fresh, out, trace=None):
        """Frame-chunked cached decode of the latent frames zc into out [3, >= n, Ho, Wo]; returns the pixel-frame count."""
        Tl, H, W, _ = zc.shape
        self._sites = sites
        site = sites["conv_in"]
        plane = out.stride(0)
        t_out = 0
        i = 0
        while i < Tl:
            first = fresh and i == 0  # the very first latent frame of a stream always goes alone (it skips time_conv)
            n = 1 if first else min(sites["frames_per_pass"], Tl - i)
            for j in range(n):
                slot = (site.start + 2 + j) % site.ring
                ops.gemm(zc[i + j].view(H * W, 64), self.pq_w, self.pq_b, out=site.buf[slot].view(H * W, -1))
            tr = [] if trace is not None else None
            t_out += self._decoder_pass(n, first, not first, out[:, t_out:], plane, tr)
            if trace is not None:
                trace.append(tr)
            i += n
        self._sites = None
        return t_out

    @torch.no_grad()
    def decode(self, z: torch.Tensor, trace=None) -> torch.Tensor:
        """z [1, z_dim, T, H, W] (de-normalised latents, as ``AutoencoderKLWan.decode`` receives them) ->
        fp32 pixels [1, 3, 1 + 4 (T-1), 8H, 8W] in [-1, 1].  With ``use_feature_cache=False`` the reference's cache-less dispatch
        (plain / spatial / temporal / tile-parallel, common.py:76-92) runs instead (wanvae.py:1213-1214)."""
        if not self.use_feature_cache:
            return self.decode_nocache(z)
        zc = self._latents_cl(z)
        Tl, H, W, _ = zc.shape
        Ho, Wo = self._out_geometry(H, W)
        Tout = 1 + self.t_ratio * (Tl - 1)
        out = torch.empty((self.conv_out.cout, Tout, Ho, Wo), dtype=torch.float32, device=self.device)
        n = self._chunked(zc, self._make_sites(H, W, self._fit_frames_per_pass(H, W)), True, out, trace)
        assert n == Tout
        return out.unsqueeze(0)

    # ------------------------------------------------------------------ streaming decode (wanvae.py:1247-1292)
    def get_streaming_cache(self):
        """The reference returns ``[None] * conv_num``; here the cache is the ring set, created on the first streaming call."""
        return {"sites": None, "fresh": True, "geom": None}

    @torch.no_grad()
    def streaming_decode(self, z: torch.Tensor, cache: dict | None, is_first_chunk: bool = False):
        """Decode the next latent frames of a stream; ``cache`` carries every conv's causal history between calls.  As in the
        reference (non-light VAE), what makes the very first latent frame special is the EMPTY cache, not ``is_first_chunk``."""
        if cache is None:
            cache = self.get_streaming_cache()
        zc = self._latents_cl(z)
        Tl, H, W, _ = zc.shape
        if cache["sites"] is None:
            cache["sites"], cache["geom"] = self._make_sites(H, W, self._fit_frames_per_pass(H, W)), (H, W)
        elif cache["geom"] != (H, W):
            raise ValueError(f"streaming_decode: latent size {(H, W)} differs from the cache's {cache['geom']}")
        Ho, Wo = self._out_geometry(H, W)
        Tout = self.t_ratio * Tl - (self.t_ratio - 1 if cache["fresh"] else 0)
        out = torch.empty((self.conv_out.cout, Tout, Ho, Wo), dtype=torch.float32, device=self.device)
        n = self._chunked(zc, cache["sites"], cache["fresh"], out)
        assert n == Tout
        cache["fresh"] = False
        return out.unsqueeze(0), cache

    # ------------------------------------------------------------------ cache-less tile decode + tiling (common.py)
    def enable_tiling(self, tile_sample_min_height=None, tile_sample_min_width=None, tile_sample_min_num_frames=None,
                      tile_sample_stride_height=None, tile_sample_stride_width=None, tile_sample_stride_num_frames=None,
                      blend_num_frames=None, use_tiling=None, use_temporal_tiling=None, use_parallel_tiling=None) -> None:
        """common.py:376-428, including its reset of blend_num_frames to min - stride when none is given."""
        self.use_tiling = True
        self.tile_sample_min_height = tile_sample_min_height or self.tile_sample_min_height
        self.tile_sample_min_width = tile_sample_min_width or self.tile_sample_min_width
        self.tile_sample_min_num_frames = tile_sample_min_num_frames or self.tile_sample_min_num_frames
        self.tile_sample_stride_height = tile_sample_stride_height or self.tile_sample_stride_height
        self.tile_sample_stride_width = tile_sample_stride_width or self.tile_sample_stride_width
        self.tile_sample_stride_num_frames = tile_sample_stride_num_frames or self.tile_sample_stride_num_frames
        self.blend_num_frames = blend_num_frames if blend_num_frames is not None else \
            self.tile_sample_min_num_frames - self.tile_sample_stride_num_frames
        self.use_tiling = use_tiling or self.use_tiling
        self.use_temporal_tiling = use_temporal_tiling or self.use_temporal_tiling
        self.use_parallel_tiling = use_parallel_tiling or self.use_parallel_tiling

    def disable_tiling(self) -> None:
        self.use_tiling = False

    def _latent_tiles(self):
        s, t = self.s_ratio, self.t_ratio
        return dict(mh=self.tile_sample_min_height // s, mw=self.tile_sample_min_width // s, mt=self.tile_sample_min_num_frames // t,
                    sh=self.tile_sample_stride_height // s, sw=self.tile_sample_stride_width // s,
                    st=self.tile_sample_stride_num_frames // t,
                    bh=self.tile_sample_min_height - self.tile_sample_stride_height,
                    bw=self.tile_sample_min_width - self.tile_sample_stride_width)

    def _decode_tile(self, zc):
        """``_decode`` (wanvae.py:1217-1224) of a channels-last latent tile [T, h, w, 64] -> planar fp32 [3, 4T, 8h, 8w]."""
        Tl, H, W, _ = zc.shape
        key, sites = self._tile_sites
        if key != (Tl, H, W):
            self._tile_sites, sites = (None, None), None  # drop the previous geometry's rings before allocating the next
            sites = self._make_sites(H, W, Tl)
            self._tile_sites = ((Tl, H, W), sites)
        for v in sites.values():  # rings restart at slot 0: slots 0-1 are never written in this mode and stay zero
            if isinstance(v, _Site):
                v.start = 0
        self._sites = sites
        site = sites["conv_in"]
        ops.gemm(zc.reshape(Tl * H * W, 64), self.pq_w, self.pq_b, out=site.buf[2:2 + Tl].view(Tl * H * W, -1))
        Ho, Wo = self._out_geometry(H, W)
        out = torch.empty((self.conv_out.cout, self.t_ratio * Tl, Ho, Wo), dtype=torch.float32, device=self.device)
        n = self._decoder_pass(Tl, False, False, out, out.stride(0))
        self._sites = None
        assert n == out.shape[1]
        return out

    def _merge_spatial(self, tiles, L):
        """_merge_spatial_tiles (common.py:260-273): cross-fade every tile with its upper and left neighbour (in place, in
        the reference's order), then copy its [:stride, :stride] crop into the merged frame."""
        sh, sw = self.tile_sample_stride_height, self.tile_sample_stride_width
        heights = [min(r[0].shape[2], sh) for r in tiles]
        widths = [min(t.shape[3], sw) for t in tiles[0]]
        Cc, T = tiles[0][0].shape[:2]
        out = torch.empty((Cc, T, sum(heights), sum(widths)), dtype=torch.float32, device=self.device)
        y0 = 0
        for i, row in enumerate(tiles):
            x0 = 0
            for j, tile in enumerate(row):
                if i > 0:
                    ops.vae_blend(tiles[i - 1][j], tile, L["bh"], 2)
                if j > 0:
                    ops.vae_blend(row[j - 1], tile, L["bw"], 3)
                out[:, :, y0:y0 + heights[i], x0:x0 + widths[j]].copy_(tile[:, :, :sh, :sw])
                x0 += widths[j]
            y0 += heights[i]
        return out

    def _spatial_base(self, zc, L):
        Tl, H, W, _ = zc.shape
        tiles = [[self._decode_tile(zc[:, i:i + L["mh"], j:j + L["mw"]].contiguous()) for j in range(0, W, L["sw"])]
                 for i in range(0, H, L["sh"])]
        return self._merge_spatial(tiles, L)

    def _temporal_merge(self, parts):
        """The tail of tiled_decode / parallel_tiled_decode (common.py:363-373, 246-257): cross-fade consecutive temporal
        slices over blend_num_frames frames, keep stride (+1 for the first) frames of each."""
        st = self.tile_sample_stride_num_frames
        keep = [min(p.shape[1], st + 1 if i == 0 else st) for i, p in enumerate(parts)]
        Cc, _, Ho, Wo = parts[0].shape
        out = torch.empty((Cc, sum(keep), Ho, Wo), dtype=torch.float32, device=self.device)
        t0 = 0
        for i, p in enumerate(parts):
            if i > 0:
                ops.vae_blend(parts[i - 1], p, self.blend_num_frames, 1)
            out[:, t0:t0 + keep[i]].copy_(p[:, :keep[i]])
            t0 += keep[i]
        return out

    def spatial_tiled_decode(self, zc):
        """Wan override (wanvae.py:1233-1237): ParallelTiledVAE.spatial_tiled_decode minus the t_ratio - 1 leading frames."""
        return self._spatial_base(zc, self._latent_tiles())[:, self.t_ratio - 1:]

    def tiled_decode(self, zc):
        """Wan override (wanvae.py:1226-1231) of ParallelTiledVAE.tiled_decode (common.py:347-374).  Mirrors the reference's
        state change: every call doubles ``blend_num_frames`` persistently."""
        self.blend_num_frames *= 2
        L = self._latent_tiles()
        parts = []
        for i in range(0, zc.shape[0], L["st"]):
            tile = zc[i:i + L["mt"] + 1]
            if self.use_tiling and (tile.shape[2] > L["mw"] or tile.shape[1] > L["mh"]):
                d = self.spatial_tiled_decode(tile)
            else:
                d = self._decode_tile(tile.contiguous())
            parts.append(d[:, 1:] if i > 0 else d)
        return self._temporal_merge(parts)[:, self.t_ratio - 1:]

    def tile_plan(self, T, H, W):
        """(t, h, w) latent origins of parallel_tiled_decode in global tile order + the tile grid (common.py:186-206)."""
        L = self._latent_tiles()
        nt, nh, nw = -(-T // L["st"]), -(-H // L["sh"]), -(-W // L["sw"])
        return [(ti * L["st"], hi * L["sh"], wi * L["sw"]) for ti in range(nt) for hi in range(nh) for wi in range(nw)], (nt, nh, nw)

    def _tile_out_shape(self, origin, T, H, W, L):
        t0, h0, w0 = origin
        tl, hl, wl = min(L["mt"] + 1, T - t0), min(L["mh"], H - h0), min(L["mw"], W - w0)
        return (self.conv_out.cout, self.t_ratio * tl - (1 if t0 > 0 else 0), hl * self.s_ratio, wl * self.s_ratio)

    def parallel_tiled_decode(self, zc):
        """Wan override (wanvae.py:1239-1245) of ParallelTiledVAE.parallel_tiled_decode (common.py:166-258): rank r decodes the
        r-th contiguous run of ceil(n / world) tiles; one fp32 all_gather_into_tensor of the flattened runs (padded to the longest);
        every rank then merges all tiles.  Tile shapes follow from the plan, so no shape / size exchange is needed."""
        import torch.distributed as dist
        self.blend_num_frames *= 2
        L = self._latent_tiles()
        T, H, W, _ = zc.shape
        plan, (nt, nh, nw) = self.tile_plan(T, H, W)
        world = dist.get_world_size(self.sp_group) if self.sp_group is not None else 1
        rank = dist.get_rank(self.sp_group) if self.sp_group is not None else 0
        per = -(-len(plan) // world)
        shapes = [self._tile_out_shape(o, T, H, W, L) for o in plan]
        numel = [s[0] * s[1] * s[2] * s[3] for s in shapes]
        run_len = [sum(numel[r * per:min((r + 1) * per, len(plan))]) for r in range(world)]
        flat = torch.zeros(max(run_len), dtype=torch.float32, device=self.device)
        off = 0
        for k in range(rank * per, min((rank + 1) * per, len(plan))):
            t0, h0, w0 = plan[k]
            d = self._decode_tile(zc[t0:t0 + L["mt"] + 1, h0:h0 + L["mh"], w0:w0 + L["mw"]].contiguous())
            if t0 > 0:
                d = d[:, 1:]
            flat[off:off + numel[k]].view(shapes[k]).copy_(d)
            off += numel[k]
        if world > 1:
            allr = torch.empty(world * flat.numel(), dtype=torch.float32, device=self.device)
            dist.all_gather_into_tensor(allr, flat, group=self.sp_group)
            allr = allr.view(world, -1)
        else:
            allr = flat.view(1, -1)
        decoded = []
        for r in range(world):
            off = 0
            for k in range(r * per, min((r + 1) * per, len(plan))):
                decoded.append(allr[r, off:off + numel[k]].view(shapes[k]))
                off += numel[k]
        parts = [self._merge_spatial([[decoded[(ti * nh + hi) * nw + wi] for wi in range(nw)] for hi in range(nh)], L) for ti in range(nt)]
        return self._temporal_merge(parts)[:, self.t_ratio - 1:]

    @torch.no_grad()
    def decode_nocache(self, z: torch.Tensor) -> torch.Tensor:
        """ParallelTiledVAE.decode (common.py:76-92).  Frame counts are the reference's, including its short temporal-tiling output."""
        import torch.distributed as dist
        zc = self._latents_cl(z)
        T, H, W, _ = zc.shape
        L = self._latent_tiles()
        n_out = (T - 1) * self.t_ratio + 1
        world = dist.get_world_size(self.sp_group) if self.sp_group is not None else 1
        if self.use_tiling and self.use_parallel_tiling and world > 1:
            y = self.parallel_tiled_decode(zc)
        elif self.use_tiling and self.use_temporal_tiling and T > L["mt"]:
            y = self.tiled_decode(zc)
        elif self.use_tiling and (W > L["mw"] or H > L["mh"]):
            y = self.spatial_tiled_decode(zc)
        else:
            y = self._decode_tile(zc)
        self._tile_sites = (None, None)
        return y[:, :n_out].unsqueeze(0)

    # ------------------------------------------------------------------ pixels -> uint8 frames
    @torch.no_grad()
    def postprocess_u8(self, pixels: torch.Tensor) -> torch.Tensor:
        """[1, 3, T, H, W] fp32 in [-1, 1] -> uint8 [T, H, W, 3]: DecodingStage's (x/2+0.5).clamp(0,1) followed by VideoGenerator's
        on-device (x*255).clamp(0,255).to(uint8), in the frame-major channels-last layout the video writer consumes."""
        return ops.vae_postprocess_u8(pixels)
