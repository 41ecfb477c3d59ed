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
                 device="cuda"):
        self.device = torch.device(device)
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
        self._sites = None
        self._geom = None

    # ------------------------------------------------------------------ per-decode state
    def _make_sites(self, H, W):
        """One ring per cached conv, zero-initialised (= the reference's empty feature cache)."""
        dev = self.device
        sites = {}
        t, h, w = 1, H, W
        sites["conv_in"] = _Site(1, h, w, self.conv_in.cin, dev)
        for p in ("decoder.mid_block.resnets.0.", "decoder.mid_block.resnets.1."):
            sites[p + "1"] = _Site(1, h, w, self.res[p]["conv1"].cin, dev)
            sites[p + "2"] = _Site(1, h, w, self.res[p]["conv2"].cin, dev)
        for i in range(len(self.dim_mult)):
            for j in range(self.nres + 1):
                p = f"decoder.up_blocks.{i}.resnets.{j}."
                sites[p + "1"] = _Site(t, h, w, self.res[p]["conv1"].cin, dev)
                sites[p + "2"] = _Site(t, h, w, self.res[p]["conv2"].cin, dev)
            if i in self.ups:
                if "tc" in self.ups[i]:
                    # linear buffer [2 history frames + chunk frames]; the last resnet writes its output straight into it
                    sites[f"tc{i}"] = torch.zeros((t + 2, h, w, self.ups[i]["tc"].cin), dtype=BF16, device=dev)
                    t *= 2
                h, w = 2 * h, 2 * w
        sites["conv_out"] = _Site(t, h, w, self.conv_out.cin, dev)
        return sites

    # ------------------------------------------------------------------ building blocks
    def _cached_conv(self, site: _Site, conv: _Conv, x, gamma, T, residual=None, out=None, out_f32=None, plane_stride=0):
        """norm+SiLU of x ([T,H,W,C] un-normed) into the ring, then the causal conv over [history | chunk]."""
        HW = site.H * site.W
        slot0 = (site.start + 2) % site.ring
        if gamma is not None:
            ops.vae_rmsnorm_silu(x, gamma, site.buf, HW=HW, slot0=slot0, silu=True)
        y = ops.vae_conv(site.buf, conv.w, conv.b, T=T, H=site.H, W=site.W, kt=3, ks=3, ring_start=site.start, out=out,
                         residual=residual, out_f32=out_f32, plane_stride=plane_stride)
        site.start = (site.start + T) % site.ring
        return y

    def _res_block(self, x, p, T, out=None):
        r, S = self.res[p], self._sites
        H, W = S[p + "1"].H, S[p + "1"].W
        if "sc_w" in r:
            h = ops.gemm(x.view(T * H * W, -1), r["sc_w"], r["sc_b"]).view(T, H, W, -1)
        else:
            h = x
        y = self._cached_conv(S[p + "1"], r["conv1"], x, r["g1"], T)
        return self._cached_conv(S[p + "2"], r["conv2"], y, r["g2"], T, residual=h, out=out)

    def _mid_attn(self, x):
        a = self.attn
        _, H, W, C = x.shape
        n = torch.empty((1, H * W, C), dtype=BF16, device=x.device)
        ops.vae_rmsnorm_silu(x, a["g"], n, HW=H * W, slot0=0, silu=False)
        qkv = ops.gemm(n.view(H * W, C), a["qkv_w"], a["qkv_b"])
        q, k, v = qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:]
        if C == 384:
            o = ops.attn_dense_wide(q, k, v)
        else:
            o = ops.attn_dense(q[None, :, None], k[None, :, None], v[None, :, None]).reshape(H * W, C)
        o = ops.gemm(o, a["proj_w"], a["proj_b"], epilogue=ops.EPI_RESIDUAL_GATE, residual=x.view(H * W, C))
        return o.view(1, H, W, C)

    def _decoder_chunk(self, xin, first, out_f32, plane_stride, trace=None):
        """One latent frame.  xin: view of conv_in's ring slot already holding the post-quant frame."""
        S = self._sites
        site = S["conv_in"]
        x = ops.vae_conv(site.buf, self.conv_in.w, self.conv_in.b, T=1, H=site.H, W=site.W, kt=3, ks=3, ring_start=site.start)
        site.start = (site.start + 1) % site.ring
        x = self._res_block(x, "decoder.mid_block.resnets.0.", 1)
        x = self._mid_attn(x)
        x = self._res_block(x, "decoder.mid_block.resnets.1.", 1)
        if trace is not None:
            trace.append(("mid", x))
        T = 1
        n_up = len(self.dim_mult)
        for i in range(n_up):
            u = self.ups.get(i)
            for j in range(self.nres + 1):
                p = f"decoder.up_blocks.{i}.resnets.{j}."
                dst = None
                if j == self.nres and u is not None and "tc" in u:
                    dst = S[f"tc{i}"][2:2 + T]  # last resnet of the block writes into the time_conv buffer (frames 2..)
                x = self._res_block(x, p, T, out=dst)
            if u is not None:
                _, H, W, C = x.shape
                if "tc" in u and not first:
                    buf = S[f"tc{i}"]
                    y = torch.empty((2 * T, H, W, C), dtype=BF16, device=x.device)
                    for jj in range(2):  # frame interleave: output channel half jj -> output frame 2t + jj
                        ops.vae_conv(buf[:T + 2], u["tc_w"][jj], u["tc_b"][jj], T=T, H=H, W=W, kt=3, ks=1, ring_start=0,
                                     out=y[jj], out_frame_stride=2 * H * W * C)
                    # history <- the last two input frames (wanvae.py:343-351)
                    if T == 1:
                        buf[0].copy_(buf[1]); buf[1].copy_(buf[2])
                    else:
                        buf[0:2].copy_(buf[T:T + 2])
                    x, T = y, 2 * T
                rs = u["resample"]
                x = ops.vae_conv(x.contiguous(), rs.w, rs.b, T=T, H=2 * H, W=2 * W, kt=1, ks=3, upsample2x=True)
            if trace is not None:
                trace.append((f"up{i}", x))
        self._cached_conv(S["conv_out"], self.conv_out, x, self.g_out, T, out_f32=out_f32, plane_stride=plane_stride)
        return T

    @torch.no_grad()
    def decode(self, z: torch.Tensor, trace=None) -> torch.Tensor:
        """z [1, z_dim, T, H, W] (de-normalised latents, as ``AutoencoderKLWan.decode`` receives them) ->
        fp32 pixels [1, 3, 1 + 4 (T-1), 8H, 8W] in [-1, 1]."""
        if z.device.type != "cuda":
            raise RuntimeError("WanVaeDecoderHip runs on a ROCm device only (no CPU fallback)")
        B, Cz, Tl, H, W = z.shape
        if B != 1 or Cz != self.z_dim:
            raise ValueError(f"decode: expected [1,{self.z_dim},T,H,W], got {tuple(z.shape)}")
        dev = self.device
        self._sites = self._make_sites(H, W)
        n_sp = len(self.dim_mult) - 1
        Ho, Wo = H * 2**n_sp, W * 2**n_sp
        n_t = sum(1 for i in range(n_sp) if self.t_up[i])
        Tout = 1 + (2**n_t) * (Tl - 1)
        cout = self.conv_out.cout
        out = torch.empty((cout, Tout, Ho, Wo), dtype=torch.float32, device=dev)
        # latents: channels-last, zero-padded to the post-quant GEMM's K = 64 (layout plumbing)
        zc = torch.zeros((Tl, H, W, 64), dtype=BF16, device=dev)
        zc[..., :Cz] = z[0].permute(1, 2, 3, 0).to(BF16)
        site = self._sites["conv_in"]
        t_out = 0
        for i in range(Tl):
            slot = (site.start + 2) % site.ring
            ops.gemm(zc[i].view(H * W, 64), self.pq_w, self.pq_b, out=site.buf[slot].view(H * W, -1))
            tr = [] if trace is not None else None
            T = self._decoder_chunk(None, i == 0, out[:, t_out:], Tout * Ho * Wo, tr)
            if trace is not None:
                trace.append(tr)
            t_out += T
        self._sites = None
        return out.unsqueeze(0)
