"""Architecture constants of the Wan DiT family and a synthetic-weight generator with the reference's parameter
names (fastvideo/models/dits/wanvideo.py module tree; geometry from fastvideo/configs/models/dits/wanvideo.py:64-76 and
fastvideo/tests/golden_gate/test_wan_t2v.py:20-33).  Used by bench.py / smoke when no checkpoint is available."""
from __future__ import annotations

from dataclasses import dataclass

import torch


@dataclass(frozen=True)
class WanConfig:
    name: str
    num_heads: int
    head_dim: int
    ffn_dim: int
    num_layers: int
    text_dim: int = 4096
    freq_dim: int = 256
    in_channels: int = 16
    out_channels: int = 16
    patch_size: tuple = (1, 2, 2)
    eps: float = 1e-6

    @property
    def dim(self):
        return self.num_heads * self.head_dim


WAN21_T2V_1_3B = WanConfig("Wan2.1-T2V-1.3B", 12, 128, 8960, 30)
WAN22_T2V_A14B = WanConfig("Wan2.2-T2V-A14B", 40, 128, 13824, 40)
WAN_TINY = WanConfig("wan-tiny", 2, 128, 512, 2, text_dim=64)

# latent shapes [B,C,T,H,W] of BASELINE.json's configs
LATENT_81F_480P = (1, 16, 21, 60, 104)   # 81f x 832 x 480  -> 21*30*52 = 32760 tokens
LATENT_CFG1 = (1, 16, 9, 64, 64)         # plumbing case     -> 9216 tokens
LATENT_81F_720P = (1, 16, 21, 90, 160)   # 81f x 1280 x 720  -> 75600 tokens


def algorithmic_flops(cfg: WanConfig, S: int, L_text: int = 512) -> dict:
    """2*MAC FLOPs of one forward (SURVEY.md §8d formula), split by kernel class."""
    d, f, L = cfg.dim, cfg.ffn_dim, cfg.num_layers
    per = dict(self_attn=4.0 * S * S * d, qkvo=8.0 * S * d * d, cross=4.0 * S * d * d + 4.0 * L_text * d * d + 4.0 * S * L_text * d,
               ffn=4.0 * S * d * f)
    out = {k: v * L for k, v in per.items()}
    out["total"] = sum(out.values())
    return out


def random_state_dict(cfg: WanConfig, seed: int = 0, device="cpu", dtype=torch.bfloat16, with_vsa_gate: bool = False) -> dict:
    gdev = "cuda" if str(device).startswith("cuda") else "cpu"   # generate where the weights live (1.4 B params)
    g = torch.Generator(device=gdev).manual_seed(seed)
    d, f = cfg.dim, cfg.ffn_dim

    def mat(n_out, n_in):
        bound = (6.0 / (n_in + n_out))**0.5
        return ((torch.rand((n_out, n_in), generator=g, device=gdev) * 2 - 1) * bound).to(dtype).to(device)

    def vec(n, std=0.02, mean=0.0):
        return (torch.randn((n, ), generator=g, device=gdev) * std + mean).to(dtype).to(device)

    pt, ph, pw = cfg.patch_size
    sd = {}
    sd["patch_embedding.proj.weight"] = mat(d, cfg.in_channels * pt * ph * pw).view(d, cfg.in_channels, pt, ph, pw)
    sd["patch_embedding.proj.bias"] = vec(d)
    ce = "condition_embedder."
    for name, (o, i) in {"time_embedder.mlp.fc_in": (d, cfg.freq_dim), "time_embedder.mlp.fc_out": (d, d),
                         "time_modulation.linear": (6 * d, d), "text_embedder.fc_in": (d, cfg.text_dim),
                         "text_embedder.fc_out": (d, d)}.items():
        sd[ce + name + ".weight"], sd[ce + name + ".bias"] = mat(o, i), vec(o)
    sd["scale_shift_table"] = (torch.randn((1, 2, d), generator=g, device=gdev) / d**0.5).to(dtype).to(device)
    sd["proj_out.weight"], sd["proj_out.bias"] = mat(cfg.out_channels * pt * ph * pw, d), vec(cfg.out_channels * pt * ph * pw)
    for i in range(cfg.num_layers):
        p = f"blocks.{i}."
        sd[p + "scale_shift_table"] = (torch.randn((1, 6, d), generator=g, device=gdev) / d**0.5).to(dtype).to(device)
        names = ["to_q", "to_k", "to_v", "to_out", "attn2.to_q", "attn2.to_k", "attn2.to_v", "attn2.to_out"]
        if with_vsa_gate:
            names.append("to_gate_compress")
        for n in names:
            sd[p + n + ".weight"], sd[p + n + ".bias"] = mat(d, d), vec(d)
        for n in ("norm_q", "norm_k", "attn2.norm_q", "attn2.norm_k"):
            sd[p + n + ".weight"] = vec(d, 0.02, 1.0)
        sd[p + "self_attn_residual_norm.norm.weight"] = vec(d, 0.02, 1.0)
        sd[p + "self_attn_residual_norm.norm.bias"] = vec(d)
        sd[p + "ffn.fc_in.weight"], sd[p + "ffn.fc_in.bias"] = mat(f, d), vec(f)
        sd[p + "ffn.fc_out.weight"], sd[p + "ffn.fc_out.bias"] = mat(d, f), vec(d)
    return sd


# ---------------------------------------------------------------------------------------------------------------------------
# Wan VAE decoder (fastvideo/models/vaes/wanvae.py:857-953, fastvideo/configs/models/vaes/wanvae.py:10-17)
def wan_vae_param_spec(base_dim: int = 96, z_dim: int = 16, dim_mult=(1, 2, 4, 4), num_res_blocks: int = 2,
                       temperal_upsample=(True, True, False), out_channels: int = 3):
    """Ordered [(reference parameter name, shape)] of the decoder + post_quant_conv (for synthetic weights)."""
    dims = [base_dim * u for u in [dim_mult[-1]] + list(dim_mult[::-1])]
    spec = [("post_quant_conv.weight", (z_dim, z_dim, 1, 1, 1)), ("post_quant_conv.bias", (z_dim,)),
            ("decoder.conv_in.weight", (dims[0], z_dim, 3, 3, 3)), ("decoder.conv_in.bias", (dims[0],))]

    def res(p, cin, cout):
        s = [(p + "norm1.gamma", (cin, 1, 1, 1)), (p + "conv1.weight", (cout, cin, 3, 3, 3)), (p + "conv1.bias", (cout,)),
             (p + "norm2.gamma", (cout, 1, 1, 1)), (p + "conv2.weight", (cout, cout, 3, 3, 3)), (p + "conv2.bias", (cout,))]
        if cin != cout:
            s += [(p + "conv_shortcut.weight", (cout, cin, 1, 1, 1)), (p + "conv_shortcut.bias", (cout,))]
        return s

    d0 = dims[0]
    spec += res("decoder.mid_block.resnets.0.", d0, d0)
    a = "decoder.mid_block.attentions.0."
    spec += [(a + "norm.gamma", (d0, 1, 1)), (a + "to_qkv.weight", (3 * d0, d0, 1, 1)), (a + "to_qkv.bias", (3 * d0,)),
             (a + "proj.weight", (d0, d0, 1, 1)), (a + "proj.bias", (d0,))]
    spec += res("decoder.mid_block.resnets.1.", d0, d0)
    for i, (cin, cout) in enumerate(zip(dims[:-1], dims[1:])):
        if i > 0:
            cin //= 2
        for j in range(num_res_blocks + 1):
            spec += res(f"decoder.up_blocks.{i}.resnets.{j}.", cin if j == 0 else cout, cout)
        if i != len(dim_mult) - 1:
            u = f"decoder.up_blocks.{i}.upsamplers.0."
            spec += [(u + "resample.1.weight", (cout // 2, cout, 3, 3)), (u + "resample.1.bias", (cout // 2,))]
            if temperal_upsample[i]:
                spec += [(u + "time_conv.weight", (2 * cout, cout, 3, 1, 1)), (u + "time_conv.bias", (2 * cout,))]
    spec += [("decoder.norm_out.gamma", (dims[-1], 1, 1, 1)), ("decoder.conv_out.weight", (out_channels, dims[-1], 3, 3, 3)),
             ("decoder.conv_out.bias", (out_channels,))]
    return spec


def vae_decode_flops(T: int, H: int, W: int, base_dim: int = 96, z_dim: int = 16, dim_mult=(1, 2, 4, 4), num_res_blocks: int = 2,
                     temperal_upsample=(True, True, False)) -> float:
    """2*MAC FLOPs of the chunked decode of a [T,H,W] latent (convs + mid attention), chunk 0 has no temporal upsampling."""
    spec = dict(wan_vae_param_spec(base_dim, z_dim, dim_mult, num_res_blocks, temperal_upsample))
    total = 0.0
    for c in range(T):
        t, h, w = 1, H, W

        def conv(name, frames, hh, ww):
            s = spec[name + ".weight"]
            k = 1
            for d in s[1:]:
                k *= d
            return 2.0 * frames * hh * ww * s[0] * k

        total += conv("decoder.conv_in", 1, h, w)
        d0 = spec["decoder.conv_in.weight"][0]
        for p in ("decoder.mid_block.resnets.0.", "decoder.mid_block.resnets.1."):
            total += conv(p + "conv1", 1, h, w) + conv(p + "conv2", 1, h, w)
        total += conv("decoder.mid_block.attentions.0.to_qkv", 1, h, w) + conv("decoder.mid_block.attentions.0.proj", 1, h, w)
        total += 4.0 * (h * w)**2 * d0
        for i in range(len(dim_mult)):
            for j in range(num_res_blocks + 1):
                p = f"decoder.up_blocks.{i}.resnets.{j}."
                total += conv(p + "conv1", t, h, w) + conv(p + "conv2", t, h, w)
                if (p + "conv_shortcut.weight") in spec:
                    total += conv(p + "conv_shortcut", t, h, w)
            if i != len(dim_mult) - 1:
                u = f"decoder.up_blocks.{i}.upsamplers.0."
                if temperal_upsample[i] and c > 0:
                    total += conv(u + "time_conv", t, h, w)
                    t *= 2
                h, w = 2 * h, 2 * w
                total += conv(u + "resample.1", t, h, w)
        total += conv("decoder.conv_out", t, h, w)
    return total
