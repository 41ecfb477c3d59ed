"""Checkpoint → HBM loading for the HIP models.

Reference flow being replaced (fastvideo/models/loader/component_loader.py:1013-1150 → fsdp_load.py:121-206): build the module on
``torch.device("meta")``, stream the diffusers-format ``*.safetensors`` shards, rename every key through the arch config's
``param_names_mapping`` regex table (fastvideo/configs/models/dits/wanvideo.py:16-43, applied by
fastvideo/models/loader/utils.py:27-56: first matching pattern wins, unmatched names pass through), shard with FSDP, then
optionally ``convert_model_to_fp8`` (fastvideo/layers/quantization/fp8_config.py:211-245).

MI355X-first: a Wan2.2-A14B expert is 28 GB of bf16 and the part has 288 GB of HBM, so there is no meta-device / FSDP dance — each
tensor is read from its shard straight into device memory (one shard file mapped at a time), renamed, and handed to the HIP model's
constructor, which packs the layouts the kernels want (fused QKV rows, per-layer text-KV panel, [Cout, taps*Cin] conv weights) and
quantises to e4m3 on the GPU when ``quantization`` is set.  Nothing here touches a kernel; it is host-side name plumbing."""
from __future__ import annotations

import glob
import json
import os
import re
from typing import Callable, Iterator

import torch

# diffusers (HF) parameter name -> reference parameter name; first match wins (fastvideo/configs/models/dits/wanvideo.py:16-43)
_WAN_DIT_RULES = [(re.compile(p), r) for p, r in (
    (r"^patch_embedding\.(.*)$", r"patch_embedding.proj.\1"),
    (r"^condition_embedder\.text_embedder\.linear_1\.(.*)$", r"condition_embedder.text_embedder.fc_in.\1"),
    (r"^condition_embedder\.text_embedder\.linear_2\.(.*)$", r"condition_embedder.text_embedder.fc_out.\1"),
    (r"^condition_embedder\.time_embedder\.linear_1\.(.*)$", r"condition_embedder.time_embedder.mlp.fc_in.\1"),
    (r"^condition_embedder\.time_embedder\.linear_2\.(.*)$", r"condition_embedder.time_embedder.mlp.fc_out.\1"),
    (r"^condition_embedder\.delta_embedder\.linear_1\.(.*)$", r"condition_embedder.delta_embedder.mlp.fc_in.\1"),
    (r"^condition_embedder\.delta_embedder\.linear_2\.(.*)$", r"condition_embedder.delta_embedder.mlp.fc_out.\1"),
    (r"^condition_embedder\.time_proj\.(.*)$", r"condition_embedder.time_modulation.linear.\1"),
    (r"^condition_embedder\.image_embedder\.ff\.net\.0\.proj\.(.*)$", r"condition_embedder.image_embedder.ff.fc_in.\1"),
    (r"^condition_embedder\.image_embedder\.ff\.net\.2\.(.*)$", r"condition_embedder.image_embedder.ff.fc_out.\1"),
    (r"^blocks\.(\d+)\.attn1\.to_(q|k|v)\.(.*)$", r"blocks.\1.to_\2.\3"),
    (r"^blocks\.(\d+)\.attn1\.to_out\.0\.(.*)$", r"blocks.\1.to_out.\2"),
    (r"^blocks\.(\d+)\.attn1\.norm_(q|k)\.(.*)$", r"blocks.\1.norm_\2.\3"),
    (r"^blocks\.(\d+)\.attn2\.to_out\.0\.(.*)$", r"blocks.\1.attn2.to_out.\2"),
    (r"^blocks\.(\d+)\.ffn\.net\.0\.proj\.(.*)$", r"blocks.\1.ffn.fc_in.\2"),
    (r"^blocks\.(\d+)\.ffn\.net\.2\.(.*)$", r"blocks.\1.ffn.fc_out.\2"),
    (r"^blocks\.(\d+)\.norm2\.(.*)$", r"blocks.\1.self_attn_residual_norm.norm.\2"),
)]

# original-Wan (LoRA / official checkpoint) names -> diffusers names, applied BEFORE the table above (wanvideo.py:48-61)
_WAN_OFFICIAL_RULES = [(re.compile(p), r) for p, r in (
    (r"^blocks\.(\d+)\.self_attn\.(q|k|v)\.(.*)$", r"blocks.\1.attn1.to_\2.\3"),
    (r"^blocks\.(\d+)\.self_attn\.o\.(.*)$", r"blocks.\1.attn1.to_out.0.\2"),
    (r"^blocks\.(\d+)\.cross_attn\.(q|k|v)\.(.*)$", r"blocks.\1.attn2.to_\2.\3"),
    (r"^blocks\.(\d+)\.cross_attn\.o\.(.*)$", r"blocks.\1.attn2.to_out.0.\2"),
    (r"^blocks\.(\d+)\.ffn\.0\.(.*)$", r"blocks.\1.ffn.fc_in.\2"),
    (r"^blocks\.(\d+)\.ffn\.2\.(.*)$", r"blocks.\1.ffn.fc_out.\2"),
)]


def _apply(rules, name: str) -> str:
    for pat, rep in rules:
        if pat.match(name):
            return pat.sub(rep, name)
    return name


def wan_dit_param_name(hf_name: str, official_names: bool = False) -> str:
    """diffusers ``WanTransformer3DModel`` key -> reference ``WanTransformer3DModel`` key."""
    if official_names:
        hf_name = _apply(_WAN_OFFICIAL_RULES, hf_name)
    return _apply(_WAN_DIT_RULES, hf_name)


def safetensors_files(path: str) -> list[str]:
    """The shard files of a component directory (or the single file itself), index-ordered when an index json exists."""
    if os.path.isfile(path):
        return [path]
    idx = glob.glob(os.path.join(path, "*.safetensors.index.json"))
    if idx:
        with open(idx[0]) as f:
            shard_of = json.load(f)["weight_map"]
        files = sorted(set(shard_of.values()))
        return [os.path.join(path, f) for f in files]
    files = sorted(glob.glob(os.path.join(path, "*.safetensors")))
    if not files:
        raise FileNotFoundError(f"no *.safetensors under {path}")
    return files


def iter_safetensors(path: str, device="cpu") -> Iterator[tuple[str, torch.Tensor]]:
    """(name, tensor) over every shard, each tensor materialised directly on ``device`` (weight_utils.py:150-190 without the
    rank-0 broadcast: every rank maps the files itself; page cache makes the second reader free)."""
    from safetensors import safe_open
    for fn in safetensors_files(path):
        with safe_open(fn, framework="pt", device=str(device)) as f:
            for name in f.keys():  # noqa: SIM118
                yield name, f.get_tensor(name)


def load_state_dict(path: str, device="cpu", rename: Callable[[str], str] | None = None, dtype: torch.dtype | None = None,
                    keep: Callable[[str], bool] | None = None) -> dict:
    sd = {}
    for name, t in iter_safetensors(path, device):
        if rename is not None:
            name = rename(name)
        if keep is not None and not keep(name):
            continue
        if name in sd:
            raise ValueError(f"{path}: two checkpoint keys map to {name}")
        sd[name] = t.to(dtype) if (dtype is not None and t.is_floating_point()) else t
    return sd


def _config(path: str) -> dict:
    cfg = os.path.join(path if os.path.isdir(path) else os.path.dirname(path), "config.json")
    if not os.path.exists(cfg):
        raise FileNotFoundError(f"{cfg} not found (diffusers component directories carry their architecture there)")
    with open(cfg) as f:
        return json.load(f)


def load_wan_transformer(path: str, device="cuda", attention: str = "dense", quantization: str | None = None,
                         official_names: bool = False, model_cls=None, **model_kw):
    """``<model>/transformer`` (or ``transformer_2``) directory of a diffusers Wan checkpoint -> WanTransformer3DModelHip.
    ``quantization`` = None | "fp8" | "fp8_channel" converts the linears named by the reference's ``_FP8_SUFFIXES`` on load
    (``transformer_quant=get_quantization_config("FP8")()``, docs/inference/optimizations.md:282-303)."""
    cfg = _config(path)
    if cfg.get("image_dim") or cfg.get("added_kv_proj_dim"):
        raise ValueError("load_wan_transformer: I2V checkpoints (image_dim / added_kv_proj_dim) are not supported by the HIP model")
    if cfg.get("qk_norm", "rms_norm_across_heads") != "rms_norm_across_heads" or not cfg.get("cross_attn_norm", True):
        raise ValueError("load_wan_transformer: expected qk_norm='rms_norm_across_heads' and cross_attn_norm=True (Wan2.1 / 2.2)")
    sd = load_state_dict(path, device, lambda n: wan_dit_param_name(n, official_names), dtype=torch.bfloat16)
    if model_cls is None:
        from .wan_dit import WanTransformer3DModelHip as model_cls
    return model_cls(sd, cfg["num_attention_heads"], cfg["attention_head_dim"], tuple(cfg.get("patch_size", (1, 2, 2))),
                     cfg.get("eps", 1e-6), cfg.get("freq_dim", 256), attention=attention, device=device, quantization=quantization,
                     **model_kw)


def load_wan_vae_decoder(path: str, device="cuda", model_cls=None, **model_kw):
    """``<model>/vae`` directory -> (WanVaeDecoderHip, latents_mean, latents_std).  The reference VAE uses the checkpoint's own key
    names (no renaming); only ``decoder.*`` and ``post_quant_conv.*`` are read into HBM — the encoder is not on the path."""
    cfg = _config(path)
    sd = load_state_dict(path, device, keep=lambda n: n.startswith(("decoder.", "post_quant_conv.")), dtype=torch.float32)
    t_up = tuple(cfg.get("temperal_downsample", (False, True, True)))[::-1]  # wanvae.py: decoder mirrors the encoder
    if model_cls is None:
        from .wan_vae import WanVaeDecoderHip as model_cls
    dec = model_cls(sd, tuple(cfg.get("dim_mult", (1, 2, 4, 4))), cfg.get("num_res_blocks", 2), t_up, device=device, **model_kw)
    return dec, cfg.get("latents_mean"), cfg.get("latents_std")
