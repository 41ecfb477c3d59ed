"""fastvideo_amd.loader: diffusers-format safetensors -> reference parameter names -> HIP model constructor.
The rename table is checked against the REAL reference's ``param_names_mapping`` applied by its own
``get_param_names_mapping`` (live, when /root/reference is present) and against a frozen list of (hf, reference) pairs."""
import json
import os
import re

import pytest
import torch
from safetensors.torch import save_file

from fastvideo_amd import loader as L
from oracle import ref_loader as R

PAIRS = [
    ("patch_embedding.weight", "patch_embedding.proj.weight"),
    ("condition_embedder.text_embedder.linear_1.weight", "condition_embedder.text_embedder.fc_in.weight"),
    ("condition_embedder.text_embedder.linear_2.bias", "condition_embedder.text_embedder.fc_out.bias"),
    ("condition_embedder.time_embedder.linear_1.weight", "condition_embedder.time_embedder.mlp.fc_in.weight"),
    ("condition_embedder.time_embedder.linear_2.weight", "condition_embedder.time_embedder.mlp.fc_out.weight"),
    ("condition_embedder.time_proj.bias", "condition_embedder.time_modulation.linear.bias"),
    ("blocks.0.attn1.to_q.weight", "blocks.0.to_q.weight"),
    ("blocks.12.attn1.to_k.bias", "blocks.12.to_k.bias"),
    ("blocks.3.attn1.to_v.weight", "blocks.3.to_v.weight"),
    ("blocks.39.attn1.to_out.0.weight", "blocks.39.to_out.weight"),
    ("blocks.7.attn1.norm_q.weight", "blocks.7.norm_q.weight"),
    ("blocks.7.attn1.norm_k.weight", "blocks.7.norm_k.weight"),
    ("blocks.2.attn2.to_q.weight", "blocks.2.attn2.to_q.weight"),           # cross-attention q/k/v/norms keep their names
    ("blocks.2.attn2.norm_k.weight", "blocks.2.attn2.norm_k.weight"),
    ("blocks.2.attn2.to_out.0.bias", "blocks.2.attn2.to_out.bias"),
    ("blocks.5.ffn.net.0.proj.weight", "blocks.5.ffn.fc_in.weight"),
    ("blocks.5.ffn.net.2.bias", "blocks.5.ffn.fc_out.bias"),
    ("blocks.5.norm2.weight", "blocks.5.self_attn_residual_norm.norm.weight"),
    ("blocks.5.scale_shift_table", "blocks.5.scale_shift_table"),
    ("scale_shift_table", "scale_shift_table"),
    ("proj_out.weight", "proj_out.weight"),
]


def test_rename_table_frozen_pairs():
    for hf, ref in PAIRS:
        assert L.wan_dit_param_name(hf) == ref, hf
    assert L.wan_dit_param_name("blocks.4.self_attn.q.weight", official_names=True) == "blocks.4.to_q.weight"
    assert L.wan_dit_param_name("blocks.4.cross_attn.o.bias", official_names=True) == "blocks.4.attn2.to_out.bias"
    assert L.wan_dit_param_name("blocks.4.ffn.0.weight", official_names=True) == "blocks.4.ffn.fc_in.weight"


def _hf_name(ref_name: str) -> str:
    """Inverse of the table for the keys of the tiny fixture (test helper)."""
    inv = [(r"^patch_embedding\.proj\.(.*)$", r"patch_embedding.\1"),
           (r"^condition_embedder\.text_embedder\.fc_in\.(.*)$", r"condition_embedder.text_embedder.linear_1.\1"),
           (r"^condition_embedder\.text_embedder\.fc_out\.(.*)$", r"condition_embedder.text_embedder.linear_2.\1"),
           (r"^condition_embedder\.time_embedder\.mlp\.fc_in\.(.*)$", r"condition_embedder.time_embedder.linear_1.\1"),
           (r"^condition_embedder\.time_embedder\.mlp\.fc_out\.(.*)$", r"condition_embedder.time_embedder.linear_2.\1"),
           (r"^condition_embedder\.time_modulation\.linear\.(.*)$", r"condition_embedder.time_proj.\1"),
           (r"^blocks\.(\d+)\.to_(q|k|v)\.(.*)$", r"blocks.\1.attn1.to_\2.\3"),
           (r"^blocks\.(\d+)\.to_out\.(.*)$", r"blocks.\1.attn1.to_out.0.\2"),
           (r"^blocks\.(\d+)\.norm_(q|k)\.(.*)$", r"blocks.\1.attn1.norm_\2.\3"),
           (r"^blocks\.(\d+)\.attn2\.to_out\.(.*)$", r"blocks.\1.attn2.to_out.0.\2"),
           (r"^blocks\.(\d+)\.ffn\.fc_in\.(.*)$", r"blocks.\1.ffn.net.0.proj.\2"),
           (r"^blocks\.(\d+)\.ffn\.fc_out\.(.*)$", r"blocks.\1.ffn.net.2.\2"),
           (r"^blocks\.(\d+)\.self_attn_residual_norm\.norm\.(.*)$", r"blocks.\1.norm2.\2")]
    for p, r in inv:
        if re.match(p, ref_name):
            return re.sub(p, r, ref_name)
    return ref_name


@pytest.mark.skipif(not R.available(), reason="needs the reference checkout (/root/reference)")
def test_rename_table_equals_the_reference_mapping(golden_dir):
    R.install()
    from fastvideo.configs.models.dits.wanvideo import WanVideoArchConfig
    from fastvideo.models.loader.utils import get_param_names_mapping
    arch = WanVideoArchConfig()
    ref_map = get_param_names_mapping(arch.param_names_mapping)
    lora_map = get_param_names_mapping(arch.lora_param_names_mapping)
    fx = torch.load(os.path.join(golden_dir, "wan_tiny.pt"), weights_only=False)
    names = [_hf_name(k) for k in fx["state_dict"]] + [hf for hf, _ in PAIRS] + [
        "condition_embedder.image_embedder.ff.net.0.proj.weight", "condition_embedder.image_embedder.ff.net.2.bias",
        "condition_embedder.delta_embedder.linear_1.weight", "condition_embedder.delta_embedder.linear_2.bias", "blocks.1.attn2.add_k_proj.weight"]
    for n in names:
        assert L.wan_dit_param_name(n) == ref_map(n)[0], n
    for n in ["blocks.4.self_attn.q.weight", "blocks.4.self_attn.o.bias", "blocks.0.cross_attn.k.weight", "blocks.0.cross_attn.o.weight",
              "blocks.9.ffn.0.bias", "blocks.9.ffn.2.weight", "head.head.weight"]:
        assert L.wan_dit_param_name(n, official_names=True) == ref_map(lora_map(n)[0])[0], n


class _Capture:
    def __init__(self, sd, *args, **kw):
        self.sd, self.args, self.kw = sd, args, kw


def _write_transformer(tmp, sd, shards):
    cfg = {"_class_name": "WanTransformer3DModel", "num_attention_heads": 2, "attention_head_dim": 128, "ffn_dim": 512, "num_layers": 2,
           "patch_size": [1, 2, 2], "text_dim": 64, "freq_dim": 256, "eps": 1e-6, "qk_norm": "rms_norm_across_heads",
           "cross_attn_norm": True, "image_dim": None, "added_kv_proj_dim": None}
    os.makedirs(tmp, exist_ok=True)
    with open(os.path.join(tmp, "config.json"), "w") as f:
        json.dump(cfg, f)
    hf = {_hf_name(k): v.contiguous() for k, v in sd.items()}
    keys = sorted(hf)
    if shards == 1:
        save_file(hf, os.path.join(tmp, "diffusion_pytorch_model.safetensors"))
    else:
        wm = {}
        for s in range(shards):
            fn = f"diffusion_pytorch_model-{s + 1:05d}-of-{shards:05d}.safetensors"
            part = {k: hf[k] for k in keys[s::shards]}
            save_file(part, os.path.join(tmp, fn))
            wm.update({k: fn for k in part})
        with open(os.path.join(tmp, "diffusion_pytorch_model.safetensors.index.json"), "w") as f:
            json.dump({"metadata": {}, "weight_map": wm}, f)


@pytest.mark.parametrize("shards", [1, 3])
def test_load_wan_transformer_round_trip(tmp_path, golden_dir, shards):
    fx = torch.load(os.path.join(golden_dir, "wan_tiny.pt"), weights_only=False)
    d = str(tmp_path / "transformer")
    _write_transformer(d, fx["state_dict"], shards)
    m = L.load_wan_transformer(d, device="cpu", quantization="fp8", model_cls=_Capture)
    assert set(m.sd) == set(fx["state_dict"])
    for k, v in fx["state_dict"].items():
        assert m.sd[k].dtype == torch.bfloat16 and torch.equal(m.sd[k], v.to(torch.bfloat16)), k
    assert m.args[:5] == (2, 128, (1, 2, 2), 1e-6, 256) and m.kw["quantization"] == "fp8" and m.kw["attention"] == "dense"


def test_load_wan_transformer_refuses_i2v(tmp_path, golden_dir):
    fx = torch.load(os.path.join(golden_dir, "wan_tiny.pt"), weights_only=False)
    d = str(tmp_path / "transformer")
    _write_transformer(d, fx["state_dict"], 1)
    cfg = json.load(open(os.path.join(d, "config.json")))
    cfg["image_dim"] = 1280
    json.dump(cfg, open(os.path.join(d, "config.json"), "w"))
    with pytest.raises(ValueError, match="I2V"):
        L.load_wan_transformer(d, device="cpu", model_cls=_Capture)
    with pytest.raises(FileNotFoundError):
        L.load_wan_transformer(str(tmp_path / "missing"), device="cpu", model_cls=_Capture)


def test_load_wan_vae_decoder_reads_only_the_decoder(tmp_path, golden_dir):
    from oracle.vae_oracle import seeded_state_dict
    g = torch.load(os.path.join(golden_dir, "vae_tiny.pt"), weights_only=False)
    sd = seeded_state_dict(g["param_spec"], g["seed"])
    sd["encoder.conv_in.weight"] = torch.zeros(4, 3, 3, 3, 3)           # must be skipped
    d = str(tmp_path / "vae")
    os.makedirs(d)
    json.dump({"_class_name": "AutoencoderKLWan", "base_dim": 32, "z_dim": 16, "dim_mult": [1, 2, 4, 4], "num_res_blocks": 2,
               "temperal_downsample": [False, True, True], "latents_mean": [0.0] * 16, "latents_std": [1.0] * 16},
              open(os.path.join(d, "config.json"), "w"))
    save_file({k: v.contiguous() for k, v in sd.items()}, os.path.join(d, "diffusion_pytorch_model.safetensors"))
    dec, mean, std = L.load_wan_vae_decoder(d, device="cpu", model_cls=_Capture)
    assert all(k.startswith(("decoder.", "post_quant_conv.")) for k in dec.sd) and "decoder.conv_in.weight" in dec.sd
    assert dec.args == ((1, 2, 4, 4), 2, (True, True, False)) and len(mean) == 16 and len(std) == 16
    assert torch.equal(dec.sd["decoder.conv_out.bias"], sd["decoder.conv_out.bias"])
