"""SURVEY §8 f3 end to end ON THE DEVICE: diffusers-format ``*.safetensors`` shards -> ``fastvideo_amd.loader`` (rename through the
reference's ``param_names_mapping`` restated in loader.py, fastvideo/configs/models/dits/wanvideo.py:16-61; tensors materialised straight
in HBM) -> the HIP models' packed layouts (fused QKV rows, per-layer text-KV panel, e4m3 weights + per-row scales, [Cout, taps*Cin] conv
weights) -> a forward that must equal, BIT FOR BIT, the model constructed directly from the same state dict — and stay inside the DiT / VAE
bounds of the reference's golden outputs.  Reference flow replaced: fastvideo/models/loader/fsdp_load.py:121-206 (meta-device build, shard
stream, rename, FSDP shard) + fp8_config.py:211-245 (convert_model_to_fp8)."""
import json
import os

import pytest
import torch
from safetensors.torch import save_file

pytestmark = pytest.mark.gpu

from test_loader import _write_transformer  # noqa: E402  (the CPU test's shard writer: inverse rename + 1 or 3 shards + index json)


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


@pytest.mark.parametrize("shards,quant", [(1, None), (3, None), (3, "fp8"), (1, "fp8_channel")])
def test_load_wan_transformer_on_device_equals_direct_construction(tmp_path, golden_dir, shards, quant):
    _need_gpu()
    from fastvideo_amd import loader as L
    from fastvideo_amd.wan_dit import WanTransformer3DModelHip
    fx = torch.load(os.path.join(golden_dir, "wan_tiny.pt"), weights_only=False)
    d = str(tmp_path / "transformer")
    _write_transformer(d, fx["state_dict"], shards)
    loaded = L.load_wan_transformer(d, device="cuda", quantization=quant)
    direct = WanTransformer3DModelHip(fx["state_dict"], num_heads=fx["config"]["num_heads"], quantization=quant)
    assert isinstance(loaded, WanTransformer3DModelHip) and loaded.num_layers == direct.num_layers == fx["config"]["num_layers"]
    assert (loaded.H, loaded.D, loaded.patch, loaded.eps, loaded.freq_dim) == (direct.H, direct.D, direct.patch, direct.eps, direct.freq_dim)
    # the packed device layouts are the same bytes
    for bl, bd in zip(loaded.blocks, direct.blocks):
        assert bl.keys() == bd.keys()
        for k in bl:
            if isinstance(bl[k], torch.Tensor):
                assert bl[k].device.type == "cuda" and bl[k].dtype == bd[k].dtype and torch.equal(bl[k].view(torch.uint8), bd[k].view(torch.uint8)), k
    for case in fx["cases"]:
        args = (case["latent"].cuda(), case["ctx"].cuda(), case["timestep"].cuda())
        y, y_direct = loaded(*args), direct(*args)
        assert torch.equal(y, y_direct), f"loaded model differs from the directly constructed one ({quant}, {shards} shards)"
        ref = case["out"].float()
        err = (y.float().cpu() - ref).abs()
        if quant is None:   # the reference's DiT bound (fastvideo/tests/transformers/test_wanvideo.py:109)
            assert (err <= 1e-1 + 1e-2 * ref.abs()).all(), f"max err {err.max().item():.4g} vs the reference's golden output"
        else:               # fp8 linears against the bf16 golden: the bound of tests/test_gpu_model.py::test_wan_tiny_fp8_matches_oracle
            assert 0 < err.mean().item() < 0.1 * ref.abs().mean().item() + 5e-2


def test_load_wan_transformer_official_names_on_device(tmp_path, golden_dir):
    """The original-Wan (LoRA / official checkpoint) key names go through BOTH tables (wanvideo.py:48-61 then :16-43) to the same model."""
    _need_gpu()
    import re
    from fastvideo_amd import loader as L
    from test_loader import _hf_name
    fx = torch.load(os.path.join(golden_dir, "wan_tiny.pt"), weights_only=False)
    inv = [(r"^blocks\.(\d+)\.attn1\.to_(q|k|v)\.(.*)$", r"blocks.\1.self_attn.\2.\3"), (r"^blocks\.(\d+)\.attn1\.to_out\.0\.(.*)$", r"blocks.\1.self_attn.o.\2"),
           (r"^blocks\.(\d+)\.attn2\.to_(q|k|v)\.(.*)$", r"blocks.\1.cross_attn.\2.\3"), (r"^blocks\.(\d+)\.attn2\.to_out\.0\.(.*)$", r"blocks.\1.cross_attn.o.\2"),
           (r"^blocks\.(\d+)\.ffn\.net\.0\.proj\.(.*)$", r"blocks.\1.ffn.0.\2"), (r"^blocks\.(\d+)\.ffn\.net\.2\.(.*)$", r"blocks.\1.ffn.2.\2")]

    def official(hf):
        for p, r in inv:
            if re.match(p, hf):
                return re.sub(p, r, hf)
        return hf

    d = str(tmp_path / "transformer")
    _write_transformer(d, fx["state_dict"], 1)
    from safetensors.torch import load_file
    fn = os.path.join(d, "diffusion_pytorch_model.safetensors")
    hf = load_file(fn)
    save_file({official(k): v for k, v in hf.items()}, fn)
    a = L.load_wan_transformer(d, device="cuda", official_names=True)
    c = fx["cases"][0]
    y = a(c["latent"].cuda(), c["ctx"].cuda(), c["timestep"].cuda())
    _write_transformer(d, fx["state_dict"], 1)
    b = L.load_wan_transformer(d, device="cuda")
    assert torch.equal(y, b(c["latent"].cuda(), c["ctx"].cuda(), c["timestep"].cuda()))
    assert _hf_name("blocks.0.to_q.weight") == "blocks.0.attn1.to_q.weight"


def test_load_wan_vae_decoder_on_device_equals_direct_construction(tmp_path, golden_dir):
    _need_gpu()
    from fastvideo_amd import loader as L
    from fastvideo_amd.wan_vae import WanVaeDecoderHip
    from oracle.vae_oracle import seeded_state_dict
    g = torch.load(os.path.join(golden_dir, "vae_tiny.pt"), weights_only=False)
    sd = seeded_state_dict(g["param_spec"], g["seed"])
    full = dict(sd)
    full["encoder.conv_in.weight"] = torch.zeros(4, 3, 3, 3, 3)   # an encoder tensor in the checkpoint: must be skipped, never uploaded
    d = str(tmp_path / "vae")
    os.makedirs(d)
    json.dump({"_class_name": "AutoencoderKLWan", "base_dim": 32, "z_dim": 16, "dim_mult": [1, 2, 4, 4], "num_res_blocks": 2,
               "temperal_downsample": [False, True, True], "latents_mean": [0.1] * 16, "latents_std": [2.0] * 16},
              open(os.path.join(d, "config.json"), "w"))
    save_file({k: v.contiguous() for k, v in full.items()}, os.path.join(d, "diffusion_pytorch_model.safetensors"))
    dec, mean, std = L.load_wan_vae_decoder(d, device="cuda")
    assert isinstance(dec, WanVaeDecoderHip) and mean == [0.1] * 16 and std == [2.0] * 16
    z = g["z"].cuda()
    y = dec.decode(z)
    y_direct = WanVaeDecoderHip(sd, device="cuda").decode(z)
    assert torch.equal(y, y_direct)
    ref_bf16_err = (g["y_bf16_autocast"] - g["y"]).abs().max().item()
    err = (y.cpu() - g["y"]).abs()
    assert err.mean() <= 1e-2 and err.max() <= max(6e-2, 2 * ref_bf16_err)   # the bound of tests/test_gpu_vae.py::test_decode_tiny_vs_reference_golden
