"""Generate tests/golden/vae_tiny.pt by running the REAL reference VAE decode (imported from /root/reference, CPU fp32).

Run in the build container only:  ``python oracle/make_golden_vae.py``.
Fixture: a Wan2.1-VAE-geometry decoder at base_dim 32 (channels 128/128/128/64/32; same block structure, 33 causal convs,
two temporal upsamplers), seeded init, latent [1,16,3,4,6] (ragged H != W) -> reference pixels [1,3,9,32,48], plus the
bf16-autocast decode of the same latent (the reference's own reduced-precision eager path, spark_performance.md:56)."""
from __future__ import annotations

import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import ref_loader as R  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden", "vae_tiny.pt")


def build_ref_vae(base_dim=32, seed=0):
    R.install()
    from fastvideo.configs.models.vaes.wanvae import WanVAEArchConfig, WanVAEConfig
    from fastvideo.models.vaes.wanvae import AutoencoderKLWan
    cfg = WanVAEConfig(arch_config=WanVAEArchConfig(base_dim=base_dim))
    cfg.load_encoder = False
    vae = AutoencoderKLWan(cfg).float().eval()
    vae.param_spec = init_vae_params(vae, seed)
    return vae


def init_vae_params(vae, seed=0):
    """Deterministic init from oracle.vae_oracle.seeded_state_dict (the tests regenerate the same weights from the spec
    stored in the fixture instead of shipping 30 MB of tensors)."""
    from oracle.vae_oracle import seeded_state_dict
    spec = [(n, tuple(p.shape)) for n, p in vae.named_parameters()]
    sd = seeded_state_dict(spec, seed)
    with torch.no_grad():
        for n, p in vae.named_parameters():
            p.copy_(sd[n])
    return spec


def main():
    vae = build_ref_vae()
    g = torch.Generator().manual_seed(1)
    z = torch.randn((1, 16, 3, 4, 6), generator=g)
    with torch.no_grad():
        y = vae.decode(z)
        with torch.autocast("cpu", dtype=torch.bfloat16):
            y_bf16 = vae.decode(z)
    checksum = sum(float(v.double().abs().sum()) for k, v in vae.state_dict().items() if k.startswith(("decoder.", "post_quant_conv.")))
    torch.save({"param_spec": vae.param_spec, "seed": 0, "weights_abs_sum": checksum, "z": z, "y": y,
                "y_bf16_autocast": y_bf16.float(), "base_dim": 32}, OUT)
    print("wrote", OUT, os.path.getsize(OUT) // 1024, "KiB", "| fp32 vs bf16-autocast max abs diff", (y - y_bf16.float()).abs().max().item(),
          "| y absmax", y.abs().max().item(), "std", y.std().item())


if __name__ == "__main__":
    main()
