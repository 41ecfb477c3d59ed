"""Generate tests/golden/vae_full_480p.pt: the REAL reference VAE decode of the FULL contract latent (VERDICT r4 weak #2 / next #1b).

Run in the build container only:  ``python oracle/make_golden_vae_full.py [--case 480p|720p]``  (480p: 9 + 28 minutes of CPU on 8 cores for the
fp32 and the bf16-autocast decode; 720p = BASELINE config 5's decode, 129 frames of 720x1280: about four times that; /root/reference imported).

Test infrastructure (a checker's fixture), never part of the product path.

tests/golden/vae_real.pt stops at 3 latent frames; the bench decodes the contract latent [1,16,21,60,104] -> 81 frames of 480x832
with 4 latent frames per pass, persistent workgroups that stream across tile boundaries and an XCD-aware tile order with frames
fastest (fastvideo_amd/csrc/vae_conv3w.hip).  This fixture is the reference's own `AutoencoderKLWan.decode`
(ref: fastvideo/models/vaes/wanvae.py:1189-1245) of that whole latent at Wan2.1 channel widths with seeded weights, on a sampled
pixel set of EVERY one of the 81 frames and 3 channels:
  y        fp32  the reference's default-precision output
  e_bf16   fp16  |reference bf16-autocast decode - y| per sampled pixel (the reference's own reduced-precision mode, decoding.py:164-180)
The sample is sparser than vae_real.pt's (the seams are crossed there): four 16x16 windows (corner, centre, two random) + a 2-row
band over the full width + a 2-column band over the full height = about 3.6 k pixels per frame."""
from __future__ import annotations

import os
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle.make_golden_vae import build_ref_vae  # noqa: E402

CASES = {"480p": dict(out="vae_full_480p.pt", latent=(1, 16, 21, 60, 104), z_seed=21),      # the contract latent: 81 frames of 480x832
         "720p": dict(out="vae_full_720p.pt", latent=(1, 16, 33, 90, 160), z_seed=22)}      # BASELINE config 5: 129 frames of 720x1280
WEIGHT_SEED = 5


def sample_mask_sparse(H: int, W: int, seed: int) -> torch.Tensor:
    """Deterministic boolean [H, W] mask (the test rebuilds it from (H, W, seed))."""
    g = torch.Generator().manual_seed(seed)
    m = torch.zeros((H, W), dtype=torch.bool)
    wins = [(0, 0), (H // 2 - 8, W // 2 - 8)]
    for _ in range(2):
        wins.append((int(torch.randint(0, H - 16, (1,), generator=g)), int(torch.randint(0, W - 16, (1,), generator=g))))
    for h0, w0 in wins:
        m[h0:h0 + 16, w0:w0 + 16] = True
    r0 = int(torch.randint(16, H - 18, (1,), generator=g))
    c0 = int(torch.randint(16, W - 18, (1,), generator=g))
    m[r0:r0 + 2, :] = True
    m[:, c0:c0 + 2] = True
    return m


def main():
    case = sys.argv[sys.argv.index("--case") + 1] if "--case" in sys.argv else "480p"
    OUT = os.path.join(os.path.dirname(HERE), "tests", "golden", CASES[case]["out"])
    LATENT, Z_SEED = CASES[case]["latent"], CASES[case]["z_seed"]
    torch.set_num_threads(os.cpu_count() or 1)
    vae = build_ref_vae(base_dim=96, seed=WEIGHT_SEED)
    z = torch.randn(LATENT, generator=torch.Generator().manual_seed(Z_SEED))
    part = OUT + ".fp32.part"      # the fp32 decode kept on disk (full output, fp16-free: ~1.4 GB at 720p) so that an interrupted run resumes at the bf16 decode
    t0 = time.time()
    if os.path.exists(part):
        y, dt = torch.load(part, weights_only=False)
        t1 = t0 + dt
        print(f"fp32 decode: resumed from {part} ({dt:.0f} s when it ran)", flush=True)
    else:
        with torch.no_grad():
            y = vae.decode(z)
        t1 = time.time()
        print(f"fp32 decode {t1 - t0:.0f} s -> {tuple(y.shape)}", flush=True)
        torch.save((y, t1 - t0), part)
    H, W = y.shape[-2:]
    m = sample_mask_sparse(H, W, Z_SEED)
    ys = y[0][:, :, m].clone()
    whole = dict(y_absmean=y.abs().mean().item(), y_std=y.std().item(), clamped_frac=(y.abs() == 1).float().mean().item())
    fp32_s = t1 - t0
    t_b0 = time.time()
    with torch.no_grad(), torch.autocast("cpu", dtype=torch.bfloat16):
        yb = vae.decode(z)
    t2 = time.time()
    bf16_s = t2 - t_b0
    e = (yb.float() - y).abs()
    whole.update(e_mean=e.mean().item(), e_max=e.max().item())
    fx = dict(param_spec=vae.param_spec, weight_seed=WEIGHT_SEED, base_dim=96, latent=LATENT, z_seed=Z_SEED, out_shape=tuple(y.shape),
              y=ys, e_bf16=e[0][:, :, m].half(), whole=whole, seconds=dict(fp32=fp32_s, bf16_autocast=bf16_s),
              threads=torch.get_num_threads())
    torch.save(fx, OUT)
    print(f"bf16-autocast decode {bf16_s:.0f} s; sampled {int(m.sum())} px per frame; whole-output stats {whole}")
    print("wrote", OUT, os.path.getsize(OUT) // 1024, "KiB")
    if os.path.exists(part):
        os.remove(part)


if __name__ == "__main__":
    main()
