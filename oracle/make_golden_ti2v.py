"""Generate tests/golden/wan_tiny_ti2v.pt: the REAL reference WanTransformer3DModel (imported from /root/reference, CPU, bf16
autocast) driven with per-token timesteps [B, S] — the Wan2.2 TI2V branch (wanvideo.py:375-385, 690-712, 747-751; the pipeline
builds such timesteps at denoising.py:441-446).  Same tiny model and weights as wan_tiny.pt (rebuilt from seed 0), so only the
inputs and outputs are stored.

Cases: "ti2v" (first-frame tokens t = 0, all others t = 501), "per_frame" (a distinct timestep per latent frame, the
diffusion-forcing / causal pattern), "per_token" (an arbitrary timestep per token).
Run in the build container only:  ``python oracle/make_golden_ti2v.py``."""
from __future__ import annotations

import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import ref_loader as R  # noqa: E402
from oracle.make_golden import TINY  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden", "wan_tiny_ti2v.pt")


def timesteps(kind, grid):
    F_, hw = grid[0], grid[1] * grid[2]
    if kind == "ti2v":
        return torch.cat([torch.zeros(hw), torch.full(((F_ - 1) * hw,), 501.0)]).unsqueeze(0)
    if kind == "per_frame":
        return torch.tensor([900.0, 640.0, 77.0, 333.0, 5.0][:F_]).repeat_interleave(hw).unsqueeze(0)
    g = torch.Generator().manual_seed(9)
    return torch.randint(0, 1000, (1, F_ * hw), generator=g).float()


def main():
    R.init_distributed()
    from fastvideo.forward_context import set_forward_context
    m = R.build_wan(**TINY, seed=0, modulation_std=0.05, dtype=torch.bfloat16)
    ref_sd = torch.load(os.path.join(os.path.dirname(OUT), "wan_tiny.pt"), weights_only=False)["state_dict"]
    assert all(torch.equal(v, ref_sd[k]) for k, v in m.state_dict().items()), "weights differ from wan_tiny.pt"
    cases = []
    for kind, shape, L, seed in [("ti2v", (1, 16, 3, 8, 8), 16, 11), ("per_frame", (1, 16, 4, 10, 14), 24, 12),
                                 ("per_token", (1, 16, 3, 10, 6), 16, 13)]:
        g = torch.Generator().manual_seed(seed)
        x = torch.randn(shape, generator=g).bfloat16()
        ctx = torch.randn(1, L, TINY["text_dim"], generator=g).bfloat16()
        ts = timesteps(kind, (shape[2], shape[3] // 2, shape[4] // 2))
        with torch.no_grad(), torch.autocast("cpu", dtype=torch.bfloat16), set_forward_context(current_timestep=0, attn_metadata=None):
            y = m(hidden_states=x, encoder_hidden_states=ctx, timestep=ts)
        cases.append(dict(kind=kind, latent=x, ctx=ctx, timestep=ts, out=y.detach().clone()))
        print(kind, tuple(ts.shape), tuple(y.shape), float(y.float().abs().mean()))
    torch.save(dict(config=TINY, cases=cases), OUT)
    print("wrote", OUT, os.path.getsize(OUT) // 1024, "KiB")


if __name__ == "__main__":
    main()
