"""Pins oracle/fp8_oracle.py against the reference's own quantisers (golden fixture everywhere; live import when /root/reference exists)."""
import os
import pytest
import torch
from oracle import fp8_oracle as O
from oracle import ref_loader as R

GOLD = os.path.join(os.path.dirname(__file__), "golden", "fp8_quant.pt")


def test_quantisers_match_reference_golden_bit_exact():
    g = torch.load(GOLD)
    qt, st = O.quantize_tensorwise(g["x"])
    qr, sr = O.quantize_rowwise(g["x"])
    assert torch.equal(qt.view(torch.uint8), g["q_tensor"]) and torch.equal(st, g["s_tensor"])
    assert torch.equal(qr.view(torch.uint8), g["q_row"]) and torch.equal(sr, g["s_row"])
    assert sr[5].item() == pytest.approx(1.0 / (448.0 * 512.0))  # the all-zero row hits FP8_MIN_SCALE


@pytest.mark.skipif(not R.available(), reason="needs the reference checkout")
def test_quantisers_match_live_reference():
    R.install()
    from fastvideo.layers.quantization import fp8_config as F8
    x = (torch.randn((64, 128), generator=torch.Generator().manual_seed(3)) * 7).bfloat16()
    for mine, ref in ((O.quantize_tensorwise, F8._quantize_tensorwise), (O.quantize_rowwise, F8._quantize_rowwise)):
        a, b = mine(x), ref(x)
        assert torch.equal(a[0].view(torch.uint8), b[0].view(torch.uint8)) and torch.equal(a[1], b[1])
    assert (O.FP8_MAX, O.FP8_MIN_SCALE) == (F8.FP8_MAX, F8.FP8_MIN_SCALE)
